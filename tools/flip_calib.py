import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ.setdefault("MVK_TUNE", "1")
import numpy as np, torch
import golden_cases as G
import test_assembled_configs as T
from multivae_amd.data.datasets.base import DatasetOutput
d = torch.device("cuda:0")
def stats(tag, og, mg):
    rows = []
    for k, g in og.items():
        r = g.detach().double().reshape(-1); x = mg[k].detach().double().cpu().reshape(-1)
        s = float(r.abs().max().clamp_min(1e-30)); e = (x - r).abs() / s
        rows.append((float((e > 1e-4).double().mean()), float((e > 5e-4).double().mean()), float(e.median()), float(e.max()), e.numel(), k))
    rows.sort(reverse=True)
    print(tag, "worst by frac>1e-4:")
    for r in rows[:8]: print("   f>1e-4 %.4f f>5e-4 %.4f med %.2e max %.2e n %d %s" % r)
    print("   max median over tensors %.2e" % max(r[2] for r in rows))
for name in T.MMVAEPLUS_RESNET_CASES:
    cfg, a = G.load_case(name); sd_np, data = T.mmvaeplus_inputs(cfg)
    model = T.build_mmvaeplus(cfg, d)
    model.load_state_dict({k: G.t(v) for k, v in sd_np.items()}, strict=False)
    with torch.no_grad():
        for k, v in a.items():
            if k.startswith("prior_logvar/"): model.logvars_priors[k.split("/")[1]].copy_(G.t(v).to(d))
    names = cfg["names"]
    noise = {c: {k.split("/")[2]: G.t(v).to(d) for k, v in a.items() if k.startswith(f"noise/{c}/")} for c in names}
    out = model(DatasetOutput(data={m: G.t(v).to(d) for m, v in data.items()}), noise=noise)
    out.loss.backward()
    o, og = T.mmvaeplus_oracle(cfg, a, sd_np, data)
    stats(name, og, T.model_grads(model))
for name in T.JMVAE_CUB_CASES:
    cfg, a = G.load_case(name); sd_np, data = T.jmvae_inputs(cfg)
    model = T.build_jmvae(cfg, d); model.load_state_dict({k: G.t(v) for k, v in sd_np.items()})
    out = model(DatasetOutput(data={m: G.t(v).to(d) for m, v in data.items()}), noise=G.t(a["eps"]).to(d), epoch=cfg["epoch"])
    out.loss.backward()
    o, og, _ = T.jmvae_oracle(cfg, a, sd_np, data)
    stats(name, og, T.model_grads(model))
