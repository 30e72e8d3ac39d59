"""Which parameter gradients differ between two identical steps (atomics), and the fc weight-gradient check."""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import golden_cases as G
import test_assembled_configs as T
from multivae_amd.data.datasets.base import DatasetOutput
from multivae_amd import kernels as K

d = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "jmvae"
if which == "jmvae":
    cfg, a = G.load_case("jmvae_celeba_cub_resnet")
    sd_np, data = T.jmvae_inputs(cfg)
    model = T.build_jmvae(cfg, d)
    model.load_state_dict({k: G.t(v) for k, v in sd_np.items()})
    inputs = DatasetOutput(data={m: G.t(v).to(d) for m, v in data.items()})
    kw = dict(noise=G.t(a["eps"]).to(d), epoch=cfg["epoch"])
    o, og, _ = T.jmvae_oracle(cfg, a, sd_np, data)
else:
    cfg, a = G.load_case("mmvaeplus_polymnist_resnet_k10")
    sd_np, data = T.mmvaeplus_inputs(cfg)
    model = T.build_mmvaeplus(cfg, d)
    model.load_state_dict({k: G.t(v) for k, v in sd_np.items()}, strict=False)
    with torch.no_grad():
        for k, v in a.items():
            if k.startswith("prior_logvar/"):
                model.logvars_priors[k.split("/")[1]].copy_(G.t(v).to(d))
    names = cfg["names"]
    inputs = DatasetOutput(data={m: G.t(v).to(d) for m, v in data.items()})
    kw = dict(noise={c: {k.split("/")[2]: G.t(v).to(d) for k, v in a.items() if k.startswith(f"noise/{c}/")} for c in names})
    o, og = T.mmvaeplus_oracle(cfg, a, sd_np, data)

def run():
    model.zero_grad(set_to_none=True)
    out = model(inputs, **kw)
    out.loss.backward()
    torch.cuda.synchronize()
    return float(out.loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

l1, g1 = run(); l2, g2 = run()
print("loss", l1, l2, float(o["loss"]))
for k in g1:
    same = torch.equal(g1[k], g2[k])
    e = T.rel(og[k], g1[k]) if k in og else -1
    if not same or e > 1e-4:
        print(f"{k:60s} repro={same} rel-to-max err vs oracle {e:.2e} shape {tuple(g1[k].shape)}")

if which == "jmvae":
    k = "decoders.image.fc.bias"
    ref, got = og[k].double(), g1[k].double().cpu()
    err = (ref - got).abs() / ref.abs().max()
    top = torch.topk(err, 6)
    print("fc.bias elementwise errors (rel to max): top", [f"{v:.2e}" for v in top.values.tolist()], "median", float(err.median()))
    # pre-activation of those units in the oracle (fc output, per sample)
    sd = {kk: G.t(v) for kk, v in sd_np.items()}
    o2, og2, joint = T.jmvae_oracle(cfg, a, sd_np, data)
    z = o2["z"].detach()
    h = torch.nn.functional.linear(z, sd["decoders.image.fc.weight"], sd["decoders.image.fc.bias"])
    for i in top.indices.tolist()[:4]:
        print("unit", i, "fc outputs per sample", h[:, i].tolist(), "grad ref/got", float(ref[i]), float(got[i]))
