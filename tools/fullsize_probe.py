"""Diagnostic: full-size goldens on the DEFAULT dispatch — per-tensor gradient error vs the oracle; dumps the GPU gradients
to gpurun_out/fullsize_<case>.npz so the flip analysis can be developed on the CPU."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as G  # noqa: E402
import test_gpu_golden as T  # noqa: E402

for name in sys.argv[1:] or ["mopoe_mnistsvhn_k10_b512", "mmvae_mnistsvhn_normal_iwae_k1_b256"]:
    cfg, a, dims, data, masks, sd_np, model, inputs, d = T.prep(name)
    if cfg["model"] == "MoPoE":
        out = model(inputs, noise=G.t(a["eps"]).to(d))
    else:
        with torch.no_grad():
            model.prior_log_var.copy_(G.t(a["prior_log_var"]).to(d))
        out = model(inputs, noise={m: G.t(a["noise/" + m]).to(d) for m in cfg["names"]}, detailed_output=True)
    out.loss.backward()
    torch.cuda.synchronize()
    print(name, "loss", float(out.loss), "golden", float(a["loss"]), "rel", abs(float(out.loss) - float(a["loss"])) / abs(float(a["loss"])))
    t0 = time.time()
    o, og = T.oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    print("oracle", time.time() - t0, "s")
    mg = T.model_grads(model)
    dump = {}
    for k, g in og.items():
        if k not in mg:
            continue
        r = g.detach().double().reshape(-1)
        x = mg[k].detach().double().cpu().reshape(-1)
        err = (x - r).abs() / float(r.abs().max())
        print(f"  {k:40s} max {float(err.max()):.2e} median {float(err.median()):.2e} n>1e-4 {int((err > 1e-4).sum())} / {err.numel()}")
        dump[k] = mg[k].detach().cpu().numpy()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fullsize_{name}.npz"), **dump)
