"""The evidence behind the ONE tolerance above north_star's 1e-4: IWAE / DReG gradients at K >= 10.

The importance weights are exp(lw - logsumexp(lw)) with |lw| ~ 3e3 - 4e3 (reference: models/mmvae/mmvae_model.py:238-292,
models/mmvaePlus/mmvaePlus_model.py:270-362): one fp32 ulp of lw is 2.4e-4, so the weights — and every gradient they scale — carry
~1e-4 of relative noise in ANY fp32 evaluation order.  This test makes that executable on the CPU: the oracle (itself pinned to the
reference's fp32 outputs by the golden fixtures) is evaluated in fp32 and in float64 on the K = 10 goldens; the per-tensor
rel-to-max distance between the two is what an fp32 implementation CAN differ from the exact gradients by, and is asserted to
be (a) above the 1e-6 an ordinary fp32 chain shows — the phenomenon is real — and (b) inside the 5e-4 the GPU tests allow for
these cases.  The GPU tests (tests/test_gpu_golden.py::test_mmvae_golden / test_mmvaeplus_golden at K >= 10) measure the HIP
path's distance from the same float64 gradients beside it and record both (profiles/r06_iwae_float64.json)."""
import json
import os

import pytest
import torch

import golden_cases as G
import test_gpu_golden as TG

CASES = [n for n in list(G.MMVAE_CASES) + list(G.MMVAEPLUS_CASES) if G.load_case(n)[0]["K"] >= 10]


@pytest.mark.parametrize("name", CASES)
def test_fp32_oracle_vs_its_float64_evaluation(name):
    cfg, a = G.load_case(name)
    dims, data, masks, sd_np = G.build_inputs(cfg)
    o32, g32 = TG.oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    o64, g64 = TG.oracle_full_grads_f64(cfg, dims, data, masks, sd_np, a)
    assert o64["loss"].dtype == torch.float64 and all(g.dtype == torch.float64 for g in g64.values())
    assert abs(float(o32["loss"]) - float(o64["loss"])) <= 1e-5 * abs(float(o64["loss"]))  # the LOSS is an ordinary fp32 sum
    dist = {k: TG.rel(g64[k].detach().numpy(), g32[k].detach()) for k in g64}
    worst = max(dist.values())
    print(json.dumps(dict(case=name, K=cfg["K"], loss=cfg.get("loss"), oracle_fp32_vs_float64=worst,
                          worst_tensor=max(dist, key=dist.get))))
    assert 1e-6 < worst <= 5e-4, dist
