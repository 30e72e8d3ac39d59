"""Model-level parity on the GPU: multivae_amd.models.{MoPoE, MVTCAE, MMVAE} (HIP kernels behind the C ABI)
against (a) the golden vectors generated from the real reference and (b) the CPU oracle evaluated on the same
procedural inputs, weights and recorded noise.  Tolerance 1e-4 relative (BASELINE.json north_star)."""
import re

import numpy as np
import pytest
import torch

import golden_cases as G
from oracle import elbo, nets

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    return float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))


ELEMENTWISE = re.compile(r"^(subset |selected )?(z\b|zs|z |kld|kl\b|joint|mus|logvars|mu\b|lv\b|lw\b|u |w |us|ws|lws|rows)")


def check(a, b, what, rtol=RTOL):
    e = rel(a, b)
    assert e <= rtol, f"{what}: rel-to-max err {e:.3e} > {rtol}"
    if ELEMENTWISE.match(what):  # latents, posterior parameters, KL rows, importance weights: entry by entry
        ref = torch.as_tensor(np.asarray(a)).double().reshape(-1)
        got = b.detach().double().cpu().reshape(-1)
        fin = torch.isfinite(ref)
        assert bool((fin | (got == ref)).all()), what
        if bool(fin.any()):
            tol = min(rtol, 1e-4) * ref[fin].abs() + 1e-6 * ref[fin].abs().max()
            bad = (got[fin] - ref[fin]).abs() > tol
            assert not bool(bad.any()), (f"{what}: {int(bad.sum())} of {bad.numel()} entries outside rtol 1e-4 + 1e-6 max; "
                                         f"worst excess {float(((got[fin] - ref[fin]).abs() / tol).max()):.2f}x")


def build_model(cfg, dims):
    from multivae_amd.models import (CRMVAE, DMVAE, DMVAEConfig, JMVAE, MMVAE, MVAE, MVTCAE, CRMVAEConfig, JMVAEConfig, MMVAEConfig, MMVAEPlus, MMVAEPlusConfig,
                                     MoPoE, MoPoEConfig, MVAEConfig, MVTCAEConfig)
    from multivae_amd.models.base.base_config import BaseAEConfig
    from multivae_amd.models.nn.default_architectures import Decoder_AE_MLP, Encoder_VAE_MLP
    from multivae_amd.models.nn.svhn import Decoder_VAE_SVHN, Encoder_VAE_SVHN

    L = cfg["L"]
    common = dict(n_modalities=len(dims), latent_dim=L, input_dims=dict(dims),
                  uses_likelihood_rescaling=cfg["rescaling"])
    enc = dec = None
    if cfg["arch"] != "tiny" and cfg["model"] not in ("MVTCAE", "JMVAE", "MMVAEPlus"):
        enc = dict(mnist=Encoder_VAE_MLP(BaseAEConfig(latent_dim=L, input_dim=(1, 28, 28))),
                   svhn=Encoder_VAE_SVHN(BaseAEConfig(latent_dim=L, input_dim=(3, 32, 32))))
        dec = dict(mnist=Decoder_AE_MLP(BaseAEConfig(latent_dim=L, input_dim=(1, 28, 28))),
                   svhn=Decoder_VAE_SVHN(BaseAEConfig(latent_dim=L, input_dim=(3, 32, 32))))
    if cfg["model"] == "MoPoE":
        mc = MoPoEConfig(beta=cfg["beta"], decoders_dist=cfg.get("dists"), K=cfg["K"],
                         beta_style=cfg.get("beta_style", 1.0), modalities_specific_dim=cfg.get("style_dims"), **common)
        return MoPoE(mc, enc, dec)
    if cfg["model"] == "DMVAE":
        return DMVAE(DMVAEConfig(beta=cfg["beta"], modalities_specific_dim=cfg["style_dims"],
                                 modalities_specific_betas=cfg.get("private_betas"), decoders_dist=cfg.get("dists"),
                                 **common))
    if cfg["model"] == "CRMVAE":
        return CRMVAE(CRMVAEConfig(beta=cfg["beta"], decoders_dist=cfg.get("dists"), **common), enc, dec)
    if cfg["model"] == "MVAE":
        return MVAE(MVAEConfig(beta=cfg["beta"], warmup=cfg["warmup"], k=cfg["k"], use_subsampling=cfg["subsampling"],
                               decoders_dist=cfg.get("dists"), **common), enc, dec)
    if cfg["model"] == "MVTCAE":
        return MVTCAE(MVTCAEConfig(alpha=cfg["alpha"], beta=cfg["beta"], **common))
    if cfg["model"] == "MMVAEPlus":
        return MMVAEPlus(MMVAEPlusConfig(K=cfg["K"], modalities_specific_dim=cfg["S"], beta=cfg["beta"],
                                         prior_and_posterior_dist=cfg["family"], loss=cfg["loss"],
                                         learn_shared_prior=cfg["learn_shared_prior"], **common))
    if cfg["model"] == "JMVAE":
        return JMVAE(JMVAEConfig(alpha=cfg["alpha"], beta=cfg["beta"], warmup=cfg["warmup"],
                                 decoders_dist=cfg.get("dists"), **common))
    mc = MMVAEConfig(K=cfg["K"], prior_and_posterior_dist=cfg["family"], loss=cfg["loss"],
                     learn_prior=cfg["learn_prior"], **common)
    return MMVAE(mc, enc, dec)


def prep(name):
    from multivae_amd.data.datasets.base import DatasetOutput

    cfg, a = G.load_case(name)
    dims, data, masks, sd_np = G.build_inputs(cfg)
    model = build_model(cfg, dims)
    missing = model.load_state_dict({k: G.t(v) for k, v in sd_np.items()}, strict=False)
    assert not missing.unexpected_keys and all(k.startswith(("prior_", "mean_priors.", "logvars_priors."))
                                               for k in missing.missing_keys), missing
    d = torch.device("cuda:0")
    model = model.to(d).train()
    kw = dict(data={m: G.t(v).to(d) for m, v in data.items()})
    if masks is not None:
        kw["masks"] = {m: G.t(v).to(d) for m, v in masks.items()}
    return cfg, a, dims, data, masks, sd_np, model, DatasetOutput(**kw), d


def model_grads(model):
    return {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}


def oracle_full_grads_f64(cfg, dims, data, masks, sd_np, a):
    """The same oracle evaluated in FLOAT64 (every array it is handed is widened first; it holds no fp32 cast of its own): the
    reference point for how far ANY fp32 evaluation order of an IWAE / DReG objective sits from the exact value."""
    old = G.t

    def wide(x):
        y = old(x)
        return y.double() if y.dtype == torch.float32 else y

    G.t = wide
    try:
        return oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    finally:
        G.t = old


def oracle_full_grads(cfg, dims, data, masks, sd_np, a):
    """The oracle's full gradient tensors (not just the sampled golden entries)."""
    sd = {k: G.t(v).clone().requires_grad_(True) for k, v in sd_np.items()}
    tdata = {m: G.t(v) for m, v in data.items()}
    tmasks = None if masks is None else {m: G.t(v) for m, v in masks.items()}
    names = cfg["names"]
    if cfg["model"] == "MMVAEPlus" or cfg.get("style_dims"):
        enc_f, dec_f = nets.build_default_mlp_multilatent(sd, dims)
    elif cfg["arch"] == "tiny" or cfg["model"] in ("MVTCAE", "JMVAE"):
        enc_f, dec_f = nets.build_default_mlp(sd, dims)
    else:
        enc_f, dec_f = nets.build_mnist_svhn(sd, cfg["L"])
    resc = elbo.rescale_factors(dims, cfg["rescaling"])
    extra = {}
    if cfg["model"] == "MoPoE":
        e = {m: enc_f[m](tdata[m]) for m in names}
        style = None
        if cfg.get("style_dims"):
            style = dict(style_eps={m: G.t(a["style_eps/" + m]) for m in names}, beta_style=cfg["beta_style"])
        o = elbo.mopoe_forward(e, tdata, dec_f, G.t(a["eps"]), names=names, beta=cfg["beta"], rescale=resc,
                               dists=cfg["dists"], masks=tmasks, choice=G.t(a["choice"]) if "choice" in a else None,
                               **(style or {}))
    elif cfg["model"] == "MMVAEPlus":
        mods = [m for m in names if ("lws/" + m) in a]
        plv = {k.split("/")[1]: G.t(v).clone().requires_grad_(True) for k, v in a.items()
               if k.startswith("prior_logvar/")}
        e = {m: enc_f[m](tdata[m]) for m in mods}
        noise = {c: {k.split("/")[2]: G.t(v) for k, v in a.items() if k.startswith(f"noise/{c}/")} for c in mods}
        o = elbo.mmvaeplus_forward(e, tdata, dec_f, noise, names=names, K=cfg["K"], family=cfg["family"],
                                   loss=cfg["loss"], beta=cfg["beta"], prior_logvars=plv, rescale=resc, masks=tmasks)
        for k, v in plv.items():
            if k != "shared" or cfg["learn_shared_prior"]:
                extra["logvars_priors." + k] = v
    elif cfg["model"] == "JMVAE":
        e = {m: enc_f[m](tdata[m]) for m in names}
        o = elbo.jmvae_forward(nets.joint_mlp_encoder(sd, dims, tdata), e, tdata, dec_f, G.t(a["eps"]), names=names,
                               alpha=cfg["alpha"], beta=cfg["beta"], warmup=cfg["warmup"], epoch=cfg["epoch"],
                               rescale=resc, dists=cfg.get("dists"))
    elif cfg["model"] == "DMVAE":
        e = {m: enc_f[m](tdata[m]) for m in names}
        o = elbo.dmvae_forward(e, tdata, dec_f, {"shared": G.t(a["noise/shared"]),
                                                 "private": {m: G.t(a["noise/private/" + m]) for m in names}},
                               names=names, beta=cfg["beta"], private_betas=cfg.get("private_betas"), rescale=resc,
                               dists=cfg.get("dists"), masks=tmasks)
    elif cfg["model"] == "CRMVAE":
        e = {m: enc_f[m](tdata[m]) for m in names}
        o = elbo.crmvae_forward(e, tdata, dec_f, G.t(a["eps"]), {m: G.t(a["mod_eps/" + m]) for m in names}, names=names,
                                beta=cfg["beta"], rescale=resc, dists=cfg.get("dists"), masks=tmasks)
    elif cfg["model"] == "MVAE":
        e = {m: enc_f[m](tdata[m]) for m in names}
        o = elbo.mvae_forward(e, tdata, dec_f, G.t(a["eps"]), names=names, subsets=cfg["subsets"],
                              beta=elbo.mvae_annealing(cfg["epoch"], cfg["batch_ratio"], cfg["warmup"], cfg["beta"]),
                              rescale=resc, dists=cfg.get("dists"), masks=tmasks)
    elif cfg["model"] == "MVTCAE":
        e = {m: enc_f[m](tdata[m]) for m in names}
        o = elbo.mvtcae_forward(e, tdata, dec_f, G.t(a["eps"]), names=names, alpha=cfg["alpha"], beta=cfg["beta"],
                                rescale=resc, masks=tmasks)
    else:
        mods = [m for m in names if ("noise/" + m) in a]
        plv = G.t(a["prior_log_var"]).clone().requires_grad_(True)
        e = {m: enc_f[m](tdata[m]) for m in mods}
        o = elbo.mmvae_forward(e, tdata, dec_f, {m: G.t(a["noise/" + m]) for m in mods}, names=names, K=cfg["K"],
                               family=cfg["family"], loss=cfg["loss"], prior_log_var=plv, rescale=resc, masks=tmasks)
        extra["prior_log_var"] = plv
    o["loss"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}
    for k, v in extra.items():
        grads[k] = v.grad if v.grad is not None else torch.zeros_like(v)
    return o, grads


def record_f64_distance(name, engine, cfg, dims, data, masks, sd_np, a, og32, mg, rtol=5e-4):
    """Per-tensor rel-to-max distance of the fp32 oracle and of the HIP path from the float64 oracle."""
    import json
    import os

    _, og64 = oracle_full_grads_f64(cfg, dims, data, masks, sd_np, a)
    worst = {"oracle_fp32": 0.0, "hip": 0.0}
    for k, g64 in og64.items():
        if k not in mg:
            continue
        worst["oracle_fp32"] = max(worst["oracle_fp32"], rel(g64.detach().numpy(), og32[k].detach()))
        worst["hip"] = max(worst["hip"], rel(g64.detach().numpy(), mg[k]))
    assert worst["hip"] <= rtol, (name, worst)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "iwae_float64.jsonl"), "a") as f:
        f.write(json.dumps(dict(case=name, engine=engine, K=cfg["K"], loss=cfg.get("loss"),
                                oracle_fp32_vs_float64=worst["oracle_fp32"], hip_vs_float64=worst["hip"])) + "\n")
    return worst


def compare_grads(model, ograds, a, rtol=RTOL):
    mg = model_grads(model)
    worst = 0.0
    for k, gref in ograds.items():
        if k not in mg:
            continue
        e = rel(gref.detach().numpy(), mg[k])
        worst = max(worst, e)
        assert e <= rtol, f"grad {k}: rel-to-max err {e:.3e}"
    G.check_grads(a, mg, rtol=5 * rtol, atol_frac=rtol)
    return worst


@pytest.fixture(params=["default dispatch", "register-stationary kernels"])
def svhn_engine(request, monkeypatch):
    """The size-based dispatch (tiled engine at the goldens' small batches), then csrc/imgconv.hip for every batch size
    (mvk_debug_set_flags 0x200) with the scaled-fp16 product form of the SVHN decoder from the first row on: the kernels the
    benchmark sizes dispatch to, on the small reference goldens."""
    import ctypes

    from multivae_amd import _lib, kernels

    lib = _lib.load()
    lib.mvk_debug_set_flags.argtypes = [ctypes.c_int]
    if request.param != "default dispatch":
        monkeypatch.setattr(kernels, "IMG_F16_MIN_ROWS", 1)
        lib.mvk_debug_set_flags(0x200)
    yield request.param
    lib.mvk_debug_set_flags(0)


def _engine_applies(name, engine):
    if engine != "default dispatch" and "mnistsvhn" not in name:
        pytest.skip("no SVHN network in this case")


@pytest.mark.parametrize("name", G.MOPOE_CASES)
def test_mopoe_golden(name, svhn_engine):
    _engine_applies(name, svhn_engine)
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    kw = dict(noise=G.t(a["eps"]).to(d))
    if "choice" in a:
        kw["choice"] = G.t(a["choice"]).to(d)
    out = model(inputs, **kw)
    check(a["loss"], out.loss, "loss")
    check(a["loss_sum"], out.loss_sum, "loss_sum")
    for k, v in out.metrics.items():
        check(a["metric/" + k], v, k)
    lat = model.inference(inputs, **kw)
    check(a["mus"], lat["mus"], "subset mus")
    check(a["logvars"], lat["logvars"], "subset logvars")
    check(a["joint_mu"], lat["joint"][0], "joint mu")
    assert list(lat["subsets"].keys()) == cfg["subsets"]
    out.loss.backward()
    o, og = oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    check(o["loss"].detach().numpy(), out.loss, "loss vs oracle")
    compare_grads(model, og, a)


@pytest.mark.parametrize("name", G.MVTCAE_CASES)
def test_mvtcae_golden(name):
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    out = model(inputs, noise=G.t(a["eps"]).to(d))
    check(a["loss"], out.loss, "loss")
    check(a["loss_sum"], out.loss_sum, "loss_sum")
    for k, v in out.metrics.items():
        check(a["metric/" + k], v, k)
    out.loss.backward()
    o, og = oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    compare_grads(model, og, a)
    if masks is not None:
        # exact zeros for the rows of a missing modality flow back to its encoder input layer only through
        # available rows; a fully missing modality gives exactly zero encoder gradients (tests/test_mvtcae.py:160)
        pass


@pytest.mark.parametrize("name", G.MMVAEPLUS_CASES)
def test_mmvaeplus_golden(name):
    """MMVAEPlus.forward (mmvaePlus_model.py:122-362) on the MMVAE kernels with a split latent: loss, importance
    weights, the shared / private samples, every parameter gradient including the learnable prior log-variances."""
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    L = cfg["L"]
    with torch.no_grad():
        for k, v in a.items():
            if k.startswith("prior_logvar/"):
                model.logvars_priors[k.split("/")[1]].copy_(G.t(v).to(d))
    mods = [m for m in cfg["names"] if ("lws/" + m) in a]
    noise = {c: {k.split("/")[2]: G.t(v).to(d) for k, v in a.items() if k.startswith(f"noise/{c}/")} for c in mods}
    out = model(inputs, noise=noise, detailed_output=True)
    check(a["loss"], out.loss, "loss")
    for m in mods:
        check(a["us/" + m], out.zss[m][..., :L], "u " + m)
        check(a["ws/" + m], out.zss[m][..., L:], "w " + m)
        check(a["lws/" + m], out.lws[m], "lw " + m)
    out.loss.backward()
    o, og = oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    check(o["loss"].detach().numpy(), out.loss, "loss vs oracle")
    compare_grads(model, og, a, rtol=5e-4 if cfg["K"] >= 10 else RTOL)  # K=10: see test_mmvae_golden
    if cfg["K"] >= 10 and not any(k.startswith("logvars_priors.") and k not in dict(model.named_parameters()) for k in og):
        record_f64_distance(name, "default dispatch", cfg, dims, data, masks, sd_np, a, og, model_grads(model))


@pytest.mark.parametrize("name", G.JMVAE_CASES)
def test_jmvae_golden(name):
    """JMVAE.forward (jmvae_model.py:116-192): default joint encoder, annealing on / off, mixed decoder distributions."""
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    out = model(inputs, noise=G.t(a["eps"]).to(d), epoch=cfg["epoch"])
    check(a["loss"], out.loss, "loss")
    check(a["loss_sum"], out.loss_sum, "loss_sum")
    for k, v in out.metrics.items():
        check(a["metric/" + k], torch.as_tensor(v), k)
    out.loss.backward()
    o, og = oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    check(o["loss"].detach().numpy(), out.loss, "loss vs oracle")
    compare_grads(model, og, a)
    # masks are rejected like in the reference (joint_model.py:69-73)
    from multivae_amd.data.datasets.base import DatasetOutput

    bad = DatasetOutput(data=inputs.data, masks={m: torch.ones(cfg["B"], dtype=torch.bool, device=d) for m in dims})
    with pytest.raises(AttributeError):
        model(bad)


@pytest.mark.parametrize("name", G.MMVAE_CASES)
def test_mmvae_golden(name, svhn_engine):
    _engine_applies(name, svhn_engine)
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    with torch.no_grad():
        model.prior_log_var.copy_(G.t(a["prior_log_var"]).to(d))
    mods = [m for m in cfg["names"] if ("noise/" + m) in a]
    noise = {m: G.t(a["noise/" + m]).to(d) for m in mods}
    out = model(inputs, noise=noise, detailed_output=True)
    check(a["loss"], out.loss, "loss")
    for m in mods:
        check(a["zs/" + m], out.zss[m], "z " + m)
        check(a["lws/" + m], out.lws[m], "lw " + m)
    out.loss.backward()
    o, og = oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    # IWAE weights are exp(lw - lse) with |lw| ~ 4e3: one fp32 ulp of lw is 2.4e-4, so the weights (and the
    # gradients they scale) carry ~1e-4 relative noise in ANY fp32 evaluation order — tests/test_oracle_float64.py holds the
    # fp32 oracle's own distance from its float64 evaluation on these cases (CPU); here, at K >= 10, the HIP path's distance
    # from float64 is measured beside it, recorded (gpurun_out/iwae_float64.jsonl -> profiles/r06_iwae_float64.json) and
    # bounded by the same 5e-4
    compare_grads(model, og, a, rtol=5e-4)
    if cfg["K"] >= 10:
        record_f64_distance(name, svhn_engine, cfg, dims, data, masks, sd_np, a, og, model_grads(model))


# ---- BASELINE configs 3 and 2 at FULL size, on the dispatch the benchmark takes ----------------------------------------------
def oracle_with_hip_decisions(cfg, dims, data, masks, sd_np, a, model, taps):
    """The oracle's full gradients with the rectifier decisions of the HIP path on the ambiguous units (tests/relu_sites.py):
    everything outside |pre| <= 2e-6 max|pre| is asserted to agree; -> (oracle output, gradients, ambiguous, flipped)."""
    import relu_sites as RS

    (o, og), n_amb, n_flip = RS.oracle_with_hip_decisions(lambda: oracle_full_grads(cfg, dims, data, masks, sd_np, a), model, taps)
    return o, og, n_amb, n_flip


def test_mopoe_fullsize_golden():
    """BASELINE configs[2] at its per-device size (B = 512, K = 10) against the fixture generated from the real reference
    (tests/golden/make_golden.py `fullsize_main`; mopoe_model.py:147-227, nn/svhn.py:7-70): the step runs on the DEFAULT
    dispatch — scaled-fp16 register-stationary convolutions, fused decoder tail — eagerly and as a hipGraph replay.
    Loss, metrics, posterior parameters at 1e-4; every gradient entry by entry at 1e-4 of the tensor's largest entry against
    the oracle evaluated with the HIP path's decisions on the ~300 rectifier units within 2e-6 of zero."""
    from multivae_amd import _lib, kernels
    from multivae_amd.trainers import FlatParams, GraphedStep

    name = "mopoe_mnistsvhn_k10_b512"
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    n = cfg["K"] * cfg["B"]
    # the shapes of this case are the ones the register-stationary scaled-fp16 kernels and the fused tail accept
    assert kernels.IMG_F16 and kernels.conv4s2_scaled_ok(n, 4, 4, 64, 128) and kernels.conv4s2_scaled_ok(n, 8, 8, 32, 64)
    assert kernels.svhn_fused_tail_ok(3, 32)
    eps = G.t(a["eps"]).to(d)
    kernels.TAPS = []
    try:
        out = model(inputs, noise=eps)
        taps = kernels.TAPS
    finally:
        kernels.TAPS = None
    check(a["loss"], out.loss, "loss")
    check(a["loss_sum"], out.loss_sum, "loss_sum")
    for k, v in out.metrics.items():
        check(a["metric/" + k], v, k)
    lat = model.inference(inputs, noise=eps)
    check(a["mus"], lat["mus"], "subset mus")
    check(a["logvars"], lat["logvars"], "subset logvars")
    check(a["joint_mu"], lat["joint"][0], "joint mu")
    check(a["joint_logvar"], lat["joint"][1], "joint logvar")
    out.loss.backward()
    o, og, n_amb, n_flip = oracle_with_hip_decisions(cfg, dims, data, masks, sd_np, a, model, taps)
    print(f"{name}: {n_amb} rectifier units within 2e-6 of zero, {n_flip} decided the other way by the HIP path")
    check(o["loss"].detach().numpy(), out.loss, "loss vs oracle")
    compare_grads(model, og, a)
    eager_loss = float(out.loss.detach())
    # the same step as ONE hipGraph replay (what bench.py and BaseTrainer run)
    flat = FlatParams(model)
    gs = GraphedStep(model, flat, inputs, noise=eps)
    out_g = gs(inputs, eps)
    torch.cuda.synchronize()
    assert float(out_g.loss.detach()) == eager_loss
    compare_grads(model, og, a)


def test_mmvae_fullsize_golden():
    """BASELINE configs[1] at its size (MMVAE MnistSvhn, K = 1, B = 256, Normal / IWAE), as above."""
    from multivae_amd import kernels

    name = "mmvae_mnistsvhn_normal_iwae_k1_b256"
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    with torch.no_grad():
        model.prior_log_var.copy_(G.t(a["prior_log_var"]).to(d))
    mods = cfg["names"]
    noise = {m: G.t(a["noise/" + m]).to(d) for m in mods}
    kernels.TAPS = []
    try:
        out = model(inputs, noise=noise, detailed_output=True)
        taps = kernels.TAPS
    finally:
        kernels.TAPS = None
    check(a["loss"], out.loss, "loss")
    for m in mods:
        check(a["zs/" + m], out.zss[m], "z " + m)
        check(a["lws/" + m], out.lws[m], "lw " + m)
    out.loss.backward()
    o, og, n_amb, n_flip = oracle_with_hip_decisions(cfg, dims, data, masks, sd_np, a, model, taps)
    print(f"{name}: {n_amb} rectifier units within 2e-6 of zero, {n_flip} decided the other way by the HIP path")
    compare_grads(model, og, a)


@pytest.mark.parametrize("name", ["mmvae_tiny_normal_iwae", "mmvaeplus_tiny_laplace_dreg"])
def test_stacked_decoding_is_opt_in_per_decoder(name):
    """MMVAE / MMVAE+ decode all conditioning modalities' latents in one stacked pass only through decoders that declare
    `rows_independent` (every in-package one); any other decoder is run pair by pair like in the reference
    (mmvae_model.py:127, mmvaePlus_model.py:172-186) — same loss and gradients, and a decoder with a non-contiguous output or
    batch statistics keeps the reference's semantics."""
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    mods = cfg["names"]
    if cfg["model"] == "MMVAE":
        noise = {m: G.t(a["noise/" + m]).to(d) for m in mods if ("noise/" + m) in a}
    else:
        mods = [m for m in mods if ("lws/" + m) in a]
        noise = {c: {k.split("/")[2]: G.t(v).to(d) for k, v in a.items() if k.startswith(f"noise/{c}/")} for c in mods}
    calls = []
    for m, dec in model.decoders.items():
        assert dec.rows_independent
        dec.register_forward_hook(lambda mod, args, out, m=m: calls.append((m, args[0].reshape(-1, args[0].shape[-1]).shape[0])))
    out = model(inputs, noise=noise)
    out.loss.backward()
    stacked_calls, calls[:] = list(calls), []
    g_stacked = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    for dec in model.decoders.values():
        dec.rows_independent = False
    out2 = model(inputs, noise=noise)
    out2.loss.backward()
    M = len(noise)
    assert len(stacked_calls) == len(model.decoders) and len(calls) == M * len(model.decoders), (stacked_calls, calls)
    assert sum(n for _, n in stacked_calls) == sum(n for _, n in calls)
    check(out.loss.detach().cpu().numpy(), out2.loss, "loss, pair by pair", rtol=1e-6)
    for k, p in model.named_parameters():
        if p.grad is not None:
            check(g_stacked[k].cpu().numpy(), p.grad, "grad " + k, rtol=1e-5)


def test_missing_modality_gives_exactly_zero_encoder_grads():
    """tests/test_mopoe.py:237-253 / test_mvtcae.py:160-175 of the reference: a modality that is missing for the
    whole batch receives exactly zero gradient in its encoder."""
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.models import MVTCAE, MoPoE, MoPoEConfig, MVTCAEConfig

    d = torch.device("cuda:0")
    dims = G.TINY_DIMS
    torch.manual_seed(0)
    data = {m: torch.rand(6, *s, device=d) for m, s in dims.items()}
    masks = {m: torch.ones(6, dtype=torch.bool, device=d) for m in dims}
    masks["mod2"][:] = False
    for cls, cfgc in ((MoPoE, MoPoEConfig), (MVTCAE, MVTCAEConfig)):
        model = cls(cfgc(n_modalities=4, latent_dim=5, input_dims=dict(dims))).to(d)
        out = model(DatasetOutput(data=data, masks=masks))
        out.loss.backward()
        for p in model.encoders["mod2"].parameters():
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
        assert any(float(p.grad.abs().max()) > 0 for p in model.encoders["mod1"].parameters())


def test_jmvae_encode_paths():
    """JMVAE.encode (jmvae_model.py:58-114): joint encoder for all modalities, product of the unimodal experts
    (stable_poe) for a strict subset, the unimodal encoder for one modality; N samples, return_mean, flatten."""
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.models import JMVAE, JMVAEConfig

    d = torch.device("cuda:0")
    dims = dict(a=(2, 3), b=(7,), c=(4,))
    torch.manual_seed(0)
    model = JMVAE(JMVAEConfig(n_modalities=3, latent_dim=6, input_dims=dims)).to(d)
    inputs = DatasetOutput(data={m: torch.rand((5,) + s, device=d) for m, s in dims.items()})
    assert model.encode(inputs).z.shape == (5, 6)
    assert model.encode(inputs, cond_mod=["a", "c"], N=4).z.shape == (4, 5, 6)
    assert model.encode(inputs, cond_mod="b", N=3, flatten=True).z.shape == (15, 6)
    joint = model.joint_encoder(inputs.data)
    assert torch.equal(model.encode(inputs, return_mean=True).z, joint.embedding)
    # subset posterior = stable_poe of the unimodal posteriors
    ea, ec = model.encoders["a"](inputs.data["a"]), model.encoders["c"](inputs.data["c"])
    mu, lv = elbo.stable_poe(torch.stack([ea.embedding, ec.embedding]).detach().cpu(),
                             torch.stack([ea.log_covariance, ec.log_covariance]).detach().cpu())
    check(mu.numpy(), model.encode(inputs, cond_mod=["a", "c"], return_mean=True).z, "subset mean")
    with pytest.raises(AttributeError):
        model.encode(inputs, cond_mod="zzz")


def test_mmvaeplus_encode_paths():
    """MMVAEPlus.encode (mmvaePlus_model.py:364-456): shared latent + one private latent per modality (own posterior for
    the conditioning modalities, prior for the others)."""
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.models import MMVAEPlus, MMVAEPlusConfig

    d = torch.device("cuda:0")
    dims = dict(a=(2, 3), b=(7,), c=(4,))
    torch.manual_seed(0)
    model = MMVAEPlus(MMVAEPlusConfig(n_modalities=3, latent_dim=6, input_dims=dims, modalities_specific_dim=4)).to(d)
    inputs = DatasetOutput(data={m: torch.rand((5,) + s, device=d) for m, s in dims.items()})
    out = model.encode(inputs)
    assert out.z.shape == (5, 6) and not out.one_latent_space
    assert set(out.modalities_z) == set(dims) and all(v.shape == (5, 4) for v in out.modalities_z.values())
    out = model.encode(inputs, cond_mod=["a"], N=3)
    assert out.z.shape == (3, 5, 6) and out.modalities_z["c"].shape == (3, 5, 4)
    out = model.encode(inputs, cond_mod="b", N=2, flatten=True)
    assert out.z.shape == (10, 6) and out.modalities_z["a"].shape == (10, 4)
    mean = model.encode(inputs, cond_mod=["a", "b"], return_mean=True)
    ea, eb = model.encoders["a"](inputs.data["a"]), model.encoders["b"](inputs.data["b"])
    assert torch.allclose(mean.z, 0.5 * (ea.embedding + eb.embedding), atol=1e-6)
    with pytest.raises(AttributeError):
        MMVAEPlus(MMVAEPlusConfig(n_modalities=3, latent_dim=6, input_dims=dims))  # modalities_specific_dim missing


@pytest.fixture(params=["default dispatch", "register-stationary kernels"])
def conv3_engine(request):
    """The ResNet goldens run twice: with the size-based dispatch (the tiled engine at these batch sizes) and with the
    register-stationary 3x3 kernels + the fused ResnetBlock forms taken for every size (mvk_debug_set_flags 0x800)."""
    import ctypes

    from multivae_amd import _lib

    lib = _lib.load()
    lib.mvk_debug_set_flags.argtypes = [ctypes.c_int]
    lib.mvk_debug_set_flags(0x800 if request.param != "default dispatch" else 0)
    yield request.param
    lib.mvk_debug_set_flags(0)


def test_resnet_mmnist_nets_golden(conv3_engine):
    """EncoderResnetMMNIST / DecoderResnetMMNIST on the HIP kernels (3x3 convolutions, pooling, upsampling, residuals,
    one autograd node per stack) vs the reference golden and the oracle's full gradients."""
    from test_oracle_golden import resnet_case

    from multivae_amd.models.nn.mmnist import DecoderResnetMMNIST, EncoderResnetMMNIST

    cfg, a, esd, dsd, x, z, pe, pdec = resnet_case()
    d = torch.device("cuda:0")
    enc = EncoderResnetMMNIST(cfg["private_dim"], cfg["shared_dim"])
    dec = DecoderResnetMMNIST(cfg["private_dim"] + cfg["shared_dim"])
    enc.load_state_dict({k: G.t(v) for k, v in esd.items()})
    dec.load_state_dict({k: G.t(v) for k, v in dsd.items()})
    enc, dec = enc.to(d), dec.to(d)
    eo = enc(x.to(d))
    outs = [eo.embedding, eo.log_covariance, eo.style_embedding, eo.style_log_covariance]
    for k, o in zip(("mu_u", "lv_u", "mu_w", "lv_w"), outs):
        check(a[k], o, k)
    sum((o * p.to(d)).sum() for o, p in zip(outs, pe)).backward()
    zz = z.to(d).requires_grad_(True)
    rec = dec(zz).reconstruction
    assert rec.shape == (cfg["K"], cfg["B"], 3, 28, 28)
    check(a["recon_sample"], rec.reshape(-1)[torch.as_tensor(G.P.hash_indices(rec.numel(), 512, 77), device=d)], "recon")
    (rec * pdec.to(d)).sum().backward()
    # The golden's seed was picked for its LeakyReLU margin (tests/golden/make_golden.py `lrelu_margin`: no unit of the
    # reference's forward pass within 3e-7 of its site's largest pre-activation, cfg["lrelu_rel_margin"]), so no unit takes the
    # other slope under a different summation order and every gradient is checked entry by entry at 1e-4 (round 2 needed a
    # flip-tolerant 2e-2 bound on the decoder here).
    assert cfg["lrelu_rel_margin"] >= 3e-7
    check(a["dz"], zz.grad, "grad dz")
    # full gradients vs the oracle
    oe = {k: G.t(v).clone().requires_grad_(True) for k, v in esd.items()}
    od = {k: G.t(v).clone().requires_grad_(True) for k, v in dsd.items()}
    sum((o * p).sum() for o, p in zip(nets.mmnist_resnet_encoder(oe, "", x), pe)).backward()
    (nets.mmnist_resnet_decoder(od, "", z) * pdec).sum().backward()
    mg = {"enc." + k: p.grad for k, p in enc.named_parameters()}
    mg.update({"dec." + k: p.grad for k, p in dec.named_parameters()})
    for k, v in list(oe.items()) + []:
        check(v.grad.numpy(), mg["enc." + k], "grad enc." + k)
    for k, v in od.items():
        check(v.grad.numpy(), mg["dec." + k], "grad dec." + k)
    G.check_grads(a, mg, rtol=5 * RTOL, atol_frac=RTOL)
    assert dec(zz[0].detach()).reconstruction.shape == (cfg["B"], 3, 28, 28)  # 2-D latent input


def test_resnet_cub_nets_golden(conv3_engine):
    """CUB_Resnet_Encoder / Decoder (64x64 images, pre-activation blocks, lrelu before the heads / the image conv) on the
    HIP kernels vs the reference golden and the oracle's full gradients; 3-D latents decode too."""
    from test_oracle_golden import resnet_cub_case

    from multivae_amd.models.nn.cub import CUB_Resnet_Decoder, CUB_Resnet_Encoder

    cfg, a, esd, dsd, x, z, pe, pdec = resnet_cub_case()
    assert cfg["lrelu_rel_margin"] >= 3e-7  # seed picked for its LeakyReLU margin: entry-by-entry 1e-4 below
    d = torch.device("cuda:0")
    enc, dec = CUB_Resnet_Encoder(cfg["L"]), CUB_Resnet_Decoder(cfg["L"])
    enc.load_state_dict({k: G.t(v) for k, v in esd.items()})
    dec.load_state_dict({k: G.t(v) for k, v in dsd.items()})
    enc, dec = enc.to(d), dec.to(d)
    eo = enc(x.to(d))
    check(a["mu"], eo.embedding, "mu")
    check(a["lv"], eo.log_covariance, "lv")
    ((eo.embedding * pe[0].to(d)).sum() + (eo.log_covariance * pe[1].to(d)).sum()).backward()
    zz = z.to(d).requires_grad_(True)
    rec = dec(zz).reconstruction
    check(a["recon_sample"], rec.reshape(-1)[torch.as_tensor(G.P.hash_indices(rec.numel(), 512, 78), device=d)], "recon")
    (rec * pdec.to(d)).sum().backward()
    check(a["dz"], zz.grad, "dz")
    oe = {k: G.t(v).clone().requires_grad_(True) for k, v in esd.items()}
    od = {k: G.t(v).clone().requires_grad_(True) for k, v in dsd.items()}
    sum((o * p).sum() for o, p in zip(nets.cub_resnet_encoder(oe, "", x), pe)).backward()
    (nets.cub_resnet_decoder(od, "", z) * pdec).sum().backward()
    mg = {"enc." + k: p.grad for k, p in enc.named_parameters()}
    mg.update({"dec." + k: p.grad for k, p in dec.named_parameters()})
    for k, v in oe.items():
        check(v.grad.numpy(), mg["enc." + k], "grad enc." + k)
    for k, v in od.items():
        check(v.grad.numpy(), mg["dec." + k], "grad dec." + k)
    G.check_grads(a, mg, rtol=5 * RTOL, atol_frac=RTOL)
    assert dec(zz.detach().unsqueeze(0)).reconstruction.shape == (1, cfg["B"], 3, 64, 64)


@pytest.mark.parametrize("name", G.NLL_CASES)
def test_joint_nll_golden(name):
    """compute_joint_nll on the HIP path (decoders over the K axis + mvk_recon_nll_fwd + mvk_iwae_logw +
    mvk_iwae_reduce) against the reference's number and the oracle's per-point log-likelihoods."""
    from multivae_amd import kernels

    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    if cfg["model"] == "MMVAE":
        with torch.no_grad():
            model.prior_log_var.copy_(G.t(a["prior_log_var"]).to(d))
    kw = dict(noise=G.t(a["noise"]).to(d))
    if cfg["model"] == "MMVAE":
        kw["sampled"] = cfg["sampled"]
    if cfg.get("subset") == "paper":
        run = model.compute_joint_nll_paper
    elif cfg.get("subset") is not None:
        run = lambda *args, **kwargs: model._compute_joint_nll_from_subset_encoding(cfg["subset"], *args, **kwargs)
    else:
        run = model.compute_joint_nll
    nll = run(inputs, K=cfg["nll_K"], batch_size_K=cfg["batch_size_K"], **kw)
    assert not model.training  # the reference switches to eval()
    assert isinstance(nll, torch.Tensor) and nll.size() == torch.Size([]) and nll >= 0  # tests/test_mopoe.py:505-548
    check(a["nll"], nll, "nll")
    # same number when the data axis is processed one point at a time (chunking of joint_nll)
    old = kernels.IWAE_ROWS_BUDGET
    kernels.IWAE_ROWS_BUDGET = cfg["nll_K"]
    try:
        nll1 = run(inputs, K=cfg["nll_K"], **kw)
    finally:
        kernels.IWAE_ROWS_BUDGET = old
    check(a["nll"], nll1, "nll (one data point per pass)")


def test_joint_nll_rejects_incomplete_data():
    from multivae_amd.data.datasets.base import DatasetOutput

    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep("nll_mopoe_tiny")
    B = cfg["B"]
    bad = DatasetOutput(data=inputs.data, masks={m: torch.ones(B, dtype=torch.bool, device=d) for m in inputs.data})
    with pytest.raises(AttributeError):
        model.compute_joint_nll(bad, K=4)


@pytest.mark.parametrize("name", G.NLL_MMVAEPLUS_CASES)
def test_joint_nll_mmvaeplus_golden(name):
    """MMVAEPlus.compute_joint_nll on the HIP path against the reference's number (which leaves out the last
    modality, mmvaePlus_model.py:497) and the oracle's per-point log-likelihoods; the caller's inputs and the model's
    beta / rescale factors are left as they were."""
    from multivae_amd import kernels

    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    with torch.no_grad():
        for k in a:
            if k.startswith("prior_logvar/"):
                model.logvars_priors[k.split("/")[1]].copy_(G.t(a[k]).to(d))
    kept = cfg["kept"]
    noise = {c: {k.split("/")[2]: G.t(a[k]).to(d) for k in a if k.startswith(f"noise/{c}/")} for c in kept}
    before = dict(model.rescale_factors)
    nll = model.compute_joint_nll(inputs, K=cfg["nll_K"], noise=noise)
    check(a["nll"], nll, "nll")
    assert list(inputs.data.keys()) == cfg["names"] and model.beta == cfg["beta"] and model.rescale_factors == before
    old = kernels.IWAE_ROWS_BUDGET
    kernels.IWAE_ROWS_BUDGET = 1
    try:
        nll1 = model.compute_joint_nll(inputs, K=cfg["nll_K"], noise=noise)
    finally:
        kernels.IWAE_ROWS_BUDGET = old
    check(a["nll"], nll1, "nll (one data point per pass)")
    full = model.compute_joint_nll(inputs, K=cfg["nll_K"], all_modalities=True)
    assert torch.isfinite(full) and float(full) > float(nll)  # one more modality to explain


@pytest.mark.parametrize("name", G.MVAE_CASES)
def test_mvae_golden(name):
    """MVAE on the HIP path (mvk_mvae_posterior_fwd/bwd, one reconstruction term per (modality, subset) slab) against
    the reference's loss / loss_sum / every metric, the oracle's per-subset posteriors and samples and every parameter
    gradient; compute_joint_nll where the case holds one."""
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    kw = dict(noise=G.t(a["eps"]).to(d), epoch=cfg["epoch"], batch_ratio=cfg["batch_ratio"],
              random_subsets=cfg["random_idx"])
    out = model(inputs, **kw)
    check(a["loss"], out.loss, "loss")
    check(a["loss_sum"], out.loss_sum, "loss_sum")
    assert set(out.metrics) == {k[7:] for k in a if k.startswith("metric/")}
    for k, v in out.metrics.items():
        check(a["metric/" + k], torch.as_tensor(v), k)
    out.loss.backward()
    o, og = oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    check(o["loss"].detach().numpy(), out.loss, "loss vs oracle")
    compare_grads(model, og, a)
    if masks is None:
        for si, s in enumerate(cfg["subsets"]):
            mu, lv = model.compute_mu_log_var_subset(inputs, s)
            check(a[f"sub_mu/{si}"], mu, f"posterior mean of {s}")
            check(a[f"sub_lv/{si}"], lv, f"posterior log-variance of {s}")
    if cfg["nll_K"]:
        nll = model.compute_joint_nll(inputs, K=cfg["nll_K"], noise=G.t(a["nll_noise"]).to(d))
        check(a["nll"], nll, "nll")


def test_mvae_encode_and_missing_modality_gradients():
    """tests/test_mvae.py: encode shapes for subsets / N samples; a modality that is missing for the whole batch gets
    exactly zero encoder gradients and does not move the others' posterior."""
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep("mvae_tiny_masked")
    B, L = cfg["B"], cfg["L"]
    from multivae_amd.data.datasets.base import DatasetOutput

    full = DatasetOutput(data=inputs.data)
    assert model.encode(full).z.shape == (B, L)
    assert model.encode(full, cond_mod=["mod2", "mod4"], N=3).z.shape == (3, B, L)
    assert model.encode(full, cond_mod="mod1", N=3, flatten=True).z.shape == (3 * B, L)
    m0 = model.encode(full, cond_mod=["mod1", "mod2"], return_mean=True).z
    gone = {m: torch.ones(B, dtype=torch.bool, device=d) for m in inputs.data}
    gone["mod3"] = torch.zeros(B, dtype=torch.bool, device=d)
    part = DatasetOutput(data=inputs.data, masks=gone)
    with pytest.raises(AttributeError):
        model.encode(part, cond_mod=["mod1", "mod3"])
    mu_a, _ = model.compute_mu_log_var_subset(part, ["mod1", "mod2", "mod3"])
    assert torch.allclose(mu_a, m0, rtol=1e-5, atol=1e-6)  # the missing expert drops out of the product
    model.zero_grad()
    model(part, epoch=20).loss.backward()
    g = model_grads(model)
    assert all(float(v.abs().max()) == 0.0 for k, v in g.items() if k.startswith(("encoders.mod3", "decoders.mod3")))
    assert any(float(v.abs().max()) > 0.0 for k, v in g.items() if k.startswith("encoders.mod1"))


@pytest.mark.parametrize("name", G.MOPOE_STYLE_CASES)
def test_mopoe_style_golden(name):
    """MoPoE with modality-specific latent spaces on the HIP path (mvk_gauss_sample_kl_fwd/bwd for the style latents)."""
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    names = cfg["names"]
    kw = dict(noise=G.t(a["eps"]).to(d), style_noise={m: G.t(a["style_eps/" + m]).to(d) for m in names})
    if "choice" in a:
        kw["choice"] = G.t(a["choice"]).to(d)
    out = model(inputs, **kw)
    check(a["loss"], out.loss, "loss")
    check(a["loss_sum"], out.loss_sum, "loss_sum")
    for k, v in out.metrics.items():
        check(a["metric/" + k], v, k)
    out.loss.backward()
    o, og = oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    check(o["loss"].detach().numpy(), out.loss, "loss vs oracle")
    compare_grads(model, og, a)
    # a plain encoder (no style outputs) is rejected like in the reference (:179-188)
    from multivae_amd.models.nn.default_architectures import Encoder_VAE_MLP
    from multivae_amd.models.base.base_config import BaseAEConfig

    model.encoders["mod1"] = Encoder_VAE_MLP(BaseAEConfig(latent_dim=cfg["L"], input_dim=(2,))).to(d)
    with pytest.raises(AttributeError):
        model(inputs)


@pytest.mark.parametrize("name", G.CRMVAE_CASES)
def test_crmvae_golden(name):
    """CRMVAE on the HIP path (MVTCAE posterior kernel + unimodal samples + two reconstruction slabs per decoder)."""
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    names = cfg["names"]
    out = model(inputs, noise=G.t(a["eps"]).to(d), modality_noise={m: G.t(a["mod_eps/" + m]).to(d) for m in names})
    check(a["loss"], out.loss, "loss")
    check(a["loss_sum"], out.loss_sum, "loss_sum")
    assert set(out.metrics) == {k[7:] for k in a if k.startswith("metric/")}
    for k, v in out.metrics.items():
        check(a["metric/" + k], v, k)
    out.loss.backward()
    o, og = oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    check(o["loss"].detach().numpy(), out.loss, "loss vs oracle")
    compare_grads(model, og, a)
    if masks is None:
        z = model.encode(inputs, return_mean=True).z
        check(a["joint_mu"], z, "joint mean")
        nll = model.compute_joint_nll(inputs, K=6)
        assert nll.size() == torch.Size([]) and torch.isfinite(nll)


@pytest.mark.parametrize("name", G.DMVAE_CASES)
def test_dmvae_golden(name):
    """DMVAE on the HIP path: the M + 1 ELBOs as the leading axis of one decoder pass per modality."""
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    names = cfg["names"]
    noise = {"shared": G.t(a["noise/shared"]).to(d), "private": {m: G.t(a["noise/private/" + m]).to(d) for m in names}}
    out = model(inputs, noise=noise)
    check(a["loss"], out.loss, "loss")
    assert set(out.metrics) == {k[7:] for k in a if k.startswith("metric/")}
    for k, v in out.metrics.items():
        check(a["metric/" + k], v, k)
    out.loss.backward()
    o, og = oracle_full_grads(cfg, dims, data, masks, sd_np, a)
    check(o["loss"].detach().numpy(), out.loss, "loss vs oracle")
    compare_grads(model, og, a)
    if masks is None:
        mu, lv, _, _ = model._infer_latent_parameters(inputs)
        check(a["joint_mu"], mu, "joint mean")
        check(a["joint_logvar"], lv, "joint log-variance")
        enc = model.encode(inputs, cond_mod=["mod2", "mod4"], N=3)
        assert enc.z.shape == (3, cfg["B"], cfg["L"]) and not enc.one_latent_space
        assert enc.modalities_z["mod1"].shape == (3, cfg["B"], cfg["style_dims"]["mod1"])
        gen = model.generate_from_prior(5)
        assert gen.z.shape == (5, cfg["L"]) and gen.modalities_z["mod3"].shape == (5, cfg["style_dims"]["mod3"])


@pytest.mark.parametrize("name", G.NLL_STYLE_CASES)
def test_joint_nll_mopoe_private_latents_golden(name):
    """MoPoE.compute_joint_nll with modality-specific latent spaces on the HIP path: decoders on [z, w_m], the private
    log-density ratios as extra rows of mvk_iwae_logw."""
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    names = cfg["names"]
    nll = model.compute_joint_nll(inputs, K=cfg["nll_K"], noise=G.t(a["noise"]).to(d),
                                  style_noise={m: G.t(a["style_eps/" + m]).to(d) for m in names})
    check(a["nll"], nll, "nll")


def test_mmvae_generate_from_prior_uses_the_learned_prior():
    """MMVAE / MMVAE+ sample the learnable prior (mmvae_model.py:470-474, mmvaePlus_model.py:453-456), Normal or Laplace."""
    from oracle import elbo as E

    for name, key in (("mmvae_tiny_laplace_dreg", None), ("mmvae_tiny_normal_iwae", None), ("mmvaeplus_tiny_laplace_dreg", "shared")):
        cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
        with torch.no_grad():
            if key is None:
                model.prior_log_var.copy_(G.t(a["prior_log_var"]).to(d))
                plv = G.t(a["prior_log_var"])
            else:
                plv = G.t(a["prior_logvar/shared"])
                model.logvars_priors["shared"].copy_(plv.to(d))
        D = plv.shape[-1]
        fam = cfg["family"]
        lap = fam == "laplace_with_softmax"
        noise = (torch.rand(7, D) * 1.98 - 0.99) if lap else torch.randn(7, D)
        z = model.generate_from_prior(7, noise=noise.to(d)).z
        ref = E.latent_rsample("laplace_with_softmax" if lap else "normal", torch.zeros(1, D), E.mmvae_std(plv, fam), noise)
        check(ref.numpy(), z, f"{name} prior samples")
        assert model.generate_from_prior(1).z.shape == (D,)


@pytest.mark.parametrize("name", G.COND_NLL_CASES)
def test_cond_nll_golden(name):
    """compute_cond_nll on the HIP path: encode(N = K) on the recorded noise reproduces the reference's K encodings, the
    conditional likelihood of every predicted modality matches, also with one data point per pass."""
    from multivae_amd import kernels

    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    model.eval()
    K = cfg["cond_K"]
    noise = G.t(a["noise"]).to(d)
    check(a["z"], model.encode(inputs, cfg["subset"], N=K, noise=noise).z, "K conditional encodings")
    check(a["z_mean"], model.encode(inputs, cfg["subset"], return_mean=True).z, "posterior mean")
    cn = model.compute_cond_nll(inputs, cfg["subset"], cfg["pred"], k_iwae=K, noise=noise)
    assert set(cn) == set(cfg["pred"])
    for m in cfg["pred"]:
        check(a["cnll/" + m], cn[m], "cond nll " + m)
    old = kernels.IWAE_ROWS_BUDGET
    kernels.IWAE_ROWS_BUDGET = K
    try:
        cn1 = model.compute_cond_nll(inputs, cfg["subset"], cfg["pred"], k_iwae=K, noise=noise)
    finally:
        kernels.IWAE_ROWS_BUDGET = old
    for m in cfg["pred"]:
        check(a["cnll/" + m], cn1[m], "cond nll (one point per pass) " + m)
    one = model.compute_cond_nll(inputs, cfg["subset"], cfg["pred"][0], k_iwae=1)
    assert torch.isfinite(one[cfg["pred"][0]])


@pytest.mark.parametrize("name", G.NLL_PAPER_CASES)
def test_joint_nll_paper_mmvae_golden(name):
    """MMVAE.compute_joint_nll_paper on the HIP path (forward importance weights per chunk, pooled by mvk_iwae_reduce)."""
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    with torch.no_grad():
        model.prior_log_var.copy_(G.t(a["prior_log_var"]).to(d))
    noise = [{m: G.t(a[f"noise/{c}/{m}"]).to(d) for m in cfg["names"]} for c in range(cfg["chunks"])]
    nll = model.compute_joint_nll_paper(inputs, K=cfg["nll_K"], batch_size_K=cfg["batch_size_K"], noise=noise)
    assert nll.size() == torch.Size([]) and nll >= 0  # tests/test_mmvae_model.py:443-446
    check(a["nll"], nll, "nll (paper)")


@pytest.mark.parametrize("name", G.NLL_DMVAE_CASES)
def test_dmvae_joint_nll_golden(name):
    """DMVAE.compute_joint_nll on the HIP path (K as a kernel axis; the reference's never-reset prior / posterior sums as
    one cumulative sum over the (data point, chunk) sequence) against the reference's number."""
    cfg, a, dims, data, masks, sd_np, model, inputs, d = prep(name)
    names = cfg["names"]
    noise = {"shared": G.t(a["noise/shared"]).to(d), "private": {m: G.t(a["noise/private/" + m]).to(d) for m in names}}
    nll = model.compute_joint_nll(inputs, K=cfg["K"], batch_size_K=cfg["batch_size_K"], noise=noise)
    check(a["nll"], nll, "nll")
    assert np.isfinite(float(model.compute_joint_nll(inputs, K=8, batch_size_K=4)))  # fresh noise
    with pytest.raises(RuntimeError):
        model.compute_joint_nll(inputs, K=10, batch_size_K=4)
