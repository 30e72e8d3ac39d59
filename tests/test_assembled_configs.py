"""BASELINE.json configs[3] (MMVAE+ on PolyMNIST-shaped data with the ResNet encoders / decoders) and configs[4] (JMVAE on a
64x64 image + attribute vector with the CUB ResNets and the default joint encoder) ASSEMBLED: networks and ELBO together.

Goldens come from the real reference (tests/golden/make_golden.py, `assembled_main`: examples/mmvae_plus/mmnist.py:19-45,
models/nn/mmnist.py:254-366, models/nn/cub.py:144-246, models/jmvae/jmvae_model.py:116-192).  CPU tests pin the oracle on
them; GPU tests compare the HIP path with the goldens and with the oracle's full gradients, and check size-independent
properties of the full-size configurations."""
import numpy as np
import pytest
import torch

import golden_cases as G
import relu_sites as RS
from oracle import elbo, nets

MMVAEPLUS_RESNET_CASES = ["mmvaeplus_polymnist_resnet_k10", "mmvaeplus_polymnist_resnet_dreg"]
JMVAE_CUB_CASES = ["jmvae_celeba_cub_resnet", "jmvae_celeba_cub_resnet_trained"]
RTOL = 1e-4
# units inside the 2e-6 band of the oracle's pre-activations per assembled case, as recorded in profiles/r05_flip_counts.json (a
# property of the fixture and the CPU oracle, not of the HIP kernels): a change of fixture or oracle that widened the band
# would show here instead of passing silently
AMBIGUOUS_R05 = {"mmvaeplus_polymnist_resnet_k10": 1256, "mmvaeplus_polymnist_resnet_dreg": 166,
                 "jmvae_celeba_cub_resnet": 47, "jmvae_celeba_cub_resnet_trained": 38}


def rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double().reshape(-1)
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double().reshape(-1)
    return float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))


def check(a, b, what, rtol=RTOL):
    e = rel(a, b)
    assert e <= rtol, f"{what}: rel-to-max err {e:.3e} > {rtol}"


# ---- procedural inputs ---------------------------------------------------------------------------------------------
def mmvaeplus_inputs(cfg):
    names, seed, B = cfg["names"], cfg["seed"], cfg["B"]
    sd = G.P.make_state_dict(G.P.mmvaeplus_resnet_shapes(names, cfg["S"], cfg["L"]), seed)
    data = {m: G.P.uniform((B, 3, 28, 28), seed + 50 + i) for i, m in enumerate(names)}
    return sd, data


def jmvae_inputs(cfg):
    seed, B, n_attr = cfg["seed"], cfg["B"], cfg["n_attr"]
    sd = G.P.make_state_dict(G.P.jmvae_cub_shapes(cfg["L"], n_attr), seed)
    data = dict(image=G.P.uniform((B, 3, 64, 64), seed + 2),
                attributes=(G.P.uniform((B, n_attr), seed + 3) > 0.5).astype(np.float32))
    return sd, data


def mmvaeplus_oracle(cfg, a, sd_np, data):
    names = cfg["names"]
    sd = {k: G.t(v).clone().requires_grad_(True) for k, v in sd_np.items()}
    plv = {k.split("/")[1]: G.t(v).clone().requires_grad_(k.split("/")[1] != "shared") for k, v in a.items()
           if k.startswith("prior_logvar/")}
    tdata = {m: G.t(v) for m, v in data.items()}
    e = {m: nets.mmnist_resnet_encoder(sd, f"encoders.{m}.", tdata[m]) for m in names}
    dec = {m: (lambda z, m=m: nets.mmnist_resnet_decoder(sd, f"decoders.{m}.", z)) for m in names}
    noise = {c: {k.split("/")[2]: G.t(v) for k, v in a.items() if k.startswith(f"noise/{c}/")} for c in names}
    dims = {m: (3, 28, 28) for m in names}
    o = elbo.mmvaeplus_forward(e, tdata, dec, noise, names=names, K=cfg["K"], family=cfg["family"], loss=cfg["loss"],
                               beta=cfg["beta"], prior_logvars=plv, rescale=elbo.rescale_factors(dims, False),
                               dists={m: "laplace" for m in names}, dist_scales={m: cfg["scale"] for m in names})
    o["loss"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}
    for k, v in plv.items():
        if k != "shared":
            grads["logvars_priors." + k] = v.grad
    return o, grads


def jmvae_oracle(cfg, a, sd_np, data):
    names = cfg["names"]
    sd = {k: G.t(v).clone().requires_grad_(True) for k, v in sd_np.items()}
    tdata = {m: G.t(v) for m, v in data.items()}
    fns = dict(image=nets.cub_resnet_encoder, attributes=nets.mlp_encoder)
    e = {m: fns[m](sd, f"encoders.{m}.", tdata[m]) for m in names}
    dec = dict(image=lambda z: nets.cub_resnet_decoder(sd, "decoders.image.", z),
               attributes=lambda z: nets.mlp_decoder(sd, "decoders.attributes.", z, (cfg["n_attr"],)))
    joint = nets.joint_encoder_generic(sd, {m: fns[m] for m in names}, tdata)
    dims = dict(image=(3, 64, 64), attributes=(cfg["n_attr"],))
    o = elbo.jmvae_forward(joint, e, tdata, dec, G.t(a["eps"]), names=names, alpha=cfg["alpha"], beta=cfg["beta"],
                           warmup=cfg["warmup"], epoch=cfg["epoch"], rescale=elbo.rescale_factors(dims, False),
                           dists=cfg["dists"])
    o["loss"].backward()
    return o, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}, joint


# ---- CPU: the oracle is pinned on the reference's numbers -----------------------------------------------------------
@pytest.mark.parametrize("name", MMVAEPLUS_RESNET_CASES)
def test_oracle_reproduces_mmvaeplus_resnet_golden(name):
    cfg, a = G.load_case(name)
    sd_np, data = mmvaeplus_inputs(cfg)
    o, grads = mmvaeplus_oracle(cfg, a, sd_np, data)
    assert rel(a["loss"], o["loss"]) <= 1e-6
    for m in cfg["names"]:
        assert rel(a["lws/" + m], o["lws"][m]) <= 1e-6
    G.check_grads(a, grads, rtol=1e-5, atol_frac=1e-6)


@pytest.mark.parametrize("name", JMVAE_CUB_CASES)
def test_oracle_reproduces_jmvae_cub_golden(name):
    cfg, a = G.load_case(name)
    sd_np, data = jmvae_inputs(cfg)
    o, grads, joint = jmvae_oracle(cfg, a, sd_np, data)
    assert rel(a["loss"], o["loss"]) <= 1e-6
    assert rel(a["joint_mu"], joint[0]) <= 1e-6
    for k, v in o["metrics"].items():
        assert rel(a["metric/" + k], torch.as_tensor(v)) <= 1e-6, k
    G.check_grads(a, grads, rtol=1e-5, atol_frac=1e-6)


# ---- GPU: the HIP path against the goldens and the oracle ------------------------------------------------------------
def build_mmvaeplus(cfg, device):
    from multivae_amd.models import MMVAEPlus, MMVAEPlusConfig
    from multivae_amd.models.nn.mmnist import DecoderResnetMMNIST, EncoderResnetMMNIST

    names = cfg["names"]
    mc = MMVAEPlusConfig(n_modalities=len(names), latent_dim=cfg["L"], input_dims={m: (3, 28, 28) for m in names},
                         K=cfg["K"], modalities_specific_dim=cfg["S"], prior_and_posterior_dist=cfg["family"],
                         loss=cfg["loss"], beta=cfg["beta"], decoders_dist={m: "laplace" for m in names},
                         decoder_dist_params={m: dict(scale=cfg["scale"]) for m in names},
                         learn_shared_prior=False, learn_modality_prior=True)
    model = MMVAEPlus(mc, {m: EncoderResnetMMNIST(cfg["S"], cfg["L"]) for m in names},
                      {m: DecoderResnetMMNIST(cfg["L"] + cfg["S"]) for m in names})
    return model.to(device).train()


def build_jmvae(cfg, device):
    from multivae_amd.models import JMVAE, JMVAEConfig
    from multivae_amd.models.base.base_config import BaseAEConfig
    from multivae_amd.models.nn.cub import CUB_Resnet_Decoder, CUB_Resnet_Encoder
    from multivae_amd.models.nn.default_architectures import Decoder_AE_MLP, Encoder_VAE_MLP

    L, n_attr = cfg["L"], cfg["n_attr"]
    mc = JMVAEConfig(n_modalities=2, latent_dim=L, input_dims=dict(image=(3, 64, 64), attributes=(n_attr,)),
                     alpha=cfg["alpha"], beta=cfg["beta"], warmup=cfg["warmup"], decoders_dist=cfg["dists"])
    enc = dict(image=CUB_Resnet_Encoder(L), attributes=Encoder_VAE_MLP(BaseAEConfig(latent_dim=L, input_dim=(n_attr,))))
    dec = dict(image=CUB_Resnet_Decoder(L), attributes=Decoder_AE_MLP(BaseAEConfig(latent_dim=L, input_dim=(n_attr,))))
    return JMVAE(mc, enc, dec).to(device).train()


# Gradients of the assembled cases.  LeakyReLU(0.2) networks in fp32: a unit whose pre-activation is ~1e-8 gets the other slope
# when the forward pass sums in another order, which changes that unit's gradient by a factor 5 and everything upstream of it
# by its share (measured on `jmvae_celeba_cub_resnet`: one flipped unit of the last decoder block shifts the tensors behind it
# by a dense 1e-4 ... 6e-3 of their largest entry).  These cases have 1e7 ... 1e8 LeakyReLU units, ~10 per million within 1e-6
# of zero in the reference itself, so no choice of seed gives every unit a margin.  The tests therefore make the decisions
# explicit (tests/relu_sites.py): every unit farther than 2e-6 of its layer's largest pre-activation from zero must land on
# the oracle's side; for the units inside that band the oracle is re-evaluated with the HIP path's decisions, and then EVERY
# gradient tensor is compared entry by entry — no tensor is exempted and nothing falls back to a looser statement.


def model_grads(model):
    return {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}


@pytest.mark.gpu
@pytest.mark.parametrize("name", MMVAEPLUS_RESNET_CASES)
def test_mmvaeplus_resnet_golden_gpu(name, conv3_engine):
    from multivae_amd.data.datasets.base import DatasetOutput

    cfg, a = G.load_case(name)
    sd_np, data = mmvaeplus_inputs(cfg)
    d = torch.device("cuda:0")
    model = build_mmvaeplus(cfg, d)
    missing = model.load_state_dict({k: G.t(v) for k, v in sd_np.items()}, strict=False)
    assert not missing.unexpected_keys and all(k.startswith(("mean_priors.", "logvars_priors."))
                                               for k in missing.missing_keys), missing
    with torch.no_grad():
        for k, v in a.items():
            if k.startswith("prior_logvar/"):
                model.logvars_priors[k.split("/")[1]].copy_(G.t(v).to(d))
    names, L = cfg["names"], cfg["L"]
    noise = {c: {k.split("/")[2]: G.t(v).to(d) for k, v in a.items() if k.startswith(f"noise/{c}/")} for c in names}
    from multivae_amd import kernels

    kernels.TAPS = []
    try:
        out = model(DatasetOutput(data={m: G.t(v).to(d) for m, v in data.items()}), noise=noise, detailed_output=True)
        taps = kernels.TAPS
    finally:
        kernels.TAPS = None
    check(a["loss"], out.loss, "loss")
    for m in names:
        check(a["us/" + m], out.zss[m][..., :L], "u " + m)
        check(a["ws/" + m], out.zss[m][..., L:], "w " + m)
        check(a["lws/" + m], out.lws[m], "lw " + m)
    out.loss.backward()
    (o, og), n_amb, n_flip = RS.oracle_with_hip_decisions(lambda: mmvaeplus_oracle(cfg, a, sd_np, data), model, taps)
    rec = RS.record_counts(name, conv3_engine)  # -> gpurun_out/flip_counts.jsonl -> profiles/rNN_flip_counts.json
    assert rec["flipped"] == n_flip <= n_amb
    assert n_amb <= 2 * AMBIGUOUS_R05[name], (name, n_amb)  # the band's population is the fixture's
    check(o["loss"].detach(), out.loss, "loss vs oracle")
    mg = model_grads(model)
    # K > 1: the importance weights are exp(lw - lse) with |lw| ~ 3e3, one fp32 ulp of lw is 2.4e-4, so the weights (and the
    # gradients they scale) carry ~1e-4 relative noise in ANY fp32 evaluation order — the CPU fp32 oracle itself is 2e-4 - 3e-4 away
    # from its own float64 evaluation on the MnistSvhn K = 10 goldens (tests/test_oracle_float64.py: 2.0e-4 - 2.7e-4, CPU, asserted);
    # hence 5e-4 for the IWAE / DReG gradients, as in
    # test_gpu_golden.py's K > 1 cases (measured with the decisions reconciled: 1.1e-4 at K = 3)
    rtol = 5e-4 if cfg["K"] > 1 else RTOL
    for k, g in og.items():
        check(g, mg[k], "grad " + k, rtol=rtol)
    # the fixture holds the REFERENCE's gradients, i.e. the reference's decisions: every tensor when nothing flipped, and the
    # tensors no flipped unit can reach otherwise (the fixture was skipped altogether on any flip)
    keep = RS.unreachable_by_flips(list(mg))
    assert n_flip > 0 or len(keep) == len(mg)
    G.check_grads(a, {k: mg[k] for k in keep}, rtol=5 * rtol, atol_frac=rtol)


@pytest.fixture(params=["default dispatch", "register-stationary kernels"])
def conv3_engine(request):
    """As in test_gpu_golden.py: the size-based dispatch, then conv3rs.hip + the fused ResnetBlock forms for every size."""
    import ctypes

    from multivae_amd import _lib

    lib = _lib.load()
    lib.mvk_debug_set_flags.argtypes = [ctypes.c_int]
    lib.mvk_debug_set_flags(0x800 if request.param != "default dispatch" else 0)
    yield request.param
    lib.mvk_debug_set_flags(0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", JMVAE_CUB_CASES)
def test_jmvae_cub_golden_gpu(name, conv3_engine):
    from multivae_amd.data.datasets.base import DatasetOutput

    cfg, a = G.load_case(name)
    sd_np, data = jmvae_inputs(cfg)
    d = torch.device("cuda:0")
    model = build_jmvae(cfg, d)
    model.load_state_dict({k: G.t(v) for k, v in sd_np.items()})
    from multivae_amd import kernels

    kernels.TAPS = []
    try:
        out = model(DatasetOutput(data={m: G.t(v).to(d) for m, v in data.items()}), noise=G.t(a["eps"]).to(d),
                    epoch=cfg["epoch"])
        taps = kernels.TAPS
    finally:
        kernels.TAPS = None
    check(a["loss"], out.loss, "loss")
    check(a["loss_sum"], out.loss_sum, "loss_sum")
    for k, v in out.metrics.items():
        check(a["metric/" + k], torch.as_tensor(v), k)
    out.loss.backward()
    (o, og, _), n_amb, n_flip = RS.oracle_with_hip_decisions(lambda: jmvae_oracle(cfg, a, sd_np, data), model, taps)
    rec = RS.record_counts(name, conv3_engine)
    assert rec["flipped"] == n_flip <= n_amb
    assert n_amb <= 2 * AMBIGUOUS_R05[name], (name, n_amb)
    check(o["loss"].detach(), out.loss, "loss vs oracle")
    mg = model_grads(model)
    for k, g in og.items():
        check(g, mg[k], "grad " + k)
    keep = RS.unreachable_by_flips(list(mg))  # the REFERENCE's gradient samples wherever no flipped unit reaches
    assert n_flip > 0 or len(keep) == len(mg)
    G.check_grads(a, {k: mg[k] for k in keep}, rtol=5e-4, atol_frac=1e-4)


@pytest.mark.gpu
def test_cfg4_full_size_step_properties():
    """BASELINE configs[3] at FULL size (5 modalities, K = 10, 32 + 32 latent dimensions, batch 256): one training step
    runs, and the objective has its size-independent properties: (a) permuting the K samples of every noise tensor
    leaves the loss and every gradient unchanged, (b) the loss of a batch whose two halves are equal is twice the loss
    of one half with the same noise (sum over the batch), (c) the step is reproducible bit for bit."""
    from multivae_amd.data.datasets.base import DatasetOutput

    d = torch.device("cuda:0")
    names = [f"m{i}" for i in range(5)]
    cfg = dict(names=names, L=32, S=32, K=10, family="laplace_with_softmax", loss="iwae_looser", beta=2.5, scale=0.75)
    torch.manual_seed(0)
    model = build_mmvaeplus(cfg, d)
    B, K, L, S = 256, 10, 32, 32
    g = torch.Generator(device=d).manual_seed(1)
    half = {m: torch.rand(B // 2, 3, 28, 28, device=d, generator=g) for m in names}
    data = {m: torch.cat([v, v]) for m, v in half.items()}
    eps_lo = torch.finfo(torch.float32).eps - 1

    def draw(n):
        x = torch.empty(K, B // 2, n, device=d).uniform_(eps_lo, 1, generator=g)
        return torch.cat([x, x], 1)

    noise = {c: dict(u=draw(L), w=draw(S), **{r: draw(S) for r in names if r != c}) for c in names}

    def run(inputs, nz):
        model.zero_grad(set_to_none=True)
        out = model(inputs, noise=nz)
        out.loss.backward()
        torch.cuda.synchronize()
        return float(out.loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    l1, g1 = run(DatasetOutput(data=data), noise)
    assert np.isfinite(l1)
    perm = torch.randperm(K, device=d, generator=g)
    l2, g2 = run(DatasetOutput(data=data), {c: {k: v[perm] for k, v in nz.items()} for c, nz in noise.items()})
    assert abs(l1 - l2) <= 1e-5 * abs(l1), (l1, l2)
    for k in g1:
        e = float((g1[k] - g2[k]).abs().max() / g1[k].abs().max().clamp_min(1e-30))
        assert e <= 2e-3, (k, e)  # importance weights in fp32 (see the golden test): order-dependent at the 1e-4..1e-3 level
    lh, _ = run(DatasetOutput(data=half), {c: {k: v[:, : B // 2] for k, v in nz.items()} for c, nz in noise.items()})
    assert abs(l1 - 2 * lh) <= 1e-5 * abs(l1), (l1, lh)
    l3, g3 = run(DatasetOutput(data=data), noise)
    assert l3 == l1 and all(torch.equal(g1[k], g3[k]) for k in g1), "the step is not bit-reproducible"


@pytest.mark.gpu
def test_cfg5_full_size_step_properties():
    """BASELINE configs[4] at full size (64x64 image ResNets + 40 binary attributes, latent 64, batch 128): one step
    runs; the loss is the batch MEAN, so doubling the batch by repetition (same noise) leaves it unchanged and scales the
    summed loss by two; the step is bit-reproducible."""
    from multivae_amd.data.datasets.base import DatasetOutput

    d = torch.device("cuda:0")
    cfg = dict(L=64, n_attr=40, alpha=0.1, beta=1.0, warmup=10, dists=dict(image="normal", attributes="bernoulli"))
    torch.manual_seed(0)
    model = build_jmvae(cfg, d)
    B = 128
    g = torch.Generator(device=d).manual_seed(2)
    img = torch.rand(B // 2, 3, 64, 64, device=d, generator=g)
    att = (torch.rand(B // 2, 40, device=d, generator=g) > 0.5).float()
    eps = torch.randn(B // 2, 64, device=d, generator=g)

    def run(img_, att_, eps_):
        model.zero_grad(set_to_none=True)
        out = model(DatasetOutput(data=dict(image=img_, attributes=att_)), noise=eps_, epoch=5)
        out.loss.backward()
        torch.cuda.synchronize()
        return float(out.loss), float(out.loss_sum), {k: p.grad.clone() for k, p in model.named_parameters()
                                                       if p.grad is not None}

    l_full, s_full, g_full = run(torch.cat([img, img]), torch.cat([att, att]), torch.cat([eps, eps]))
    l_half, s_half, g_half = run(img, att, eps)
    assert np.isfinite(l_full)
    assert abs(l_full - l_half) <= 1e-5 * abs(l_full), (l_full, l_half)
    assert abs(s_full - 2 * s_half) <= 1e-5 * abs(s_full), (s_full, s_half)
    for k in g_full:  # gradient of the mean: identical for the doubled batch
        e = float((g_full[k] - g_half[k]).abs().max() / g_full[k].abs().max().clamp_min(1e-30))
        assert e <= 1e-4, (k, e)
    l2, s2, g2 = run(torch.cat([img, img]), torch.cat([att, att]), torch.cat([eps, eps]))
    assert l2 == l_full and all(torch.equal(g_full[k], g2[k]) for k in g_full), "the step is not bit-reproducible"
