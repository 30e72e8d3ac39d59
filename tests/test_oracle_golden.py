"""The oracle (oracle/) against the golden vectors generated from the real reference
(tests/golden/make_golden.py).  CPU only.  This is the pin that lets the GPU parity tests trust the oracle."""
import numpy as np
import pytest
import torch

import golden_cases as G
from oracle import elbo, nets

RTOL = 2e-6  # the oracle is the same torch-CPU arithmetic as the reference: expect (near) bit equality


def close(a, b, rtol=RTOL, atol=1e-6):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(b.detach().numpy() if torch.is_tensor(b) else np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    both_inf = torch.isinf(a) & torch.isinf(b) & (a.sign() == b.sign())
    a = torch.where(both_inf, torch.zeros_like(a), a)
    b = torch.where(both_inf, torch.zeros_like(b), b)
    assert torch.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True), float((a - b).abs().max())


def test_unit_base_utils():
    _, a = G.load_case("unit_base_utils")
    mus, lvs, fin = G.t(a["mus"]), G.t(a["lvs"]), G.t(a["fin_lvs"])
    pm, pl = elbo.poe(mus, lvs)
    close(a["poe_mu"], pm)
    close(a["poe_lv"], pl)
    sm, sl = elbo.stable_poe(mus, fin)
    close(a["spoe_mu"], sm)
    close(a["spoe_lv"], sl)
    close(a["kl"], elbo.kl_divergence(mus[0], fin[0], mus[2], fin[2]))
    close(a["z1"], elbo.rsample(mus[0], fin[0], G.t(a["eps1"])))
    close(a["zK"], elbo.rsample(mus[0], fin[0], G.t(a["epsK"])))
    close(a["zKf"], elbo.rsample(mus[0], fin[0], G.t(a["epsKf"])).reshape(-1, 5))
    recon, target = G.t(a["recon"]), G.t(a["target"])
    close(a["lp_normal"], elbo.recon_log_prob("normal", recon, target))
    close(a["lp_normal_s"], elbo.recon_log_prob("normal", recon, target, 0.75))
    close(a["lp_laplace_s"], elbo.recon_log_prob("laplace", recon, target, 0.75))
    close(a["lp_bernoulli"], elbo.recon_log_prob("bernoulli", recon, (target > 0.5).float()))
    close(a["lp_categorical"], elbo.recon_log_prob("categorical", recon, G.t(a["onehot"])))


def _oracle_nets(cfg, dims, sd):
    if cfg["model"] == "MMVAEPlus" or cfg.get("style_dims"):
        return nets.build_default_mlp_multilatent(sd, dims)
    if cfg["arch"] == "tiny" or cfg["model"] in ("MVTCAE", "JMVAE"):
        return nets.build_default_mlp(sd, dims)
    return nets.build_mnist_svhn(sd, cfg["L"])


def _prep(name):
    cfg, a = G.load_case(name)
    dims, data, masks, sd_np = G.build_inputs(cfg)
    sd = {k: G.t(v).clone().requires_grad_(True) for k, v in sd_np.items()}
    data = {m: G.t(v) for m, v in data.items()}
    masks = None if masks is None else {m: G.t(v) for m, v in masks.items()}
    enc_f, dec_f = _oracle_nets(cfg, dims, sd)
    return cfg, a, dims, data, masks, sd, enc_f, dec_f


@pytest.mark.parametrize("name", G.MOPOE_CASES)
def test_mopoe(name):
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    names = cfg["names"]
    assert [k for k, _ in elbo.mopoe_subsets(names)] == cfg["subsets"]
    e = {m: enc_f[m](data[m]) for m in names}
    choice = G.t(a["choice"]) if "choice" in a else None
    o = elbo.mopoe_forward(e, data, dec_f, G.t(a["eps"]), names=names, beta=cfg["beta"],
                           rescale=elbo.rescale_factors(dims, cfg["rescaling"]), dists=cfg["dists"],
                           masks=masks, choice=choice)
    close(a["loss"], o["loss"])
    close(a["loss_sum"], o["loss_sum"], rtol=1e-5)
    for k in ("mus", "logvars", "weights", "joint_mu", "joint_logvar", "z"):
        close(a[k], o[k])
    for k, v in o["metrics"].items():
        close(a["metric/" + k], v)
    o["loss"].backward()
    G.check_grads(a, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()},
                  rtol=1e-5)


def test_mopoe_row_bounds():
    # B=512, S=3 -> [0,170,340,512]
    assert elbo.mopoe_row_bounds(512, 3) == [0, 170, 340, 512]
    assert elbo.mopoe_row_bounds(6, 15) == [0] * 15 + [6]
    assert elbo.mopoe_row_bounds(16, 3) == [0, 5, 10, 16]


@pytest.mark.parametrize("name", G.MVTCAE_CASES)
def test_mvtcae(name):
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    names = cfg["names"]
    e = {m: enc_f[m](data[m]) for m in names}
    o = elbo.mvtcae_forward(e, data, dec_f, G.t(a["eps"]), names=names, alpha=cfg["alpha"], beta=cfg["beta"],
                            rescale=elbo.rescale_factors(dims, cfg["rescaling"]), masks=masks)
    close(a["loss"], o["loss"])
    close(a["loss_sum"], o["loss_sum"])
    for k in ("joint_mu", "joint_logvar", "z"):
        close(a[k], o[k])
    for k, v in o["metrics"].items():
        close(a["metric/" + k], v)
    o["loss"].backward()
    G.check_grads(a, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()},
                  rtol=1e-5)


@pytest.mark.parametrize("name", G.JMVAE_CASES)
def test_jmvae(name):
    """JMVAE.forward (jmvae_model.py:116-192) with the default MultipleHeadJointEncoder, annealing on and off."""
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    names = cfg["names"]
    e = {m: enc_f[m](data[m]) for m in names}
    joint = nets.joint_mlp_encoder(sd, dims, data)
    close(a["joint_mu"], joint[0])
    close(a["joint_logvar"], joint[1])
    o = elbo.jmvae_forward(joint, e, data, dec_f, G.t(a["eps"]), names=names, alpha=cfg["alpha"], beta=cfg["beta"],
                           warmup=cfg["warmup"], epoch=cfg["epoch"],
                           rescale=elbo.rescale_factors(dims, cfg["rescaling"]), dists=cfg.get("dists"))
    close(a["loss"], o["loss"])
    close(a["loss_sum"], o["loss_sum"])
    close(a["z"], o["z"])
    for k, v in o["metrics"].items():
        close(a["metric/" + k], torch.as_tensor(v))
    o["loss"].backward()
    G.check_grads(a, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()},
                  rtol=1e-5)


@pytest.mark.parametrize("name", G.MMVAE_CASES)
def test_mmvae(name):
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    names = cfg["names"]
    mods = [m for m in names if ("noise/" + m) in a]
    plv = G.t(a["prior_log_var"]).clone().requires_grad_(True)
    e = {m: enc_f[m](data[m]) for m in mods}
    noise = {m: G.t(a["noise/" + m]) for m in mods}
    o = elbo.mmvae_forward(e, data, dec_f, noise, names=names, K=cfg["K"], family=cfg["family"], loss=cfg["loss"],
                           prior_log_var=plv, rescale=elbo.rescale_factors(dims, cfg["rescaling"]), masks=masks)
    close(a["loss"], o["loss"])
    for m in mods:
        close(a["lws/" + m], o["lws"][m], rtol=1e-5, atol=1e-4)
        close(a["zs/" + m], o["zs"][m])
    o["loss"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}
    grads["prior_log_var"] = plv.grad
    G.check_grads(a, grads, rtol=2e-5)


def oracle_joint_nll(cfg, a, dims, data, sd, enc_f, dec_f, batch_size_K=None):
    """The oracle's compute_joint_nll for a golden NLL case -> (nll, ll [B], lw [K,B])."""
    names = cfg["names"]
    bsk = cfg["batch_size_K"] if batch_size_K is None else batch_size_K
    noise = G.t(a["noise"])
    with torch.no_grad():
        e = {m: enc_f[m](data[m]) for m in names}
        if cfg["model"] == "MoPoE" and cfg.get("subset") is not None:
            subset = names if cfg["subset"] == "paper" else cfg["subset"]
            return elbo.mopoe_subset_joint_nll(e, data, dec_f, noise, names=names, subset=subset,
                                               dists=cfg.get("dists"), batch_size_K=bsk)
        if cfg["model"] == "MoPoE":
            return elbo.mopoe_joint_nll(e, data, dec_f, noise, names=names, dists=cfg.get("dists"), batch_size_K=bsk)
        if cfg["model"] == "MVTCAE":
            return elbo.mvtcae_joint_nll(e, data, dec_f, noise, names=names, batch_size_K=bsk)
        if cfg["model"] == "JMVAE":
            joint = nets.joint_mlp_encoder(sd, dims, data)
            return elbo.jmvae_joint_nll(joint, data, dec_f, noise, names=names, dists=cfg.get("dists"), batch_size_K=bsk)
        return elbo.mmvae_joint_nll(e, data, dec_f, noise, names=names, sampled=cfg["sampled"], family=cfg["family"],
                                    prior_log_var=G.t(a["prior_log_var"]), dists=cfg.get("dists"), batch_size_K=bsk)


@pytest.mark.parametrize("name", G.NLL_CASES)
def test_joint_nll(name):
    """compute_joint_nll (mopoe_model.py:467-594, mmvae_model.py:365-443, mvtcae_model.py:213-291,
    joint_model.py:82-154): the oracle reproduces the reference's importance-sampled NLL on the recorded noise, and
    the number does not depend on the K-chunk size (the property the HIP path relies on)."""
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    nll, ll, lw = oracle_joint_nll(cfg, a, dims, data, sd, enc_f, dec_f)
    close(a["nll"], nll, rtol=1e-6)
    close(a["ll"], ll, rtol=1e-6)
    close(a["lw"], lw, rtol=1e-6, atol=1e-4)
    nll_one_chunk, _, _ = oracle_joint_nll(cfg, a, dims, data, sd, enc_f, dec_f, batch_size_K=cfg["nll_K"])
    close(a["nll"], nll_one_chunk, rtol=2e-6)


@pytest.mark.parametrize("name", G.NLL_STYLE_CASES)
def test_joint_nll_mopoe_private_latents(name):
    """MoPoE.compute_joint_nll with modality-specific latent spaces (mopoe_model.py:507-521, :543-567)."""
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    names = cfg["names"]
    with torch.no_grad():
        e = {m: enc_f[m](data[m]) for m in names}
        nll, ll, lw = elbo.mopoe_joint_nll(e, data, dec_f, G.t(a["noise"]), names=names, batch_size_K=cfg["batch_size_K"],
                                           style_eps={m: G.t(a["style_eps/" + m]) for m in names})
    close(a["nll"], nll, rtol=1e-6)
    close(a["ll"], ll, rtol=1e-6)


@pytest.mark.parametrize("name", G.COND_NLL_CASES)
def test_cond_nll(name):
    """compute_cond_nll (base_ae_model.py:396-442) restated on the K encodings the reference drew."""
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    with torch.no_grad():
        o = elbo.cond_nll(G.t(a["z"]), data, dec_f, pred_mods=cfg["pred"], dists=cfg.get("dists"))
    for m in cfg["pred"]:
        close(a["cnll/" + m], o[m], rtol=1e-6)


def paper_noises(cfg, a):
    return [{m: G.t(a[f"noise/{c}/{m}"]) for m in cfg["names"]} for c in range(cfg["chunks"])]


@pytest.mark.parametrize("name", G.NLL_PAPER_CASES)
def test_joint_nll_paper_mmvae(name):
    """MMVAE.compute_joint_nll_paper (mmvae_model.py:444-468): chunked, rescaled, batch-summed estimator."""
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    with torch.no_grad():
        e = {m: enc_f[m](data[m]) for m in cfg["names"]}
        nll = elbo.mmvae_joint_nll_paper(e, data, dec_f, paper_noises(cfg, a), names=cfg["names"], K=cfg["nll_K"],
                                         batch_size_K=cfg["batch_size_K"], family=cfg["family"],
                                         prior_log_var=G.t(a["prior_log_var"]),
                                         rescale=elbo.rescale_factors(dims, cfg["rescaling"]))
    close(a["nll"], nll, rtol=1e-6)


def nll_plus_noise(a, kept):
    return {c: {k.split("/")[2]: G.t(a[k]) for k in a if k.startswith(f"noise/{c}/")} for c in kept}


@pytest.mark.parametrize("name", G.NLL_MMVAEPLUS_CASES)
def test_joint_nll_mmvaeplus(name):
    """MMVAEPlus.compute_joint_nll (mmvaePlus_model.py:477-531), including the last modality the reference drops."""
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    names, kept = cfg["names"], cfg["kept"]
    assert kept == names[:-1]
    plv = {k.split("/")[1]: G.t(a[k]) for k in a if k.startswith("prior_logvar/")}
    with torch.no_grad():
        e = {m: enc_f[m](data[m]) for m in names}
        nll, ll = elbo.mmvaeplus_joint_nll(e, data, dec_f, nll_plus_noise(a, kept), names=names, K=cfg["nll_K"],
                                           family=cfg["family"], prior_logvars=plv)
    close(a["nll"], nll, rtol=1e-6)
    close(a["ll"], ll, rtol=1e-6)


@pytest.mark.parametrize("name", G.MOPOE_STYLE_CASES)
def test_mopoe_style(name):
    """MoPoE with modality-specific latent spaces (mopoe_model.py:171-178, :212-221): [shared, style] decoder inputs,
    masked style KLs x beta_style, and the joint_divergence metric that includes them."""
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    names = cfg["names"]
    e = {m: enc_f[m](data[m]) for m in names}
    o = elbo.mopoe_forward(e, data, dec_f, G.t(a["eps"]), names=names, beta=cfg["beta"],
                           rescale=elbo.rescale_factors(dims, cfg["rescaling"]), masks=masks,
                           choice=G.t(a["choice"]) if "choice" in a else None,
                           style_eps={m: G.t(a["style_eps/" + m]) for m in names}, beta_style=cfg["beta_style"])
    close(a["loss"], o["loss"])
    close(a["loss_sum"], o["loss_sum"])
    for k, v in o["metrics"].items():
        close(a["metric/" + k], v)
    for m in names:
        close(a["w/" + m], o["ws"][m])
    o["loss"].backward()
    G.check_grads(a, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}, rtol=1e-5)


@pytest.mark.parametrize("name", G.CRMVAE_CASES)
def test_crmvae(name):
    """CRMVAE.forward (crmvae_model.py:37-105): PoE joint, KL(joint || prior) + masked KL(joint || q_m), reconstructions
    from the joint and from the unimodal samples."""
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    names = cfg["names"]
    e = {m: enc_f[m](data[m]) for m in names}
    o = elbo.crmvae_forward(e, data, dec_f, G.t(a["eps"]), {m: G.t(a["mod_eps/" + m]) for m in names}, names=names,
                            beta=cfg["beta"], rescale=elbo.rescale_factors(dims, cfg["rescaling"]),
                            dists=cfg.get("dists"), masks=masks)
    close(a["loss"], o["loss"])
    close(a["loss_sum"], o["loss_sum"])
    assert {k[7:] for k in a if k.startswith("metric/")} == set(o["metrics"])
    for k, v in o["metrics"].items():
        close(a["metric/" + k], v)
    o["loss"].backward()
    G.check_grads(a, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}, rtol=1e-5)


def dmvae_noise(a, names):
    return {"shared": G.t(a["noise/shared"]), "private": {m: G.t(a["noise/private/" + m]) for m in names}}


@pytest.mark.parametrize("name", G.DMVAE_CASES)
def test_dmvae(name):
    """DMVAE.forward (dmvae_model.py:152-240): joint + unimodal ELBOs with private latents, private betas, masks."""
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    names = cfg["names"]
    e = {m: enc_f[m](data[m]) for m in names}
    o = elbo.dmvae_forward(e, data, dec_f, dmvae_noise(a, names), names=names, beta=cfg["beta"],
                           private_betas=cfg.get("private_betas"), rescale=elbo.rescale_factors(dims, cfg["rescaling"]),
                           dists=cfg.get("dists"), masks=masks)
    close(a["loss"], o["loss"])
    close(a["joint_mu"], o["joint_mu"])
    for k, v in o["metrics"].items():
        close(a["metric/" + k], v)
    o["loss"].backward()
    G.check_grads(a, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}, rtol=1e-5)


def oracle_mvae(cfg, a, dims, data, masks, enc_f, dec_f):
    names = cfg["names"]
    e = {m: enc_f[m](data[m]) for m in names}
    beta = elbo.mvae_annealing(cfg["epoch"], cfg["batch_ratio"], cfg["warmup"], cfg["beta"])
    return elbo.mvae_forward(e, data, dec_f, G.t(a["eps"]), names=names, subsets=cfg["subsets"], beta=beta,
                             rescale=elbo.rescale_factors(dims, cfg["rescaling"]), dists=cfg.get("dists"), masks=masks)


@pytest.mark.parametrize("name", G.MVAE_CASES)
def test_mvae(name):
    """MVAE.forward (mvae_model.py:145-228): joint / unimodal / random subsets, annealing, incomplete data, and
    compute_joint_nll (:266-340) where the case holds one."""
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    assert cfg["subsets"] == elbo.mvae_subsets(cfg["names"], cfg["subsampling"], cfg["subsets"][1 + len(cfg["names"]):]
                                               if cfg["subsampling"] else ())
    o = oracle_mvae(cfg, a, dims, data, masks, enc_f, dec_f)
    close(a["loss"], o["loss"])
    close(a["loss_sum"], torch.as_tensor(o["loss_sum"]))
    assert {k[7:] for k in a if k.startswith("metric/")} == set(o["metrics"])
    for k, v in o["metrics"].items():
        close(a["metric/" + k], torch.as_tensor(v))
    o["loss"].backward()
    G.check_grads(a, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}, rtol=1e-5)
    if cfg["nll_K"]:
        with torch.no_grad():
            e = {m: enc_f[m](data[m]) for m in cfg["names"]}
            nll, ll, _ = elbo.mvae_joint_nll(e, data, dec_f, G.t(a["nll_noise"]), names=cfg["names"],
                                             dists=cfg.get("dists"), batch_size_K=4)
        close(a["nll"], nll, rtol=1e-6)
        close(a["nll_ll"], ll, rtol=1e-6)


def mmvaeplus_noise(a, mods):
    return {c: {k.split("/")[2]: G.t(v) for k, v in a.items() if k.startswith(f"noise/{c}/")} for c in mods}


@pytest.mark.parametrize("name", G.MMVAEPLUS_CASES)
def test_mmvaeplus(name):
    """MMVAEPlus.forward (mmvaePlus_model.py:122-360): shared + private latents, prior-sampled private latents for the
    cross reconstructions, beta, IWAE / DReG, the three posterior families, incomplete data."""
    cfg, a, dims, data, masks, sd, enc_f, dec_f = _prep(name)
    names = cfg["names"]
    mods = [m for m in names if ("lws/" + m) in a]
    plv = {k.split("/")[1]: G.t(v).clone().requires_grad_(True) for k, v in a.items() if k.startswith("prior_logvar/")}
    e = {m: enc_f[m](data[m]) for m in mods}
    o = elbo.mmvaeplus_forward(e, data, dec_f, mmvaeplus_noise(a, mods), names=names, K=cfg["K"], family=cfg["family"],
                               loss=cfg["loss"], beta=cfg["beta"], prior_logvars=plv,
                               rescale=elbo.rescale_factors(dims, cfg["rescaling"]), masks=masks)
    close(a["loss"], o["loss"])
    for m in mods:
        close(a["lws/" + m], o["lws"][m], rtol=1e-5, atol=1e-4)
        close(a["us/" + m], o["us"][m])
        close(a["ws/" + m], o["ws"][m])
    o["loss"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}
    for k, v in plv.items():
        if k != "shared" or cfg["learn_shared_prior"]:
            grads["logvars_priors." + k] = v.grad if v.grad is not None else torch.zeros_like(v)
    G.check_grads(a, grads, rtol=2e-5)


def resnet_case():
    """Inputs of the resnet_mmnist_nets golden: (cfg, arrays, encoder sd, decoder sd, x, z, projections)."""
    cfg, a = G.load_case("resnet_mmnist_nets")
    P = G.P
    seed, B, K, pd_, sd_ = cfg["seed"], cfg["B"], cfg["K"], cfg["private_dim"], cfg["shared_dim"]
    L = pd_ + sd_
    esd = P.make_state_dict(P.mmnist_resnet_encoder_shapes(pd_, sd_), seed)
    dsd = P.make_state_dict(P.mmnist_resnet_decoder_shapes(L), seed + 1)
    x = G.t(P.uniform((B, 3, 28, 28), seed + 2))
    z = G.t(P.uniform((K, B, L), seed + 3, -1.0, 1.0))
    pe = [G.t(P.uniform((B, d), seed + 10 + i, -1.0, 1.0)) for i, d in enumerate((sd_, sd_, pd_, pd_))]
    pdec = G.t(P.uniform((K, B, 3, 28, 28), seed + 20, -1.0, 1.0))
    return cfg, a, esd, dsd, x, z, pe, pdec


def test_resnet_mmnist_nets():
    """PolyMNIST ResNet encoder / decoder (mmnist.py:214-366): outputs and gradients vs the reference's."""
    cfg, a, esd, dsd, x, z, pe, pdec = resnet_case()
    oe = {k: G.t(v).clone().requires_grad_(True) for k, v in esd.items()}
    od = {k: G.t(v).clone().requires_grad_(True) for k, v in dsd.items()}
    outs = nets.mmnist_resnet_encoder(oe, "", x)
    for k, o in zip(("mu_u", "lv_u", "mu_w", "lv_w"), outs):
        close(a[k], o)
    sum((o * p).sum() for o, p in zip(outs, pe)).backward()
    zz = z.clone().requires_grad_(True)
    rec = nets.mmnist_resnet_decoder(od, "", zz)
    close(a["recon_sample"], rec.reshape(-1)[G.P.hash_indices(rec.numel(), 512, 77)])
    (rec * pdec).sum().backward()
    close(a["dz"], zz.grad, rtol=1e-5, atol=1e-5)
    grads = {"enc." + k: v.grad for k, v in oe.items()}
    grads.update({"dec." + k: v.grad for k, v in od.items()})
    G.check_grads(a, grads, rtol=1e-5)


def resnet_cub_case():
    cfg, a = G.load_case("resnet_cub_nets")
    P = G.P
    seed, B, L = cfg["seed"], cfg["B"], cfg["L"]
    esd = P.make_state_dict(P.cub_resnet_encoder_shapes(L), seed)
    dsd = P.make_state_dict(P.cub_resnet_decoder_shapes(L), seed + 1)
    x = G.t(P.uniform((B, 3, 64, 64), seed + 2))
    z = G.t(P.uniform((B, L), seed + 3, -1.0, 1.0))
    pe = [G.t(P.uniform((B, L), seed + 10 + i, -1.0, 1.0)) for i in range(2)]
    pdec = G.t(P.uniform((B, 3, 64, 64), seed + 20, -1.0, 1.0))
    return cfg, a, esd, dsd, x, z, pe, pdec


def test_resnet_cub_nets():
    """CUB_Resnet_Encoder / Decoder (cub.py:144-293, pre-activation ResnetBlock): outputs and gradients."""
    cfg, a, esd, dsd, x, z, pe, pdec = resnet_cub_case()
    oe = {k: G.t(v).clone().requires_grad_(True) for k, v in esd.items()}
    od = {k: G.t(v).clone().requires_grad_(True) for k, v in dsd.items()}
    outs = nets.cub_resnet_encoder(oe, "", x)
    close(a["mu"], outs[0])
    close(a["lv"], outs[1])
    sum((o * p).sum() for o, p in zip(outs, pe)).backward()
    zz = z.clone().requires_grad_(True)
    rec = nets.cub_resnet_decoder(od, "", zz)
    close(a["recon_sample"], rec.reshape(-1)[G.P.hash_indices(rec.numel(), 512, 78)])
    (rec * pdec).sum().backward()
    close(a["dz"], zz.grad, rtol=1e-5, atol=1e-5)
    grads = {"enc." + k: v.grad for k, v in oe.items()}
    grads.update({"dec." + k: v.grad for k, v in od.items()})
    G.check_grads(a, grads, rtol=1e-5)


def test_numpy_conv_pins_match_torch():
    """conv2d_np / conv_transpose2d_np (independent numpy restatements) agree with the aten ops the
    reference calls, on the SVHN layer shapes."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 8, 8, generator=g)
    w = torch.randn(5, 3, 4, 4, generator=g)
    b = torch.randn(5, generator=g)
    close(nets.conv2d_np(x, w, b, 2, 1), F.conv2d(x, w, b, 2, 1), rtol=1e-5, atol=1e-5)
    close(nets.conv2d_np(x[:, :, :4, :4], w, b, 2, 0), F.conv2d(x[:, :, :4, :4], w, b, 2, 0), rtol=1e-5, atol=1e-5)
    wt = torch.randn(3, 6, 4, 4, generator=g)
    bt = torch.randn(6, generator=g)
    close(nets.conv_transpose2d_np(x, wt, bt, 2, 1), F.conv_transpose2d(x, wt, bt, 2, 1), rtol=1e-5, atol=1e-5)
    z = torch.randn(4, 3, 1, 1, generator=g)
    close(nets.conv_transpose2d_np(z, wt, bt, 1, 0), F.conv_transpose2d(z, wt, bt, 1, 0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", G.NLL_DMVAE_CASES)
def test_dmvae_joint_nll(name):
    """DMVAE.compute_joint_nll of the reference (dmvae_model.py:311-412): its running ln_prior / ln_posterior sums over the
    K-chunks and the data points are part of the number."""
    cfg, a = G.load_case(name)
    dims, data, _, sd_np = G.build_inputs(cfg)
    sd = {k: G.t(v) for k, v in sd_np.items()}
    enc_f, dec_f = nets.build_default_mlp_multilatent(sd, dims)
    names = cfg["names"]
    tdata = {m: G.t(v) for m, v in data.items()}
    noise = {"shared": G.t(a["noise/shared"]), "private": {m: G.t(a["noise/private/" + m]) for m in names}}
    with torch.no_grad():
        e = {m: enc_f[m](tdata[m]) for m in names}
        nll, ll = elbo.dmvae_joint_nll(e, tdata, dec_f, noise, names=names, batch_size_K=cfg["batch_size_K"],
                                       dists=cfg.get("dists"))
    close(a["nll"], nll)
    close(a["ll"], ll)


def test_relu_site_recorder_and_forcing_reproduce_the_oracle():
    """tests/relu_sites.py: recording the rectifier sites does not change the oracle, and forcing the oracle's OWN decisions
    reproduces its loss and gradients bit for bit (the GPU parity tests force the HIP path's decisions on the units within
    2e-6 of zero instead); a dictated decision takes effect in the network it belongs to and upstream of it, nowhere else."""
    import relu_sites as RS

    cfg, a = G.load_case("mopoe_mnistsvhn_k10")
    dims, data, masks, sd_np = G.build_inputs(cfg)

    def run():
        sd = {k: G.t(v).clone().requires_grad_(True) for k, v in sd_np.items()}
        enc_f, dec_f = nets.build_mnist_svhn(sd, cfg["L"])
        td = {m: G.t(v) for m, v in data.items()}
        e = {m: enc_f[m](td[m]) for m in cfg["names"]}
        o = elbo.mopoe_forward(e, td, dec_f, G.t(a["eps"]), names=cfg["names"], beta=cfg["beta"],
                               rescale=elbo.rescale_factors(dims, cfg["rescaling"]), dists=cfg["dists"], masks=None, choice=None)
        o["loss"].backward()
        return o["loss"].detach(), {k: v.grad.clone() for k, v in sd.items()}

    loss0, g0 = run()
    sites = RS.OracleSites()
    with sites.record():
        loss1, g1 = run()
    assert torch.equal(loss0, loss1) and all(torch.equal(g0[k], g1[k]) for k in g0)
    assert set(sites.pre) == {("encoders.mnist.", 0), ("encoders.mnist.", 1), ("encoders.svhn.", 0), ("encoders.svhn.", 1),
                              ("encoders.svhn.", 2), ("decoders.mnist.", 0), ("decoders.svhn.", 0), ("decoders.svhn.", 1),
                              ("decoders.svhn.", 2)}
    own = {k: [p > 0 for p in v] for k, v in sites.pre.items()}
    with sites.force(own):
        loss2, g2 = run()
    assert torch.equal(loss0, loss2) and all(torch.equal(g0[k], g2[k]) for k in g0)
    # switch one active unit of the svhn decoder's second layer off: the other decoder keeps its gradients exactly
    m = own[("decoders.svhn.", 1)][0]
    idx = tuple(int(i) for i in torch.nonzero(m)[0])
    m[idx] = False
    with sites.force(own):
        _, g3 = run()
    assert torch.equal(g0["decoders.mnist.layers.1.0.weight"], g3["decoders.mnist.layers.1.0.weight"])
    assert not torch.equal(g0["decoders.svhn.dec.2.weight"], g3["decoders.svhn.dec.2.weight"])
    assert not torch.equal(g0["encoders.mnist.layers.0.0.weight"], g3["encoders.mnist.layers.0.0.weight"])
