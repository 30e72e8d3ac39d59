"""Oracle: posterior aggregation + reparameterisation + KL / IWAE + reconstruction NLL.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain torch-CPU restatement of
  /root/reference/src/multivae/models/base/base_utils.py   (poe, stable_poe, kl_divergence, rsample, log-probs)
  /root/reference/src/multivae/models/mopoe/mopoe_model.py (MoPoE.forward / inference / selection / divergence)
  /root/reference/src/multivae/models/mvtcae/mvtcae_model.py (MVTCAE.forward)
  /root/reference/src/multivae/models/mmvae/mmvae_model.py (MMVAE.forward / compute_k_lws / loosers)
  /root/reference/src/multivae/models/jmvae/jmvae_model.py (JMVAE.forward)
Noise (`eps`, `choice`, `u`) is always an explicit argument.
"""
import math
from itertools import chain, combinations

import torch
import torch.nn.functional as F

HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------------------------
# a1-a5: helpers of base_utils.py
# ----------------------------------------------------------------------------------------------
def poe(mus, logvars, eps=1e-8):
    """Gaussian product of experts over dim 0.  base_utils.py:122-130.

    The 1e-8 is added to every expert's variance, also for single-expert stacks.
    """
    var = torch.exp(logvars) + eps
    T = 1.0 / var
    pd_mu = torch.sum(mus * T, dim=0) / torch.sum(T, dim=0)
    pd_var = 1.0 / torch.sum(T, dim=0)
    return pd_mu, torch.log(pd_var)


def stable_poe(mus, logvars):
    """log-sum-exp PoE without the variance epsilon.  base_utils.py:133-147."""
    if len(mus) == 1:
        return mus[0], logvars[0]
    ln_inv = torch.stack([-l for l in logvars])
    ln_var = -torch.logsumexp(ln_inv, dim=0)
    mu = (torch.exp(ln_inv) * torch.stack(list(mus))).sum(dim=0) * torch.exp(ln_var)
    return mu, ln_var


def kl_divergence(mean, log_var, prior_mean, prior_log_var):
    """General diagonal-Gaussian KL summed over the last dim.  base_utils.py:90-119."""
    kl = 0.5 * (
        prior_log_var
        - log_var
        + torch.exp(log_var - prior_log_var)
        + ((mean - prior_mean) ** 2) / torch.exp(prior_log_var)
        - 1
    )
    return kl.sum(dim=-1)


def rsample(mu, log_var, eps):
    """z = mu + exp(0.5*log_var) * eps.  base_utils.py:150-172 with the N(0,1) draw made explicit.

    eps is [B,L] (N == 1) or [K,B,L] (N == K; `Normal.rsample([K])` broadcasts the same way).
    """
    return mu + torch.exp(0.5 * log_var) * eps


def recon_log_prob(dist_name, recon, target, scale=1.0):
    """Elementwise decoder log-probability.  base_utils.py:62-87 (torch.distributions formulas)."""
    if dist_name == "normal":
        # Normal(recon, scale).log_prob(target)
        return -((target - recon) ** 2) / (2.0 * scale * scale) - math.log(scale) - HALF_LOG_2PI
    if dist_name == "laplace":
        # Laplace(recon, scale).log_prob(target)
        return -math.log(2.0 * scale) - torch.abs(target - recon) / scale
    if dist_name == "bernoulli":
        # Bernoulli(logits=recon).log_prob(target) = -BCEWithLogits
        return -F.binary_cross_entropy_with_logits(recon, target.expand_as(recon), reduction="none")
    if dist_name == "categorical":
        # base_utils.py:28-40: target * log_softmax(recon + 1e-6)
        return target * F.log_softmax(recon + 1e-6, dim=-1)
    raise ValueError(dist_name)


def rescale_factors(input_dims, uses_likelihood_rescaling, given=None):
    """base_ae_model.py:127-152."""
    if not uses_likelihood_rescaling:
        return {k: 1.0 for k in input_dims}
    if given is not None:
        return dict(given)
    prods = {k: float(math.prod(v)) for k, v in input_dims.items()}
    mx = max(prods.values())
    return {k: mx / prods[k] for k in prods}


def _row_nll(dist_name, recon, x, rescale, scale=1.0):
    """-(log p(x|recon) * rescale) summed over everything but the leading batch dims of `x`.

    recon is [B,*D] or [K,B,*D]; returns [B] or [K,B].
    """
    lp = recon_log_prob(dist_name, recon, x, scale)
    lead = recon.dim() - (x.dim() - 1)
    return (-lp * rescale).reshape(*recon.shape[:lead], -1).sum(-1)


# ----------------------------------------------------------------------------------------------
# a6-a10: MoPoE
# ----------------------------------------------------------------------------------------------
def mopoe_subsets(names):
    """Non-empty subsets in the reference's enumeration order.  mopoe_model.py:71-106, :291-292.

    `names` is the encoder-dict order.  Subsets come by increasing size, `itertools.combinations`
    order; inside a subset the experts are in sorted-name order; key = "_".join(sorted(subset)).
    """
    names = list(names)
    out = []
    for combo in chain.from_iterable(combinations(names, n) for n in range(len(names) + 1)):
        if len(combo) == 0:
            continue  # the "" key exists in model.subsets but is skipped at :292
        mods = sorted(combo)
        out.append(("_".join(mods), mods))
    return out


def mopoe_row_bounds(B, S):
    """Row-range assignment of subsets.  mopoe_model.py:435-465 with w = 1/S.

    n = int(floor(B * (1/S))) computed in float32 like the reference's tensor arithmetic.
    """
    w = torch.tensor(1.0 / float(S), dtype=torch.float32)
    n = int(torch.floor(B * w))
    bounds = [0]
    for k in range(S):
        bounds.append(B if k == S - 1 else bounds[-1] + n)
    bounds[-1] = B
    return bounds


def mopoe_inference(enc, names, masks=None, choice=None, subsets=None):
    """mopoe_model.py:274-350.

    enc: {name: (mu[B,L], logvar[B,L])}.  Returns dict(mus[S,B,L], logvars[S,B,L], weights[S,B],
    joint_mu, joint_logvar, keys).  `choice` [B,S] one-hot (float/bool) replaces the
    OneHotCategorical draw of :417-433 for masked inputs.
    """
    subsets = mopoe_subsets(names) if subsets is None else subsets
    M = len(names)
    mus, lvs, avail = [], [], []
    for key, mods in subsets:
        smu = torch.stack([enc[m][0] for m in mods])
        slv = torch.stack([enc[m][1] for m in mods])
        if smu.shape[0] == M:  # prior expert only on the full subset (:249-262)
            smu = torch.cat([smu, torch.zeros_like(smu[:1])], 0)
            slv = torch.cat([slv, torch.zeros_like(slv[:1])], 0)
        mu_s, lv_s = poe(smu, slv)
        mus.append(mu_s)
        lvs.append(lv_s)
        if masks is not None:
            f = torch.ones_like(masks[mods[0]], dtype=torch.bool)
            for m in mods:
                f = torch.logical_and(f, masks[m].bool())
            avail.append(f)
    mus = torch.stack(mus)
    lvs = torch.stack(lvs)
    S, B = mus.shape[0], mus.shape[1]
    if masks is not None:
        a = torch.stack(avail).to(mus.dtype)
        weights = a / a.sum(0)
        sel = choice.bool()  # [B,S]
        jm = mus.permute(1, 0, 2)[sel]
        jl = lvs.permute(1, 0, 2)[sel]
    else:
        weights = torch.full((S, B), 1.0 / float(S), dtype=mus.dtype)
        bnd = mopoe_row_bounds(B, S)
        jm = torch.cat([mus[k, bnd[k] : bnd[k + 1]] for k in range(S)])
        jl = torch.cat([lvs[k, bnd[k] : bnd[k + 1]] for k in range(S)])
    return dict(mus=mus, logvars=lvs, weights=weights, joint_mu=jm, joint_logvar=jl,
                keys=[k for k, _ in subsets])


def mopoe_joint_divergence(mus, logvars, weights):
    """mopoe_model.py:108-145: mean_b sum_s w[s,b] * KL(N(mu_s, var_s) || N(0, I))."""
    klds = -0.5 * (1 - logvars.exp() - mus.pow(2) + logvars).sum(-1)  # [S,B]
    return (weights * klds).sum(dim=0).mean(), klds


def mopoe_forward(enc, data, decoders, eps, *, names, beta=1.0, rescale=None, dists=None,
                  dist_scales=None, masks=None, choice=None, subsets=None, style_eps=None, beta_style=1.0):
    """MoPoE.forward.  mopoe_model.py:147-227 (Appendix A.1).

    eps [B,L] reproduces the reference exactly; eps [K,B,L] is the K-sample Monte-Carlo extension
    of SURVEY.md §0 D1: the reconstruction term is averaged over k, the analytic KL is unchanged.
    decoders: {name: callable(z[...,L]) -> recon[...,*D]}.
    Modality-specific latent spaces (:171-178, :212-221): enc[m] = (mu, lv, style_mu, style_lv), style_eps[m] [B,S_m]
    (drawn per modality inside the decoder loop, after the shared noise); the decoders see [z, w_m]; the style KLs
    (masked, averaged over the whole batch, x beta_style) are added IN PLACE to the tensor that is also the
    "joint_divergence" metric (:164, :221), so the metric includes them.
    """
    rescale = rescale or {m: 1.0 for m in names}
    dists = dists or {m: "normal" for m in names}
    dist_scales = dist_scales or {}
    inf = mopoe_inference(enc, names, masks=masks, choice=choice, subsets=subsets)
    z = rsample(inf["joint_mu"], inf["joint_logvar"], eps)
    B = inf["joint_mu"].shape[0]
    kld, klds = mopoe_joint_divergence(inf["mus"], inf["logvars"], inf["weights"])
    metrics = {"joint_divergence": kld}
    rows = {}
    loss = 0
    ws = {}
    for m in names:
        z_m = z
        if style_eps is not None:
            smu, slv = enc[m][2], enc[m][3]
            ws[m] = rsample(smu, slv, style_eps[m])
            z_m = torch.cat([z.expand(*ws[m].shape[:-1], z.shape[-1]) if z.dim() != ws[m].dim() else z, ws[m]], dim=-1)
        recon = decoders[m](z_m)
        r = _row_nll(dists[m], recon, data[m], rescale[m], dist_scales.get(m, 1.0))  # [B] or [K,B]
        rows[m] = r
        if r.dim() == 2:
            r = r.mean(0)
        if masks is not None:
            r = r * masks[m].to(r.dtype)
        metrics["recon_" + m] = r.mean()
        loss = loss + metrics["recon_" + m]
        if style_eps is not None:
            skl = -0.5 * (1 - slv.exp() - smu.pow(2) + slv).reshape(smu.size(0), -1).sum(-1)
            if masks is not None:
                skl = skl * masks[m].to(skl.dtype)
            kld = kld + skl.mean() * beta_style
    metrics["joint_divergence"] = kld
    loss = loss + beta * kld
    out = dict(inf)
    out["ws"] = ws
    out.update(loss=loss, loss_sum=loss * B, metrics=metrics, z=z, klds=klds, rows=rows)
    return out


# ----------------------------------------------------------------------------------------------
# a14: MVTCAE
# ----------------------------------------------------------------------------------------------
def mvtcae_forward(enc, data, decoders, eps, *, names, alpha=0.1, beta=2.5, rescale=None, dists=None,
                   dist_scales=None, masks=None):
    """MVTCAE.forward.  mvtcae_model.py:42-169 (Appendix A.3).  Metrics are SUMS over the batch."""
    rescale = rescale or {m: 1.0 for m in names}
    dists = dists or {m: "normal" for m in names}
    dist_scales = dist_scales or {}
    M = len(names)
    mus = torch.stack([enc[m][0] for m in names])
    lvs = []
    for m in names:
        lv = enc[m][1]
        if masks is not None:  # :128-129  missing rows get +inf log-variance
            lv = torch.where(masks[m].bool().unsqueeze(-1), lv, torch.full_like(lv, float("inf")))
        lvs.append(lv)
    lvs = torch.stack(lvs)
    jmu, jlv = poe(mus, lvs)  # no prior expert (:163)
    z = rsample(jmu, jlv, eps)
    B = jmu.shape[0]
    joint_kld = -0.5 * torch.sum(1 - jlv.exp() - jmu.pow(2) + jlv)
    metrics = {"joint_divergence": joint_kld}
    loss_rec = 0
    rows = {}
    for m in names:
        recon = decoders[m](z)
        r = _row_nll(dists[m], recon, data[m], rescale[m], dist_scales.get(m, 1.0))
        rows[m] = r
        if r.dim() == 2:
            r = r.mean(0)
        if masks is not None:
            r = masks[m].to(r.dtype) * r
        metrics[m] = r.sum()
        loss_rec = loss_rec + r.sum()
    kld_losses = 0.0
    for i, m in enumerate(names):
        mu, lv = mus[i], lvs[i]
        k = -0.5 * (1 - jlv.exp() / lv.exp() - (jmu - mu).pow(2) / lv.exp() + jlv - lv)
        k = k.reshape(B, -1).sum(-1)
        if masks is not None:
            k = torch.where(masks[m].bool(), k, torch.zeros_like(k))
        metrics["kld_" + m] = k.sum()
        kld_losses = kld_losses + k.sum()
    rec_w = (M - alpha) / M
    cvib_w = alpha / M
    vib_w = 1 - alpha
    total = rec_w * loss_rec + beta * (cvib_w * kld_losses + vib_w * joint_kld)
    return dict(loss=total / B, loss_sum=total, metrics=metrics, joint_mu=jmu, joint_logvar=jlv, z=z,
                rows=rows)


# ----------------------------------------------------------------------------------------------
# a11-a13: MMVAE
# ----------------------------------------------------------------------------------------------
def mmvae_std(log_var, family):
    """mmvae_model.py:66-74; mmvaePlus_model.py:110-120 adds 'normal_with_softplus'."""
    if family == "laplace_with_softmax":
        return F.softmax(log_var, dim=-1) * log_var.size(-1) + 1e-6
    if family == "normal_with_softplus":
        return F.softplus(log_var) + 1e-6
    return torch.exp(0.5 * log_var)


def _latent_family(family):
    """Distribution family of the latent for the log-prob / rsample helpers."""
    return "laplace_with_softmax" if family == "laplace_with_softmax" else "normal"


def latent_log_prob(family, z, loc, scale):
    """torch.distributions Normal / Laplace log_prob on the latent."""
    if family == "normal":
        return -((z - loc) ** 2) / (2 * scale * scale) - torch.log(scale) - HALF_LOG_2PI
    return -torch.log(2 * scale) - torch.abs(z - loc) / scale


def latent_rsample(family, loc, scale, noise):
    """Normal.rsample: loc + scale*eps.  Laplace.rsample: u ~ U(eps_f32-1, 1);
    loc - scale * sign(u) * log1p(-|u|)  (SURVEY.md Appendix B).  noise is [K,B,L]."""
    if family == "normal":
        return loc + scale * noise
    return loc - scale * noise.sign() * torch.log1p(-noise.abs())


def mmvae_forward(enc, data, decoders, noise, *, names, K, family="laplace_with_softmax",
                  loss="dreg_looser", prior_mean=None, prior_log_var=None, rescale=None, dists=None,
                  dist_scales=None, masks=None):
    """MMVAE.forward + compute_k_lws + iwae_looser / dreg_looser.  mmvae_model.py:95-292 (A.2).

    enc: {name: (mu, log_var)} for the modalities present; noise: {name: [K,B,L]}.
    The DReG gradient hook (:263-266) is reproduced with `Tensor.register_hook` on z.
    """
    mods = [m for m in names if m in data]
    rescale = rescale or {m: 1.0 for m in names}
    dists = dists or {m: "normal" for m in names}
    dist_scales = dist_scales or {}
    L = enc[mods[0]][0].shape[-1]
    ref = enc[mods[0]][0]
    if prior_mean is None:
        prior_mean = torch.zeros(1, L, dtype=ref.dtype)
    if prior_log_var is None:
        prior_log_var = torch.zeros(1, L, dtype=ref.dtype)
    p_std = mmvae_std(prior_log_var, family)
    post, post_det, zs, recons = {}, {}, {}, {}
    for c in mods:
        mu, lv = enc[c]
        sig = mmvae_std(lv, family)
        z = latent_rsample(family, mu, sig, noise[c])  # [K,B,L]
        post[c] = (mu, sig)
        post_det[c] = (mu.detach(), sig.detach())
        zs[c] = z
        flat = z.reshape(-1, L)
        recons[c] = {}
        for r in mods:
            rec = decoders[r](flat)
            recons[c][r] = rec.reshape(*z.shape[:-1], *rec.shape[1:])
    q = post_det if loss == "dreg_looser" else post
    if masks is not None:
        n_avail = torch.stack([masks[m] for m in masks]).int().sum(0)
    else:
        n_avail = torch.tensor([len(names)])
    lws = {}
    for c in mods:
        z = zs[c]
        lpz = latent_log_prob(family, z, prior_mean, p_std).sum(-1)
        lq = []
        for m in mods:
            v = latent_log_prob(family, z, q[m][0], q[m][1]).sum(-1)
            if masks is not None:
                v = torch.where(masks[m].bool().unsqueeze(0), v, torch.full_like(v, -float("inf")))
            lq.append(v)
        lq = torch.logsumexp(torch.stack(lq), dim=0) - torch.log(n_avail.to(z.dtype))
        lpx = 0
        for r in mods:
            lp = recon_log_prob(dists[r], recons[c][r], data[r], dist_scales.get(r, 1.0))
            lp = lp.reshape(z.shape[0], z.shape[1], -1).mul(rescale[r]).sum(-1)
            if masks is not None:
                lp = lp * masks[r].to(lp.dtype)
            lpx = lpx + lp
        lw = lpx + lpz - lq
        if masks is not None:
            lw = lw * masks[c].to(lw.dtype)
        lws[c] = lw
    if loss == "dreg_looser":
        wk = {}
        with torch.no_grad():
            for c in mods:
                wk[c] = (lws[c] - torch.logsumexp(lws[c], 0, keepdim=True)).exp()
        tot = torch.stack([lws[c] * wk[c] for c in mods]).sum(1)
        for c in mods:
            if zs[c].requires_grad:
                zs[c].register_hook(lambda g, w=wk[c]: w.unsqueeze(-1) * g)
    else:
        tot = torch.logsumexp(torch.stack([lws[c] for c in mods]), dim=1) - math.log(K)
    tot = tot.sum(0) / n_avail.to(tot.dtype)
    loss_v = -tot.sum()
    return dict(loss=loss_v, loss_sum=loss_v, metrics={}, lws=lws, zs=zs)


# ----------------------------------------------------------------------------------------------
# a15: MMVAE+ (shared latent u, one private latent w per modality)
# ----------------------------------------------------------------------------------------------
def mmvaeplus_forward(enc, data, decoders, noise, *, names, K, family="laplace_with_softmax", loss="dreg_looser",
                      beta=1.0, prior_logvars=None, rescale=None, dists=None, dist_scales=None, masks=None):
    """MMVAEPlus.forward: _compute_posteriors_and_embeddings + _compute_k_lws + iwae / dreg looser.
    mmvaePlus_model.py:122-360.

    enc: {name: (mu, lv, mu_style, lv_style)}; noise: {cond: {"u": [K,B,L], "w": [K,B,S], recon != cond: [K,B,S]}} in the
    reference's draw order (u, w, then one prior draw per other modality); prior_logvars: {"shared": [1,L+S],
    name: [1,S]} (the prior means are fixed zeros)."""
    mods = [m for m in names if m in data]
    rescale = rescale or {m: 1.0 for m in names}
    dists = dists or {m: "normal" for m in names}
    dist_scales = dist_scales or {}
    fam = _latent_family(family)
    L = enc[mods[0]][0].shape[-1]
    S = enc[mods[0]][2].shape[-1]
    ref = enc[mods[0]][0]
    if prior_logvars is None:
        prior_logvars = {"shared": torch.zeros(1, L + S, dtype=ref.dtype)}
        prior_logvars.update({m: torch.zeros(1, S, dtype=ref.dtype) for m in names})
    post, post_det, emb, recons = {}, {}, {}, {}
    for c in mods:
        mu, lv, mus, lvs = enc[c]
        sig, sigs = mmvae_std(lv, family), mmvae_std(lvs, family)
        u = latent_rsample(fam, mu, sig, noise[c]["u"])
        w = latent_rsample(fam, mus, sigs, noise[c]["w"])
        post[c] = ((mu, sig), (mus, sigs))
        post_det[c] = ((mu.detach(), sig.detach()), (mus.detach(), sigs.detach()))
        emb[c] = (u, w)
        recons[c] = {}
        for r in mods:
            if r == c:
                z = torch.cat([u, w], dim=-1)
            else:
                p_sig = mmvae_std(prior_logvars[r], family).expand(mu.shape[0], S)
                wp = latent_rsample(fam, torch.zeros_like(p_sig), p_sig, noise[c][r])
                z = torch.cat([u, wp], dim=-1)
            rec = decoders[r](z.reshape(-1, z.shape[-1]))
            recons[c][r] = rec.reshape(*z.shape[:-1], *rec.shape[1:])
    q = post_det if loss == "dreg_looser" else post
    if masks is not None:
        n_avail = torch.stack([masks[m] for m in masks]).int().sum(0)
    else:
        n_avail = torch.tensor([len(names)])
    pz_std = mmvae_std(prior_logvars["shared"], family)
    lws = {}
    for c in mods:
        u, w = emb[c]
        z = torch.cat([u, w], dim=-1)
        lpz = latent_log_prob(fam, z, torch.zeros_like(pz_std), pz_std).sum(-1)
        lqu = []
        for m in mods:
            v = latent_log_prob(fam, u, q[m][0][0], q[m][0][1]).sum(-1)
            if masks is not None:
                v = torch.where(masks[m].bool().unsqueeze(0), v, torch.full_like(v, -float("inf")))
            lqu.append(v)
        lqu = torch.logsumexp(torch.stack(lqu), dim=0) - torch.log(n_avail.to(z.dtype))
        lqw = latent_log_prob(fam, w, q[c][1][0], q[c][1][1]).sum(-1)
        lpx = 0
        for r in mods:
            lp = recon_log_prob(dists[r], recons[c][r], data[r], dist_scales.get(r, 1.0))
            lp = lp.reshape(z.shape[0], z.shape[1], -1).mul(rescale[r]).sum(-1)
            if masks is not None:
                lp = lp * masks[r].to(lp.dtype)
            lpx = lpx + lp
        lw = lpx + beta * (lpz - lqu - lqw)
        if masks is not None:
            lw = lw * masks[c].to(lw.dtype)
        lws[c] = lw
    if loss == "dreg_looser":
        wk = {}
        with torch.no_grad():
            for c in mods:
                wk[c] = (lws[c] - torch.logsumexp(lws[c], 0, keepdim=True)).exp()
        tot = torch.stack([lws[c] * wk[c] for c in mods]).sum(1)
        for c in mods:
            for tns in emb[c]:
                if tns.requires_grad:
                    tns.register_hook(lambda g, w_=wk[c]: w_.unsqueeze(-1) * g)
    else:
        tot = torch.logsumexp(torch.stack([lws[c] for c in mods]), dim=1) - math.log(K)
    tot = tot.sum(0) / n_avail.to(tot.dtype)
    loss_v = -tot.sum()
    return dict(loss=loss_v, loss_sum=loss_v, metrics={}, lws=lws, us={c: emb[c][0] for c in mods},
                ws={c: emb[c][1] for c in mods})


# ----------------------------------------------------------------------------------------------
# a16: JMVAE (loss assembly only; the joint encoder is a network, see nets.py)
# ----------------------------------------------------------------------------------------------
def jmvae_forward(joint, enc, data, decoders, eps, *, names, alpha=0.1, beta=1.0, warmup=10, epoch=1,
                  rescale=None, dists=None, dist_scales=None):
    """JMVAE.forward.  jmvae_model.py:116-192 (Appendix A.4).  joint = (mu, log_var) of the joint encoder."""
    rescale = rescale or {m: 1.0 for m in names}
    dists = dists or {m: "normal" for m in names}
    dist_scales = dist_scales or {}
    mu, lv = joint
    z = rsample(mu, lv, eps)
    B = mu.shape[0]
    rec = 0
    for m in names:
        recon = decoders[m](z)
        rec = rec + _row_nll(dists[m], recon, data[m], rescale[m], dist_scales.get(m, 1.0)).sum()
    kld = -0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp())
    ljm = 0
    for m in names:  # accumulated elementwise, then summed (:160-174)
        mu_m, lv_m = enc[m]
        ljm = ljm + 1 / 2 * (lv_m - lv + (torch.exp(lv) + (mu - mu_m) ** 2) / torch.exp(lv_m) - 1)
    ljm = ljm.sum() * alpha
    kld = kld * beta
    a = 1.0 if epoch >= warmup else epoch / warmup
    loss_sum = rec + a * (kld + ljm)
    return dict(loss=loss_sum / B, loss_sum=loss_sum,
                metrics=dict(loss_no_ponderation=rec + kld + ljm, beta=a, elbo=(rec + kld) / B), z=z)


# ----------------------------------------------------------------------------------------------
# importance-sampled joint likelihood (compute_joint_nll)
# ----------------------------------------------------------------------------------------------
def iwae_joint_nll(z, data, decoders, experts, *, names, dists=None, scales=None, family="normal", prior=None,
                   batch_size_K=100, private=None):
    """-sum_i [ logsumexp_k ( sum_m ln p(x_m,i | z_ik) + ln p(z_ik) - ln q(z_ik | x_i) ) - ln K ].

    Follows the per-data-point / per-K-chunk loops shared by mopoe_model.py:522-592, mmvae_model.py:399-441,
    mvtcae_model.py:249-289 and joint_model.py:111-152:
      * z [K,B,L] are the importance samples (the reference permutes them to [B,K,L]);
      * the likelihood terms use `recon_log_probs[m]` WITHOUT the rescale factors;
      * q is the uniform mixture of `experts` = [(loc [B,L], scale [B,L]), ...]:
        logsumexp_e sum_l log q_e(z) - ln E (a single expert gives the plain log-density);
      * the K samples are visited in chunks of batch_size_K, the chunk logsumexps are collected in a float32
        `torch.Tensor(lnpxs)` and combined by a second logsumexp minus ln K.
    prior: None = N(0, I) (`dist.Normal(0, 1)`), or (loc [1,L], scale [1,L]) of the latent family (MMVAE).
    private: {m: (w [K,B,S_m], mu [B,S_m], logvar [B,S_m])} modality-specific latents of MoPoE (mopoe_model.py:507-521,
    :543-567): decoder m sees [z, w_m]; ln N(w_m; 0, 1) joins ln p(z) and ln q(w_m | x_m) joins ln q.
    Returns (nll scalar, ll [B], lw [K,B]) -- the last two are intermediates for the kernel tests.
    """
    K, B, L = z.shape
    dists = dists or {}
    scales = scales or {}
    fam = _latent_family(family)
    ll, lws = [], []
    for i in range(B):
        lnpxs, lw_i = [], []
        start = 0
        while start < K:
            stop = min(start + batch_size_K, K)
            latents = z[start:stop, i]  # [k, L]
            lpx = 0
            lpz_priv, lq_priv = 0, 0
            for m in names:
                full = latents
                if private is not None:
                    w_m = private[m][0][start:stop, i]
                    full = torch.cat([latents, w_m], dim=-1)
                    sd_m = torch.exp(0.5 * private[m][2][i])
                    lpz_priv = lpz_priv + latent_log_prob("normal", w_m, torch.zeros(()), torch.ones(())).sum(-1)
                    lq_priv = lq_priv + latent_log_prob("normal", w_m, private[m][1][i], sd_m).sum(-1)
                recon = decoders[m](full)
                x_m = data[m][i]
                lp = recon_log_prob(dists.get(m, "normal"), recon, torch.stack([x_m] * len(recon)), scales.get(m, 1.0))
                lpx = lpx + lp.reshape(recon.size(0), -1).sum(-1)
            if prior is None:
                lpz = latent_log_prob("normal", latents, torch.zeros(()), torch.ones(())).sum(-1)
            else:
                lpz = latent_log_prob(fam, latents, prior[0], prior[1]).sum(-1)
            lqs = torch.stack([latent_log_prob(fam, latents, loc[i], scale[i]).sum(-1) for loc, scale in experts])
            lqz = torch.logsumexp(lqs, dim=0) - math.log(len(experts))
            lpz, lqz = lpz_priv + lpz, lq_priv + lqz
            w = lpx + lpz - lqz
            lw_i.append(w)
            lnpxs.append(torch.logsumexp(w, dim=0))
            start = stop
        ll.append(torch.logsumexp(torch.Tensor([float(v) for v in lnpxs]), dim=0) - math.log(K))
        lws.append(torch.cat(lw_i))
    ll = torch.stack(ll)
    return -ll.sum(), ll, torch.stack(lws, dim=1)


def mopoe_joint_nll(enc, data, decoders, eps, *, names, dists=None, batch_size_K=100, style_eps=None):
    """MoPoE.compute_joint_nll, mopoe_model.py:467-594 (complete data): samples from the row-range-selected subset
    posterior (`inference()["joint"]`), scores them under the uniform mixture of all subset posteriors.  eps [K,B,L];
    style_eps {m: [K,B,S_m]} with modality-specific latent spaces (enc[m] = (mu, lv, style_mu, style_lv))."""
    inf = mopoe_inference(enc, names)
    z = rsample(inf["joint_mu"], inf["joint_logvar"], eps)
    experts = [(inf["mus"][s], torch.exp(0.5 * inf["logvars"][s])) for s in range(inf["mus"].shape[0])]
    private = None
    if style_eps is not None:
        private = {m: (rsample(enc[m][2], enc[m][3], style_eps[m]), enc[m][2], enc[m][3]) for m in names}
    return iwae_joint_nll(z, data, decoders, experts, names=names, dists=dists, batch_size_K=batch_size_K,
                          private=private)


def mvtcae_joint_nll(enc, data, decoders, eps, *, names, batch_size_K=100):
    """MVTCAE.compute_joint_nll, mvtcae_model.py:213-291: q = the product of the unimodal experts (`poe`, no prior
    expert, mvtcae_model.py:134-169)."""
    mu, lv = poe(torch.stack([enc[m][0] for m in names]), torch.stack([enc[m][1] for m in names]))
    sd = torch.exp(0.5 * lv)
    z = mu + sd * eps
    return iwae_joint_nll(z, data, decoders, [(mu, sd)], names=names, batch_size_K=batch_size_K)


def jmvae_joint_nll(joint, data, decoders, eps, *, names, dists=None, batch_size_K=100):
    """BaseJointModel.compute_joint_nll, joint_model.py:82-154: q = the joint encoder's Gaussian."""
    mu, lv = joint
    sd = torch.exp(0.5 * lv)
    z = mu + sd * eps
    return iwae_joint_nll(z, data, decoders, [(mu, sd)], names=names, dists=dists, batch_size_K=batch_size_K)


def mmvae_joint_nll(enc, data, decoders, noise, *, names, sampled, family="laplace_with_softmax", prior_log_var=None,
                    dists=None, batch_size_K=100):
    """MMVAE.compute_joint_nll, mmvae_model.py:365-443: the samples come from ONE modality's posterior (`encode`,
    :341-351, `np.random.choice(cond_mod)` = `sampled`) and are scored under the mixture of the M unimodal
    posteriors; the prior is `prior_dist(*pz_params)` (:76-93).  noise [K,B,L] (normal or uniform, Appendix B)."""
    fam = _latent_family(family)
    L = enc[names[0]][0].shape[-1]
    experts = [(enc[m][0], mmvae_std(enc[m][1], family)) for m in names]
    loc, scale = experts[names.index(sampled)]
    z = latent_rsample(fam, loc, scale, noise)
    plv = torch.zeros(1, L) if prior_log_var is None else prior_log_var
    prior = (torch.zeros(1, L), mmvae_std(plv, family))
    return iwae_joint_nll(z, data, decoders, experts, names=names, dists=dists, family=family, prior=prior,
                          batch_size_K=batch_size_K)


def mmvaeplus_joint_nll(enc, data, decoders, noise, *, names, K, family="laplace_with_softmax", prior_logvars=None,
                        dists=None):
    """MMVAEPlus.compute_joint_nll, mmvaePlus_model.py:477-531, as the reference executes it:
      * `n_data = len(inputs.data.popitem()[1])` (:497) REMOVES the last modality from the inputs, so the loop below
        conditions on and reconstructs the first M-1 modalities only, while the mixture normaliser of
        `_compute_k_lws` stays ln(n_modalities) (:239) and k = K // n_modalities (:511);
      * rescale factors and beta are forced to 1 (:502-506);
      * per data point: the forward's importance weights lws[c] [k,1] of every conditioning modality are concatenated
        and reduced by logsumexp - ln(size) (:521-525).
    enc: {name: (mu, lv, mu_style, lv_style)} on the whole batch; noise as in mmvaeplus_forward with K -> k.
    Returns (nll, ll [B])."""
    kept = list(data.keys())[:-1]
    k = K // len(names)
    B = data[kept[0]].shape[0]
    ll = []
    for i in range(B):
        data_i = {m: data[m][i].unsqueeze(0) for m in kept}
        enc_i = {m: tuple(t[i : i + 1] for t in enc[m]) for m in kept}
        noise_i = {c: {key: v[:, i : i + 1] for key, v in noise[c].items()} for c in kept}
        o = mmvaeplus_forward(enc_i, data_i, decoders, noise_i, names=names, K=k, family=family, loss="iwae_looser",
                              beta=1.0, prior_logvars=prior_logvars, rescale={m: 1.0 for m in names}, dists=dists)
        lws = torch.cat([o["lws"][c] for c in kept], dim=0)  # [(M-1) k, 1]
        ll.append(torch.logsumexp(lws, dim=0) - math.log(lws.size(0)))
    ll = torch.cat(ll)
    return -ll.sum(), ll


def mopoe_subset_joint_nll(enc, data, decoders, eps, *, names, subset, dists=None, batch_size_K=100):
    """MoPoE._compute_joint_nll_from_subset_encoding, mopoe_model.py:596-701 (compute_joint_nll_paper :703-718 passes
    the full subset): samples AND density from the posterior of `subset` alone."""
    inf = mopoe_inference(enc, names)
    s = inf["keys"].index("_".join(sorted(subset)))
    mu, sd = inf["mus"][s], torch.exp(0.5 * inf["logvars"][s])
    return iwae_joint_nll(mu + sd * eps, data, decoders, [(mu, sd)], names=names, dists=dists, batch_size_K=batch_size_K)


# ----------------------------------------------------------------------------------------------
# MVAE (sub-sampled product-of-experts ELBOs)
# ----------------------------------------------------------------------------------------------
def mvae_subsets(names, use_subsampling=True, random_subsets=()):
    """Subsets of one MVAE objective in evaluation order, mvae_model.py:176-191: the joint one, then (sub-sampling)
    the unimodal ones, then the k randomly drawn ones (`random_subsets`: lists of names, the np.random.choice result)."""
    out = [list(names)]
    if use_subsampling:
        out += [[m] for m in names]
        out += [list(s) for s in random_subsets]
    return out


def mvae_annealing(epoch, batch_ratio, warmup, beta):
    """mvae_model.py:166-171."""
    return 1 * beta if epoch >= warmup else (epoch - 1 + batch_ratio) / warmup * beta


def mvae_forward(enc, data, decoders, eps, *, names, subsets, beta=1.0, rescale=None, dists=None, dist_scales=None,
                 masks=None):
    """MVAE.forward, mvae_model.py:145-228 with _compute_elbo_subset :86-118, compute_mu_log_var_subset :53-84 and
    _filter_inputs_with_masks :120-143.

    enc: {name: (mu [B,L], logvar [B,L])} on the whole batch (the reference re-runs the encoders on the filtered
    batch of every subset; the default networks are row-independent).  eps [S,B,L]: row b of eps[s] is the noise of
    data point b in subset s (the reference draws one [n_s, L] tensor per subset for the rows it keeps).
    beta: the annealed KL weight (mvae_annealing).  Returns loss, loss_sum, metrics and per-subset intermediates."""
    rescale = rescale or {m: 1.0 for m in names}
    dists = dists or {}
    dist_scales = dist_scales or {}
    total, metrics, zs, sub = 0, {}, [], []
    len_batch = 0.0
    for si, s in enumerate(subsets):
        if masks is not None:
            filt = torch.zeros_like(masks[s[0]], dtype=torch.bool)
            for m in s:
                filt = torch.logical_or(filt, masks[m].bool())
            if not bool(filt.any()):
                len_batch = 0.0
                zs.append(None)
                sub.append(None)
                continue
        else:
            filt = torch.ones(eps.shape[1], dtype=torch.bool)
        mus, lvs = [], []
        for m in names:
            if m in s:
                mu_m, lv_m = enc[m][0][filt], enc[m][1][filt].clone()
                if masks is not None:
                    lv_m[~masks[m].bool()[filt]] = float("inf")
                mus.append(mu_m)
                lvs.append(lv_m)
        mus.append(torch.zeros_like(mus[0]))
        lvs.append(torch.zeros_like(lvs[0]))
        sub_mu, sub_lv = stable_poe(torch.stack(mus), torch.stack(lvs))
        z = rsample(sub_mu, sub_lv, eps[si][filt])
        recon_sum = 0
        for m in names:
            if m in s:
                rec = decoders[m](z)
                r = _row_nll(dists.get(m, "normal"), rec, data[m][filt], rescale[m], dist_scales.get(m, 1.0))
                if masks is not None:
                    r = masks[m][filt].float() * r
                recon_sum = recon_sum + r.sum()
        kld = -0.5 * torch.sum(1 + sub_lv - sub_mu.pow(2) - sub_lv.exp())
        n = len(sub_mu)
        elbo = (recon_sum + kld * beta) / n
        key = "_".join(sorted(s))
        metrics[key] = elbo
        metrics["beta"] = beta
        metrics["kld" + key] = kld / n
        # `recon = elbo_sub` (:99) aliases the tensor that `elbo_sub += KLD * beta` (:102) then updates in place: the
        # reference's "recon" metric IS the subset ELBO.  Reproduced.
        metrics["recon" + key] = elbo
        total = total + elbo
        len_batch = n
        zs.append(z)
        sub.append((sub_mu, sub_lv))
    return dict(loss=total, loss_sum=total * len_batch, metrics=metrics, zs=zs, subs=sub)


def mvae_joint_nll(enc, data, decoders, eps, *, names, dists=None, batch_size_K=100):
    """MVAE.compute_joint_nll, mvae_model.py:266-340: q = stable_poe of all experts and the prior."""
    mus = [enc[m][0] for m in names] + [torch.zeros_like(enc[names[0]][0])]
    lvs = [enc[m][1] for m in names] + [torch.zeros_like(enc[names[0]][1])]
    mu, lv = stable_poe(torch.stack(mus), torch.stack(lvs))
    sd = torch.exp(0.5 * lv)
    return iwae_joint_nll(mu + sd * eps, data, decoders, [(mu, sd)], names=names, dists=dists, batch_size_K=batch_size_K)


# ----------------------------------------------------------------------------------------------
# CRMVAE
# ----------------------------------------------------------------------------------------------
def crmvae_forward(enc, data, decoders, eps, mod_eps, *, names, beta=2.5, rescale=None, dists=None, dist_scales=None,
                   masks=None):
    """CRMVAE.forward, crmvae_model.py:37-105 (+ _infer_all_latent_parameters :133-165): PoE of the available experts
    (`poe`, +inf log-variance for missing rows, no prior expert), KL(joint || N(0,I)) and KL(joint || q_m) (masked) via
    `kl_divergence` (base_utils.py:90-119), every modality reconstructed from the joint sample AND from a sample of its
    own (unmasked) posterior; per-sample loss = sum recon / (2 (M+1)) + beta (sum KL) / (M+1), summed over the batch.
    eps [B,L]: joint noise; mod_eps {m: [B,L]}: unimodal noise (drawn after the joint one, in data order)."""
    rescale = rescale or {m: 1.0 for m in names}
    dists = dists or {}
    dist_scales = dist_scales or {}
    M = len(names)
    mus = torch.stack([enc[m][0] for m in names])
    lvs = []
    for m in names:
        lv = enc[m][1]
        if masks is not None:
            lv = torch.where(masks[m].bool().unsqueeze(-1), lv, torch.full_like(lv, float("inf")))
        lvs.append(lv)
    jmu, jlv = poe(mus, torch.stack(lvs))
    zs = {"joint": rsample(jmu, jlv, eps)}
    joint_kld = kl_divergence(jmu, jlv, torch.zeros_like(jmu), torch.zeros_like(jlv))
    metrics = {"joint_divergence": joint_kld.mean()}
    divergence = joint_kld
    for m in names:
        mu, lv = enc[m]
        zs[m] = rsample(mu, lv, mod_eps[m])
        kl = kl_divergence(jmu, jlv, mu, lv)
        if masks is not None:
            kl = kl * masks[m].float()
        divergence = divergence + kl
        metrics[f"kl_{m}"] = kl.mean()
    loss_rec = 0
    for g in names:
        for src in ("joint", g):
            r = _row_nll(dists.get(g, "normal"), decoders[g](zs[src]), data[g], rescale[g], dist_scales.get(g, 1.0))
            if masks is not None:
                r = masks[g].float() * r
            loss_rec = loss_rec + r
            metrics[f"recon_{g}_from_{src}"] = r.mean()
    total = loss_rec / (2 * (M + 1)) + beta * divergence / (M + 1)
    return dict(loss=total.sum(), loss_sum=total.sum(), metrics=metrics, zs=zs, joint_mu=jmu, joint_logvar=jlv)


# ----------------------------------------------------------------------------------------------
# DMVAE
# ----------------------------------------------------------------------------------------------
def dmvae_forward(enc, data, decoders, noise, *, names, beta=1.0, private_betas=None, rescale=None, dists=None,
                  dist_scales=None, masks=None):
    """DMVAE.forward, dmvae_model.py:152-240: M + 1 negative ELBOs per sample -- q(shared) = stable_poe of the available
    shared experts and the prior (:131-148), then every modality's own shared posterior (x its mask) -- each with fresh
    private samples; -sum_m mask_m log p(x_m | [z, w_m]) + beta KL(q(shared) || N(0,I)) + sum_m beta_m mask_m KL(q(w_m) ||
    N(0,I)); loss = mean over the batch of the sum of the ELBOs.
    enc: {m: (mu, lv, style_mu, style_lv)}; noise = {"shared": [M+1,B,L], "private": {m: [M+1,B,S_m]}} (slab 0: joint
    ELBO, slab 1+k: ELBO of the k-th modality)."""
    rescale = rescale or {m: 1.0 for m in names}
    private_betas = private_betas or {m: 1.0 for m in names}
    dists = dists or {}
    dist_scales = dist_scales or {}
    mus = [enc[m][0] for m in names]
    lvs = []
    for m in names:
        lv = enc[m][1]
        if masks is not None:
            lv = torch.where(masks[m].bool().unsqueeze(-1), lv, torch.full_like(lv, float("inf")))
        lvs.append(lv)
    mus.append(torch.zeros_like(mus[0]))
    lvs.append(torch.zeros_like(lvs[0]))
    jmu, jlv = stable_poe(torch.stack(mus), torch.stack(lvs))

    def neg_elbo(e, q_mu, q_lv):
        shared_z = rsample(q_mu, q_lv, noise["shared"][e])
        recon = 0
        for m in names:
            z_mod = rsample(enc[m][2], enc[m][3], noise["private"][m][e])
            rec = decoders[m](torch.cat([shared_z, z_mod], dim=1))
            lp = -_row_nll(dists.get(m, "normal"), rec, data[m], rescale[m], dist_scales.get(m, 1.0))
            if masks is not None:
                lp = masks[m].float() * lp
            recon = recon + lp
        kl = kl_divergence(q_mu, q_lv, torch.zeros_like(q_mu), torch.zeros_like(q_lv)) * beta
        for m in names:
            kl_mod = kl_divergence(enc[m][2], enc[m][3], torch.zeros_like(enc[m][2]), torch.zeros_like(enc[m][3]))
            if masks is not None:
                kl_mod = masks[m].float() * kl_mod
            kl = kl + kl_mod * private_betas[m]
        return -recon + kl

    joint = neg_elbo(0, jmu, jlv)
    metrics = {"joint": joint.mean()}
    loss = joint
    for k, m in enumerate(names):
        mod = neg_elbo(1 + k, enc[m][0], enc[m][1])
        if masks is not None:
            mod = masks[m] * mod
        loss = loss + mod
        metrics[m] = mod.mean()
    return dict(loss=loss.mean(), metrics=metrics, joint_mu=jmu, joint_logvar=jlv)


def cond_nll(z, data, decoders, *, pred_mods, dists=None, scales=None):
    """BaseMultiVAE.compute_cond_nll, base_ae_model.py:396-442: z [K,B,L] are the K conditional encodings (one
    `encode(inputs, subset)` call per k in the reference); per predicted modality
    -(1/B) sum_b [ logsumexp_k sum_d log p(x_b | dec(z_kb)) - ln K ]  (unrescaled log-probabilities)."""
    dists = dists or {}
    scales = scales or {}
    K = z.shape[0]
    out = {}
    for m in pred_mods:
        lp = torch.stack([-_row_nll(dists.get(m, "normal"), decoders[m](z[k]), data[m], 1.0, scales.get(m, 1.0))
                          for k in range(K)])
        ll = torch.logsumexp(lp, dim=0) - math.log(K)
        out[m] = -torch.sum(ll) / len(ll)
    return out


def mmvae_joint_nll_paper(enc, data, decoders, noises, *, names, K, batch_size_K, family="laplace_with_softmax",
                          prior_log_var=None, rescale=None, dists=None):
    """MMVAE.compute_joint_nll_paper, mmvae_model.py:444-468, with `iwae` :294-311.  noises: one {modality: [n,B,L]} per
    chunk of n = min(batch_size_K, remaining) samples (a forward pass per chunk)."""
    M = len(names)
    vals, done, c = [], 0, 0
    while done < K:
        n = min(batch_size_K, K - done)
        done += n
        o = mmvae_forward(enc, data, decoders, noises[c], names=names, K=n, family=family, loss="iwae_looser",
                          prior_log_var=prior_log_var, rescale=rescale, dists=dists)
        lws = torch.stack([o["lws"][m] for m in names], dim=0)  # [M, n, B]
        l = torch.logsumexp(lws, dim=1) - math.log(lws.size(1))
        l = torch.logsumexp(l, dim=0) - math.log(M)
        vals.append(l.sum() + math.log(n * M))
        c += 1
    return -(torch.logsumexp(torch.stack(vals), dim=0) - math.log(done * M))


def dmvae_joint_nll(enc, data, decoders, noise, *, names, batch_size_K=100, dists=None, scales=None):
    """DMVAE.compute_joint_nll (dmvae_model.py:311-412) as written: K samples of the shared latent from the joint posterior
    (stable_poe of the shared experts and the prior), fresh private samples of every modality per chunk of batch_size_K, and
    `ln_prior` / `ln_posterior` initialised ONCE before the loop over the data points (:352) and only ever added to
    (:385-403): the importance weights of a chunk carry the log-densities of every earlier chunk and data point.
    enc: {m: (mu, lv, mu_private, lv_private)}; noise: {"shared": [K,B,L], "private": {m: [K,B,S_m]}} with sample
    k = chunk * batch_size_K + r.  -> (nll, ll [B])"""
    dists, scales = dists or {}, scales or {}
    K, B, L = noise["shared"].shape
    mus = torch.stack([enc[m][0] for m in names] + [torch.zeros_like(enc[names[0]][0])])
    lvs = torch.stack([enc[m][1] for m in names] + [torch.zeros_like(enc[names[0]][1])])
    mu, lv = stable_poe(mus, lvs)
    sigma = torch.exp(0.5 * lv)
    z_joint = mu + sigma * noise["shared"]  # [K,B,L]
    ln_prior, ln_posterior = 0, 0
    lls = []
    for i in range(B):
        lnpxs = []
        start = 0
        while start < K:
            stop = min(start + batch_size_K, K)
            shared = z_joint[start:stop, i]
            lpx = 0
            for m in names:
                mu_p, sd_p = enc[m][2][i], torch.exp(0.5 * enc[m][3][i])
                priv = mu_p + sd_p * noise["private"][m][start:stop, i]
                recon = decoders[m](torch.cat([shared, priv], dim=-1))
                lp = recon_log_prob(dists.get(m, "normal"), recon, torch.stack([data[m][i]] * len(recon)), scales.get(m, 1.0))
                lpx = lpx + lp.reshape(recon.size(0), -1).sum(-1)
                ln_prior = ln_prior + latent_log_prob("normal", priv, torch.zeros(()), torch.ones(())).sum(-1)
                ln_posterior = ln_posterior + latent_log_prob("normal", priv, mu_p, sd_p).sum(-1)
            ln_prior = ln_prior + latent_log_prob("normal", shared, torch.zeros(()), torch.ones(())).sum(-1)
            ln_posterior = ln_posterior + latent_log_prob("normal", shared, mu[i], sigma[i]).sum(-1)
            lnpxs.append(torch.logsumexp(lpx + ln_prior - ln_posterior, dim=0))
            start = stop
        lls.append(torch.logsumexp(torch.Tensor([float(v) for v in lnpxs]), dim=0) - math.log(K))
    ll = torch.stack(lls)
    return -ll.sum(), ll
