"""MoPoE (Sutter 2021) on the HIP kernels.  Mirrors `multivae/models/mopoe/mopoe_model.py`:
subset enumeration :71-106, forward :147-227, inference :274-350, selection :417-465, divergence :108-145.

forward = M encoder nodes -> ONE fused posterior kernel (per-subset PoE, prior expert on the full subset,
subset selection, reparameterisation over K samples, all-subset KL) -> M decoder nodes -> ONE fused
reconstruction-NLL kernel (+ d loss / d recon in the same pass) -> ONE scalar assembly kernel.
"""
import math
import os
from itertools import chain, combinations
from typing import Union

import torch

from ... import _lib, kernels
from ..base import BaseMultiVAE
from ..base.base_utils import ModelOutput
from .mopoe_config import MoPoEConfig


_EARLY_NOISE = _lib.tune("MVK_EARLY_NOISE", "1") != "0"  # 0: draw the noise behind the encoders (A/B)
# MVK_ENC_SIDE_FIRST=1: the short encoder's node is created FIRST, so that autograd enqueues the LONG encoder's backward first (its
# launches are the step's last chain; A/B)
_ENC_SIDE_FIRST = kernels._lib.tune("MVK_ENC_SIDE_FIRST", "0") == "1"


# 1: the fused decoder tails' z-independent preparation (dense16 pack + target bound, 18 us) behind the short encoder instead of in
# the decoder's own chain.  Off since round 5: the short encoder's branch had become the LONGER one (117 vs 109 us at the join in
# front of the posterior); in the MLP decoder's chain the two launches sit in slack.  1.0205-1.0238 -> 1.0090-1.0148 ms, four pairs.
_EARLY_DENSE = _lib.tune("MVK_EARLY_DENSE", "0") == "1"
_ROW_WEIGHT = {}


def _takes_row_weight(dec):
    """Whether a decoder's `reconstruction_nll` has the `row_weight` argument (in-package decoders; cached per class)."""
    k = type(dec)
    if k not in _ROW_WEIGHT:
        import inspect

        try:
            _ROW_WEIGHT[k] = "row_weight" in inspect.signature(dec.reconstruction_nll).parameters
        except (TypeError, ValueError):
            _ROW_WEIGHT[k] = False
    return _ROW_WEIGHT[k]


class MoPoE(BaseMultiVAE):
    def __init__(self, model_config: MoPoEConfig, encoders: dict = None, decoders: dict = None):
        super().__init__(model_config, encoders, decoders)
        self.multiple_latent_spaces = model_config.modalities_specific_dim is not None
        self.model_name = "MoPoE"
        list_subsets = self.model_config.subsets
        if isinstance(list_subsets, dict):
            list_subsets = list(list_subsets.values())
        if list_subsets is None:
            list_subsets = self.all_subsets()
        self.set_subsets(list_subsets)
        self._sel_cache = {}
        # decoders that can score their own output (`reconstruction_nll`, e.g. Decoder_VAE_SVHN) do so in the epilogue of their
        # last layer; False = always decode + the generic likelihood kernel (A/B, and what every user-written decoder gets)
        self.fused_decoder_tail = _lib.tune("MVK_FUSED_TAIL", "1") != "0"
        if self.multiple_latent_spaces:  # default multi-latent MLPs (mopoe_model.py:58-75)
            from ..nn.default_architectures import BaseDictDecodersMultiLatents, BaseDictEncoders_MultiLatents

            self.style_dims = model_config.modalities_specific_dim
            if encoders is None:
                self.set_encoders(BaseDictEncoders_MultiLatents(input_dims=model_config.input_dims,
                                                                latent_dim=model_config.latent_dim,
                                                                modality_dims=model_config.modalities_specific_dim))
            if decoders is None:
                self.set_decoders(BaseDictDecodersMultiLatents(input_dims=model_config.input_dims,
                                                               latent_dim=model_config.latent_dim,
                                                               modality_dims=model_config.modalities_specific_dim))

    # -- subsets ---------------------------------------------------------------------------------------
    def all_subsets(self):
        xs = list(self.encoders.keys())
        return chain.from_iterable(combinations(xs, n) for n in range(len(xs) + 1))

    def set_subsets(self, subsets_list):
        subsets = dict()
        for mod_names in subsets_list:
            mods = []
            for mod_name in sorted(mod_names):
                if (mod_name not in self.encoders.keys()) and (mod_name != ""):
                    raise AttributeError(f"The provided subsets list contains unknown modality name {mod_name}."
                                         " that is not the encoders dictionary or inputs_dim dictionary.")
                mods.append(mod_name)
            subsets["_".join(sorted(mod_names))] = mods
        self.subsets = subsets
        self.model_config.subsets = subsets
        # kernel-side description: modalities in PoE summation order (sorted names), one bit mask per
        # non-empty subset in enumeration order
        self._poe_order = sorted(self.encoders.keys())
        pos = {m: i for i, m in enumerate(self._poe_order)}
        self._subset_keys = [k for k in subsets if k != ""]
        self._subset_bits = [sum(1 << pos[m] for m in subsets[k]) for k in self._subset_keys]
        self._subset_masks_dev = {}

    def _subset_masks(self, device):
        t = self._subset_masks_dev.get(device)
        if t is None:
            t = torch.tensor(self._subset_bits, dtype=torch.int32, device=device)
            self._subset_masks_dev[device] = t
        return t

    def _row_range_selection(self, B, device):
        """deterministic_mixture_component_selection (:435-465): rows [k*n,(k+1)*n) take subset k,
        n = floor(B * float32(1/S)); the last subset takes the remainder."""
        key = (B, device)
        sel = self._sel_cache.get(key)
        if sel is None:
            S = len(self._subset_keys)
            n = int(torch.floor(B * torch.tensor(1.0 / float(S), dtype=torch.float32)))
            idx = torch.arange(B)
            sel = torch.clamp(idx // max(n, 1), max=S - 1) if n > 0 else torch.full((B,), S - 1)
            sel = sel.to(torch.int32).to(device)
            self._sel_cache[key] = sel
        return sel

    def subset_mask(self, inputs, subset):
        filt = None
        for mod in subset:
            filt = inputs.masks[mod].bool() if filt is None else torch.logical_and(filt, inputs.masks[mod])
        return filt

    # -- forward -----------------------------------------------------------------------------------------
    def modality_encode(self, inputs, side_work=None, **kwargs):
        """side_work: a callable run on the LAST branch's stream (a side stream whose encoder is the short one when the branches
        are ordered longest first), behind that encoder: launches that do not depend on the encoders ride there for free."""
        names = self._branch_order(inputs)

        def run(m):
            out = self.encoders[m](inputs.data[m])
            if side_work is not None and len(names) > 1 and m == names[-1]:
                side_work()  # BEHIND the short encoder: its stream also carries the step's weight-pack launch (kernels.pack_scope)
            return out

        enc = kernels.run_branches(names, run, inputs.data[names[0]].device, side_first=_ENC_SIDE_FIRST)
        return {m: enc[m] for m in self.encoders.keys()}

    def _posterior(self, inputs, K, noise=None, choice=None, want_stats=False):
        early = {}
        x0 = next(iter(inputs.data.values()))
        if noise is None and x0.is_cuda and kernels.DEVICE_RNG and kernels.BRANCH_STREAMS and len(self.encoders) > 1 \
                and x0.dim() > 1 and _EARLY_NOISE:
            # the noise does not depend on the encoders: its launch rides at the head of the short encoder's branch stream
            # instead of sitting between the encoders and the posterior kernel on the main one
            shape = (K, x0.shape[0], self.latent_dim)
            def side_work():
                early.setdefault("eps", self._noise(shape, x0.device))
                if self.fused_decoder_tail and _EARLY_DENSE and not hasattr(inputs, "masks") and self.training:
                    for m, dec in self.decoders.items():  # z-independent preparation of the fused decoder tails
                        if hasattr(dec, "early_work") and self.recon_dists[m][0] == kernels.DIST["normal"]:
                            dec.early_work(inputs.data[m], K * x0.shape[0])

            enc = self.modality_encode(inputs, side_work=side_work)
            if "eps" in early:  # allocated on the side stream, consumed on this one (behind the join of run_branches)
                early["eps"].record_stream(torch.cuda.current_stream(x0.device))
        else:
            enc = self.modality_encode(inputs)
        first = enc[self._poe_order[0]].embedding
        mus = [enc[m].embedding for m in self._poe_order]
        lvs = [enc[m].log_covariance for m in self._poe_order]
        if first.dim() == 1:  # single-sample batch: the SVHN encoder squeezed the batch dim (Appendix D)
            mus = [t.unsqueeze(0) for t in mus]
            lvs = [t.unsqueeze(0) for t in lvs]
        B, L = mus[0].shape
        device = mus[0].device
        masked = hasattr(inputs, "masks")
        weights = None
        if masked:
            avail = torch.stack([self.subset_mask(inputs, self.subsets[k]) for k in self._subset_keys]).float()
            weights = (avail / avail.sum(0)).contiguous()  # [S,B]
            if choice is None:  # random_mixture_component_selection (:417-433), drawn BEFORE eps
                choice_idx = torch.multinomial(weights.t(), 1).squeeze(1)
            else:
                choice_idx = choice.to(device).float().argmax(dim=1)
            sel = choice_idx.to(torch.int32).contiguous()
        else:
            sel = self._row_range_selection(B, device)
        eps = early.get("eps")
        if eps is None or tuple(eps.shape) != (K, B, L):
            eps = self._noise((K, B, L), device, noise)
        outs = kernels.MoPoEPosteriorFn.apply(eps, self._subset_masks(device), sel, weights, want_stats, *mus, *lvs)
        return enc, outs, (B, L, device, weights)

    def forward(self, inputs, **kwargs) -> ModelOutput:
        K = int(kwargs.pop("K", self.model_config.K))
        noise = kwargs.pop("noise", None)
        if noise is not None and noise.dim() == 2:
            noise = noise.unsqueeze(0)
        style_noise = kwargs.pop("style_noise", None)  # {modality: [K,B,S_m]} (drawn after the shared noise, in encoder order)
        enc, outs, (B, L, device, _) = self._posterior(inputs, K, noise=noise, choice=kwargs.pop("choice", None))
        z, kld_rows = outs[0], outs[1]
        names = list(self.encoders.keys())
        z_in = z[0] if K == 1 else z  # K == 1: decoders see [B,L] exactly like the reference
        masks = inputs.masks if hasattr(inputs, "masks") else None
        style_kl = []
        if self.multiple_latent_spaces:  # z_m = [shared, style_m], style_m ~ q(w_m | x_m)  (mopoe_model.py:171-178)
            z_ins = {}
            for m in names:
                try:
                    smu, slv = enc[m].style_embedding, enc[m].style_log_covariance
                except (AttributeError, KeyError):
                    raise AttributeError(" model_config.modality_specific_dims is not None, but encoder output for "
                                         f"modality {m} doesn't have a style_embedding attribute. When using multiple "
                                         "latent spaces, the encoders' output should be of the form : ModelOuput("
                                         "embedding = ...,style_embedding = ...,log_covariance = ..., "
                                         "style_log_covariance = ...)")
                if smu.dim() == 1:
                    smu, slv = smu.unsqueeze(0), slv.unsqueeze(0)
                sn = None if style_noise is None else style_noise[m]
                if sn is not None and sn.dim() == 2:
                    sn = sn.unsqueeze(0)
                w, kl = kernels.GaussSampleKLFn.apply(self._noise((K, B, smu.shape[-1]), device, sn), smu, slv)
                style_kl.append(kl)
                zw = torch.cat([z, w], dim=-1)
                z_ins[m] = zw[0] if K == 1 else zw
            z_of = lambda m: z_ins[m]
        else:
            z_of = lambda m: z_in

        # Opt-in fused decoder tail (round 3): an in-package decoder that can score its own output (`reconstruction_nll`:
        # Decoder_VAE_SVHN with a Normal likelihood) returns the NLL row sums [K, B] instead of the images; the reconstruction
        # term of that modality is then a plain sum of rows.  Complete data only; user decoders, other likelihoods and
        # evaluation (no grad) take the generic path below.
        normal = kernels.DIST["normal"]

        def decode(m):
            dec = self.decoders[m]
            if self.fused_decoder_tail and masks is None and self.recon_dists[m][0] == normal and hasattr(dec, "reconstruction_nll"):
                kw = {}
                if _takes_row_weight(dec):
                    kw["row_weight"] = float(self.rescale_factors[m]) / (K * B)  # = extra_coef * extra_lossw of the term below
                rows = dec.reconstruction_nll(z_of(m), inputs.data[m], "normal", self.recon_dists[m][1], **kw)
                if rows is not None:
                    return ("rows", rows)
            return ("rec", dec(z_of(m)).reconstruction)

        rec = kernels.run_branches(self._branch_order(inputs), decode, device, side_first=True)
        plain = [m for m in names if rec[m][0] == "rec"]
        fused = [m for m in names if rec[m][0] == "rows"]
        recons = [rec[m][1] for m in plain]
        spec = self._recon_spec(plain, inputs.data, masks, K, B)
        M, S, Fz = len(plain), len(style_kl), len(fused)
        beta = float(self.model_config.beta)
        spec.update(coef=[1.0 / (K * B)] * M, lossw=[1.0] * M,
                    extra_coef=[1.0 / B] * (1 + S) + [float(self.rescale_factors[m]) / (K * B) for m in fused],
                    extra_lossw=[beta] + [beta * float(self.model_config.beta_style)] * S + [1.0] * Fz,
                    loss_sum_scale=float(B),
                    # the backward nodes of every extra term (MoPoEPosteriorFn, GaussSampleKLFn, the fused tails) order themselves
                    # behind the assembly launch where they read what it fills: it may run beside the backward chain
                    # — only when every fused term's node is one of the package's (kernels.orders_behind_loss): a user decoder's
                    # `reconstruction_nll` built from plain autograd ops would read the row gradients unordered
                    async_ok=masks is None and all(kernels.orders_behind_loss(rec[m][1]) for m in fused),
                    # ... and with the unit seed nothing reads what it fills: the posterior node takes the KL rows' constant
                    # gradient from the host (kernels.const_grad), the fused tails theirs — the launch may run LAST
                    assembly_last=masks is None and not style_kl)
        if style_kl and masks is not None:  # style_kld *= mask (:217-218), still averaged over the whole batch
            style_kl = [kl * masks[m].to(kl.dtype) for kl, m in zip(style_kl, names)]
        loss, terms = kernels.ReconLossFn.apply(spec, M, *recons, kld_rows, *style_kl, *[rec[m][1] for m in fused])
        jd = terms[M]
        if style_kl:
            # `kld = results["joint_divergence"]` is updated in place by `kld += style_kld.mean() * beta_style`
            # (:164, :221): the reference's metric includes the style terms
            jd = jd + float(self.model_config.beta_style) * terms[M + 1:M + 1 + S].sum()
        metrics = {"joint_divergence": jd}
        for m in names:
            metrics["recon_" + m] = terms[plain.index(m)] if m in plain else terms[M + 1 + S + fused.index(m)]
        n_terms = M + 1 + S + Fz
        return ModelOutput(loss=loss, loss_sum=terms[n_terms + 1], metrics=metrics)

    def inference(self, inputs, **kwargs):
        """Subset and joint posterior parameters (:274-350).  Returns the same dict layout as the reference."""
        K = 1
        noise = kwargs.get("noise")
        if noise is not None:  # only the first sample matters here: the posterior parameters do not depend on it
            noise = (noise.unsqueeze(0) if noise.dim() == 2 else noise)[:1]
        enc, outs, (B, L, device, weights) = self._posterior(inputs, K, noise=noise,
                                                              choice=kwargs.get("choice"), want_stats=True)
        z, kld_rows, mus, lvs, jmu, jlv = outs
        S = len(self._subset_keys)
        if weights is None:
            weights = torch.full((S, B), 1.0 / float(S), device=device)
        return dict(modalities=enc, mus=mus, logvars=lvs, weights=weights, joint=[jmu, jlv],
                    subsets={k: [mus[i], lvs[i]] for i, k in enumerate(self._subset_keys)})

    def compute_joint_nll(self, inputs, K: int = 1000, batch_size_K: int = 100, **kwargs):
        """-sum_b ln p(x_b) by importance sampling (mopoe_model.py:467-594): K samples per data point from the
        selected subset posterior (`inference()["joint"]`), weighted against the uniform mixture of all subset
        posteriors.  `batch_size_K` is accepted for compatibility: the reference's chunked logsumexp is the same
        number.  kwargs: noise [K,B,L]."""
        self.eval()
        if hasattr(inputs, "masks"):
            raise AttributeError(self._NLL_INCOMPLETE)
        with torch.no_grad():
            enc, outs, (B, L, device, _) = self._posterior(inputs, int(K), noise=kwargs.get("noise"), want_stats=True)
            z, mus, lvs = outs[0], outs[2], outs[3]
            sds = kernels.std_from_logvar(lvs)
            S = mus.shape[0]
            private = None
            if self.multiple_latent_spaces:  # K private samples per modality too (:507-521); kwargs: style_noise
                style_noise = kwargs.get("style_noise")
                private = {}
                for m in inputs.data:
                    smu, slv = enc[m].style_embedding, enc[m].style_log_covariance
                    ssd = kernels.std_from_logvar(slv)
                    nz = self._noise((int(K), B, smu.shape[-1]), device, None if style_noise is None else style_noise[m])
                    private[m] = (kernels.iwae_sample(smu, ssd, nz), smu, ssd)
            return self._joint_nll(inputs, z, [mus[i] for i in range(S)], [sds[i] for i in range(S)], private=private)

    def _compute_joint_nll_from_subset_encoding(self, subset, inputs, K: int = 1000, batch_size_K: int = 100, **kwargs):
        """Joint NLL with ONE subset posterior as the importance distribution (mopoe_model.py:596-701): samples and
        density both come from `inference()["subsets"]["_".join(sorted(subset))]`.  kwargs: noise [K,B,L]."""
        self.eval()
        if hasattr(inputs, "masks"):
            raise AttributeError(self._NLL_INCOMPLETE)
        idx = self._subset_keys.index("_".join(sorted(subset)))
        with torch.no_grad():
            _, outs, (B, L, device, _) = self._posterior(inputs, 1, want_stats=True)
            mu, sd = outs[2][idx], kernels.std_from_logvar(outs[3][idx])
            z = kernels.iwae_sample(mu, sd, self._noise((int(K), B, L), device, kwargs.get("noise")))
            return self._joint_nll(inputs, z, [mu], [sd])

    def compute_joint_nll_paper(self, inputs, K: int = 1000, batch_size_K: int = 100, **kwargs):
        """The original paper's estimator: the PoE of all modalities as importance distribution (:703-718)."""
        return self._compute_joint_nll_from_subset_encoding(list(self.encoders.keys()), inputs, K, batch_size_K, **kwargs)

    def encode(self, inputs, cond_mod: Union[list, str] = "all", N: int = 1, return_mean=False, **kwargs):
        cond_mod = super().encode(inputs, cond_mod, N, **kwargs).cond_mod
        key = "_".join(sorted(cond_mod))
        with torch.no_grad():
            lat = self.inference(inputs)
            mu, log_var = lat["subsets"][key]
            if return_mean and len(cond_mod) == self.n_modalities:
                mu = torch.stack([lat["subsets"][k][0] for k in lat["subsets"]]).mean(0)
            z = self._gaussian_encoding(mu, log_var, N, return_mean, kwargs.pop("flatten", False), kwargs.get("noise"))
        return ModelOutput(z=z, one_latent_space=True)
