"""MMVAE (Shi 2019) on the HIP kernels.  Mirrors `multivae/models/mmvae/mmvae_model.py:44-292` (Appendix A.2):
per-modality Normal / Laplace(softmax-scale) posteriors, K samples each, M x M cross reconstructions,
mixture-of-experts log-density, IWAE or DReG objective (sum over the batch, loss_sum == loss).

Kernel sequence: M encoder nodes + M std kernels -> ONE latent kernel (samples, log p(z), log q_MoE(z)) ->
M x M decoder passes -> M reconstruction-NLL launches (rows) -> ONE objective kernel; backward: M
reconstruction launches with the per-row weights -> decoders -> ONE latent backward kernel -> encoders.
"""
from typing import Union

import numpy as np
import torch

from ... import _lib, kernels
from ...data.utils import drop_unused_modalities
from ..base import BaseMultiVAE
from ..base.base_utils import ModelOutput
from .mmvae_config import MMVAEConfig


class MMVAE(BaseMultiVAE):
    def __init__(self, model_config: MMVAEConfig, encoders: dict = None, decoders: dict = None):
        super().__init__(model_config, encoders, decoders)
        if model_config.prior_and_posterior_dist not in ("laplace_with_softmax", "normal"):
            raise AttributeError(" The posterior_dist parameter must be either 'laplace_with_softmax' or 'normal'. "
                                 f" {model_config.prior_and_posterior_dist} was provided.")
        self.prior_mean = torch.nn.Parameter(torch.zeros(1, self.latent_dim), requires_grad=False)
        self.prior_log_var = torch.nn.Parameter(torch.zeros(1, self.latent_dim),
                                                requires_grad=model_config.learn_prior)
        self.model_name = "MMVAE"

    @property
    def _family(self):
        return _lib.FAMILY[self.model_config.prior_and_posterior_dist]

    def log_var_to_std(self, log_var):
        return kernels.MMVAEStdFn.apply(log_var, self._family)

    @property
    def pz_params(self):
        return self.prior_mean, self.log_var_to_std(self.prior_log_var)

    def forward(self, inputs, **kwargs):
        inputs = drop_unused_modalities(inputs)
        K = int(kwargs.pop("K", self.model_config.K))
        noise = kwargs.pop("noise", None)  # {modality: [K,B,L]} explicit noise (Appendix B)
        mods = list(inputs.data.keys())
        M = len(mods)
        dreg = self.model_config.loss == "dreg_looser"
        if self.model_config.loss not in ("dreg_looser", "iwae_looser"):
            raise NotImplementedError()
        family = self._family
        def encode_one(m):
            out = self.encoders[m](inputs.data[m])
            mu, lv = out.embedding, out.log_covariance
            if mu.dim() == 1:
                mu, lv = mu.unsqueeze(0), lv.unsqueeze(0)
            return mu, self.log_var_to_std(lv)

        order = self._branch_order(inputs, mods)
        enc = kernels.run_branches(order, encode_one, inputs.data[order[0]].device)
        mus = [enc[m][0] for m in mods]
        sds = [enc[m][1] for m in mods]
        B, L = mus[0].shape
        device = mus[0].device
        noises = [self._noise((K, B, L), device, None if noise is None else noise[m], uniform=family == 1)
                  for m in mods]
        masks = None
        if hasattr(inputs, "masks"):
            masks = [inputs.masks[m].to(torch.bool).contiguous() for m in mods]
        state = kernels.MMVAEState()
        prior_std = self.log_var_to_std(self.prior_log_var)
        zs = kernels.MMVAELatentFn.apply(state, noises, masks, self.prior_mean.detach(), family, int(dreg), prior_std,
                                         *mus, *sds)
        # The reference decodes every (conditioning, target) pair on its own (:127: M^2 decoder passes of K * B rows); a
        # decoder that declares its rows independent (BaseDecoder.rows_independent: every in-package one) runs ONCE over the
        # M * K * B stacked rows (the same reconstructions from 1 / M of the launches); any other decoder pair by pair.
        zall = None
        if any(getattr(self.decoders[r], "rows_independent", False) for r in mods):
            zall = torch.cat([zs[c].reshape(-1, L) for c in range(M)], dim=0) if M > 1 else zs[0].reshape(-1, L)

        def decode_all(r):
            if not getattr(self.decoders[r], "rows_independent", False):
                return [self.decoders[r](zs[c].reshape(-1, L)).reconstruction for c in range(M)]
            rec = self.decoders[r](zall).reconstruction
            return list(rec.reshape(M, K * B, *rec.shape[1:]).unbind(0))

        dec = kernels.run_branches(self._branch_order(inputs, mods), decode_all, device)
        recons = [dec[r][c] for c in range(M) for r in mods]
        spec = self._recon_spec(mods, inputs.data, inputs.masks if masks is not None else None, K, B)
        loss = kernels.MMVAEObjectiveFn.apply(state, spec, M, dreg, *recons)
        out = ModelOutput(loss=loss, loss_sum=loss, metrics={})
        if kwargs.pop("detailed_output", False):
            out["zss"] = {m: zs[i] for i, m in enumerate(mods)}
            out["lws"] = {m: state.lw[i] for i, m in enumerate(mods)}
        return out

    def _unimodal_posteriors(self, inputs, mods):
        def encode_one(m):
            out = self.encoders[m](inputs.data[m])
            mu, lv = out.embedding, out.log_covariance
            if mu.dim() == 1:
                mu, lv = mu.unsqueeze(0), lv.unsqueeze(0)
            return mu, kernels.std_from_logvar(lv, self._family)

        order = self._branch_order(inputs, mods)
        return kernels.run_branches(order, encode_one, inputs.data[order[0]].device)

    def encode(self, inputs, cond_mod: Union[list, str] = "all", N: int = 1, return_mean=False, **kwargs):
        """mmvae_model.py:312-363: the mean of the conditioning posteriors' means, or N samples from ONE conditioning
        modality's posterior picked with `np.random.choice`.  kwargs: noise [N,B,L] (or [B,L] for N == 1)."""
        cond_mod = super().encode(inputs, cond_mod, N, **kwargs).cond_mod
        flatten = kwargs.pop("flatten", False)
        with torch.no_grad():
            if return_mean:
                post = self._unimodal_posteriors(inputs, list(cond_mod))
                emb = torch.stack([post[m][0] for m in cond_mod]).mean(0)
                z = torch.stack([emb] * N) if N > 1 else emb
            else:
                mod = kwargs.pop("sampled", None) or str(np.random.choice(cond_mod))
                mu, sd = self._unimodal_posteriors(inputs, [mod])[mod]
                B, L = mu.shape
                noise = kwargs.pop("noise", None)
                if noise is not None and noise.dim() == 2:
                    noise = noise.unsqueeze(0)
                z = kernels.iwae_sample(mu, sd, self._noise((N, B, L), mu.device, noise, uniform=self._family == 1),
                                        self._family)
                if N == 1:
                    z = z[0]
            if flatten:
                z = z.reshape(-1, self.latent_dim)
        return ModelOutput(z=z, one_latent_space=True)

    def compute_joint_nll(self, inputs, K: int = 1000, batch_size_K: int = 100, **kwargs):
        """-sum_b ln p(x_b) by importance sampling (mmvae_model.py:365-443): the K samples of every data point come
        from one modality's posterior (`encode`), the weights use the mixture of the M unimodal posteriors and the
        learnable prior.  kwargs: noise [K,B,L], sampled (modality name; default np.random.choice as the reference)."""
        self.eval()
        if hasattr(inputs, "masks"):
            raise AttributeError(self._NLL_INCOMPLETE)
        names = list(self.encoders.keys())
        with torch.no_grad():
            post = self._unimodal_posteriors(inputs, names)
            z = self.encode(inputs, N=int(K), noise=kwargs.get("noise"), sampled=kwargs.get("sampled")).z
            prior_sd = kernels.std_from_logvar(self.prior_log_var, self._family)
            return self._joint_nll(inputs, z, [post[m][0] for m in names], [post[m][1] for m in names],
                                   family=self._family, prior_loc=self.prior_mean, prior_sd=prior_sd)

    def generate_from_prior(self, n_samples, **kwargs):
        """n samples of the (learnable) prior `prior_dist(*pz_params)` (mmvae_model.py:470-474).  kwargs: noise [n, D]."""
        with torch.no_grad():
            mean, std = self.pz_params
            D = mean.shape[-1]
            n = max(int(n_samples), 1)
            noise = kwargs.get("noise")
            noise = self._noise((n, 1, D), mean.device, None if noise is None else noise.reshape(n, 1, D),
                                uniform=self._family == 1)
            z = kernels.iwae_sample(mean.detach().reshape(1, D), std.detach().reshape(1, D), noise, self._family)
        return ModelOutput(z=z.reshape(n, D).squeeze() if n_samples > 1 else z.reshape(D), one_latent_space=True)

    def compute_joint_nll_paper(self, inputs, K: int = 1000, batch_size_K: int = 10, **kwargs):
        """The original implementation's estimator (mmvae_model.py:444-468): for every chunk of batch_size_K samples
        per modality, the forward pass's (rescaled) importance weights of all modalities are pooled per data point
        (`iwae`, :294-311: logsumexp over samples and modalities - ln(n M)), SUMMED over the batch, shifted by
        ln(n M), and the chunk values are combined by one more logsumexp - ln(K M).  Unlike compute_joint_nll the
        result depends on the chunk size (the logsumexp acts on batch sums).  kwargs: noise = list (one entry per
        chunk) of {modality: [n,B,L]}."""
        self.eval()
        noise = kwargs.get("noise")
        M = self.n_modalities
        vals, done, c = [], 0, 0
        with torch.no_grad():
            while done < K:
                n = min(int(batch_size_K), int(K) - done)
                done += n
                out = self.forward(inputs, K=n, detailed_output=True, noise=None if noise is None else noise[c])
                ll = kernels.iwae_reduce([out["lws"][m] for m in out["lws"]])  # [B]: pooled over modalities and samples
                vals.append(ll.sum() + float(np.log(n * M)))
                c += 1
            return -(torch.logsumexp(torch.stack(vals), dim=0) - float(np.log(done * M)))
