"""DMVAE (Lee & Pavlovic 2021) on the HIP kernels.  Mirrors `multivae/models/dmvae/dmvae_model.py`:
_infer_latent_parameters :103-150, forward :152-195, _compute_elbo :197-240, encode :242-296, generate_from_prior :298-320.

M + 1 ELBOs per step (the joint posterior = stable_poe of the available shared experts and the prior, and every
modality's own shared posterior), each with fresh private samples of every modality.  Here the M + 1 shared samples are
the leading axis of ONE decoder pass per modality (`[M+1, B, L + S_m]`): `mvk_mvae_posterior_fwd/bwd` gives the joint
sample and its KL, `mvk_gauss_sample_kl_fwd/bwd` the unimodal shared samples (+ KL) and the M + 1 private samples of a
modality in one launch (+ its private KL), and the reconstruction kernel scores every (ELBO, modality) slab as one term.
"""
from typing import Union

import torch
from torch import nn

from ... import kernels
from ..base import BaseMultiVAE
from ..base.base_utils import ModelOutput
from ..nn.default_architectures import BaseDictDecodersMultiLatents, BaseDictEncoders_MultiLatents
from .dmvae_config import DMVAEConfig


class DMVAE(BaseMultiVAE):
    def __init__(self, model_config: DMVAEConfig, encoders: dict = None, decoders: dict = None):
        super().__init__(model_config, encoders, decoders)
        self.beta = model_config.beta
        self.model_name = "DMVAE"
        self._set_private_betas(model_config.modalities_specific_betas)
        self._set_modalities_specific_dim(model_config)
        self.multiple_latent_spaces = True

    def _set_modalities_specific_dim(self, model_config):
        if model_config.modalities_specific_dim is None:
            self.style_dims = {m: 1.0 for m in self.encoders}
        else:
            if model_config.modalities_specific_dim.keys() != self.encoders.keys():
                raise AttributeError("The keys in modalities_specific_dim doesn't match ",
                                     "the keys in the encoders or input_dims")
            self.style_dims = model_config.modalities_specific_dim

    def _set_private_betas(self, beta_dict):
        if beta_dict is None:
            self.private_betas = {mod: 1.0 for mod in self.encoders}
        else:
            if not self.encoders.keys() == beta_dict.keys():
                raise AttributeError("The modality_specific_betas doesn't have the same keys (modalities) as the provided "
                                     "encoders dict.")
            self.private_betas = beta_dict

    def default_encoders(self, model_config) -> nn.ModuleDict:
        return BaseDictEncoders_MultiLatents(input_dims=model_config.input_dims, latent_dim=model_config.latent_dim,
                                             modality_dims=model_config.modalities_specific_dim)

    def default_decoders(self, model_config) -> nn.ModuleDict:
        return BaseDictDecodersMultiLatents(input_dims=model_config.input_dims, latent_dim=model_config.latent_dim,
                                            modality_dims=model_config.modalities_specific_dim)

    # -- posterior parameters ---------------------------------------------------------------------------------------
    def _encode_all(self, inputs, subset):
        order = self._branch_order(inputs, subset)
        enc = kernels.run_branches(order, lambda m: self.encoders[m](inputs.data[m]), inputs.data[order[0]].device)

        def two_d(t):
            return t if t.dim() == 2 else t.unsqueeze(0)

        shared = {m: (two_d(enc[m].embedding), two_d(enc[m].log_covariance)) for m in subset}
        private = {m: (two_d(enc[m].style_embedding), two_d(enc[m].style_log_covariance)) for m in subset}
        return shared, private

    def _joint(self, inputs, subset, shared, eps, want_stats=False):
        """stable_poe of the available shared experts of `subset` and the N(0,I) prior (:131-148): one subset of the MVAE
        posterior kernel.  Returns (per-modality copies of the sample [1,B,L], kld [1,B][, mu, lv])."""
        masks = None
        if hasattr(inputs, "masks"):
            masks = [inputs.masks[m].to(torch.bool).contiguous() for m in subset]
        bits = [(1 << len(subset)) - 1]
        return kernels.MVAEPosteriorFn.apply(eps, masks, bits, want_stats, *[shared[m][0] for m in subset],
                                             *[shared[m][1] for m in subset])

    def _infer_latent_parameters(self, inputs, subset=None):
        subset = list(inputs.data.keys()) if subset is None else list(subset)
        shared, private = self._encode_all(inputs, subset)
        B, L = shared[subset[0]][0].shape
        outs = self._joint(inputs, subset, shared, torch.zeros(1, B, L, device=shared[subset[0]][0].device), True)
        return outs[-2][0], outs[-1][0], shared, private

    # -- forward ----------------------------------------------------------------------------------------------------
    def forward(self, inputs, **kwargs) -> ModelOutput:
        """kwargs: noise = {"shared": [M+1,B,L], "private": {m: [M+1,B,S_m]}}: slab 0 belongs to the joint ELBO, slab
        1 + k to the ELBO of the k-th modality (the reference draws, per ELBO, the shared sample and then one private
        sample per modality in encoder order)."""
        noise = kwargs.pop("noise", None)
        mods = list(inputs.data.keys())
        names = list(self.encoders.keys())
        M = len(mods)
        shared, private = self._encode_all(inputs, mods)
        B, L = shared[mods[0]][0].shape
        device = shared[mods[0]][0].device
        E = M + 1
        sh_noise = self._noise((E, B, L), device, None if noise is None else noise["shared"])
        masks = inputs.masks if hasattr(inputs, "masks") else None
        # shared samples and KLs of the E ELBOs
        jouts = self._joint(inputs, mods, shared, sh_noise[:1].contiguous())
        z_joint, kl_joint = jouts[:M], jouts[M]  # M identical copies [1,B,L] (one per decoder), KL rows [1,B]
        z_uni, kl_uni = [], []
        for k, m in enumerate(mods):
            zk, klk = kernels.GaussSampleKLFn.apply(sh_noise[1 + k:2 + k].contiguous(), *shared[m])
            z_uni.append(zk)
            kl_uni.append(klk if masks is None else klk * masks[m].to(klk.dtype))  # mod_elbo *= mask_k (:187-188)
        # private samples (E per modality, one launch) and private KLs
        w, kl_priv = {}, {}
        for m in names:
            pn = None if noise is None else noise["private"][m]
            w[m], kl_priv[m] = kernels.GaussSampleKLFn.apply(self._noise((E, B, private[m][0].shape[-1]), device, pn),
                                                            *private[m])
        dnames = [m for m in self.decoders.keys()]

        def decode(m):
            zs = torch.cat([z_joint[mods.index(m)]] + z_uni, dim=0)  # [E,B,L]
            return self.decoders[m](torch.cat([zs, w[m]], dim=-1)).reconstruction

        rec = kernels.run_branches(self._branch_order(inputs, dnames), decode, device)
        # one reconstruction term per (ELBO e, modality m); rows count when x_m is there and (e >= 1) x_e is there
        pairs, pair_mod, pair_e, pmasks = [], [], [], []
        for e in range(E):
            for i, m in enumerate(dnames):
                pairs.append((i, e))
                pair_mod.append(m)
                pair_e.append(e)
                if masks is not None:
                    mk = masks[m].bool()
                    pmasks.append((mk if e == 0 else mk & masks[mods[e - 1]].bool()).contiguous())
        spec = self._recon_spec(pair_mod, inputs.data, None, 1, B)
        if masks is not None:
            spec["masks"] = pmasks
        P = len(pairs)
        # private KL of modality m enters every ELBO that counts for the row: weight mask_m * (1 + sum_k mask_k)
        if masks is None:
            priv = [kl_priv[m] for m in names]
            priv_w = [float(self.private_betas[m]) * E for m in names]
        else:
            n_elbo = 1.0 + torch.stack([masks[m].float() for m in mods]).sum(0)
            priv = [kl_priv[m] * (masks[m].float() * n_elbo) for m in names]
            priv_w = [float(self.private_betas[m]) for m in names]
        beta = float(self.beta)
        spec.update(pairs=pairs, coef=[1.0 / B] * P, lossw=[1.0] * P,
                    extra_coef=[1.0 / B] * (1 + M + len(names)), extra_lossw=[beta] * (1 + M) + priv_w,
                    loss_sum_scale=1.0)
        loss, terms = kernels.ReconLossFn.apply(spec, len(dnames), *[rec[m] for m in dnames], kl_joint, *kl_uni, *priv)
        # metrics: mean over the batch of every (masked) ELBO
        with torch.no_grad():
            def elbo_mean(e):
                v = sum(terms[i] for i in range(P) if pair_e[i] == e) + beta * terms[P + e]
                for m in names:
                    k = kl_priv[m]
                    if masks is not None:
                        k = k * masks[m].float() * (1.0 if e == 0 else masks[mods[e - 1]].float())
                    v = v + float(self.private_betas[m]) * k.mean()
                return v

            metrics = {"joint": elbo_mean(0)}
            for k, m in enumerate(mods):
                metrics[m] = elbo_mean(1 + k)
        return ModelOutput(loss=loss, metrics=metrics)

    # -- inference helpers ------------------------------------------------------------------------------------------
    def encode(self, inputs, cond_mod: Union[list, str] = "all", N: int = 1, return_mean=False, **kwargs):
        cond_mod = super().encode(inputs, cond_mod, N, **kwargs).cond_mod
        flatten = kwargs.pop("flatten", False)

        def sample(mu, lv):
            if return_mean:
                z = torch.stack([mu] * N) if N > 1 else mu
            else:
                shape = (N, *mu.shape) if N > 1 else mu.shape
                z = mu + torch.exp(0.5 * lv) * torch.randn(shape, device=mu.device)
            return z.reshape(-1, z.shape[-1]) if (N > 1 and flatten) else z

        with torch.no_grad():
            mu, lv, _, private = self._infer_latent_parameters(inputs, cond_mod)
            z = sample(mu, lv)
            modalities_z = {}
            for m in self.encoders:
                if m in cond_mod:
                    pm, pl = private[m]
                else:
                    pm = torch.zeros((mu.shape[0], int(self.style_dims[m])), device=mu.device)
                    pl = torch.zeros_like(pm)
                modalities_z[m] = sample(pm, pl)
        return ModelOutput(z=z, one_latent_space=False, modalities_z=modalities_z)

    def generate_from_prior(self, n_samples, **kwargs):
        dev = next(self.parameters()).device
        shape = [n_samples, self.latent_dim] if n_samples > 1 else [self.latent_dim]
        modalities_z = {}
        for k, dim in self.style_dims.items():
            modalities_z[k] = torch.randn([n_samples, int(dim)] if n_samples > 1 else [int(dim)], device=dev)
        return ModelOutput(z=torch.randn(shape, device=dev), one_latent_space=False, modalities_z=modalities_z)

    def compute_joint_nll(self, inputs, K: int = 1000, batch_size_K: int = 100, **kwargs):
        """The reference's estimator (dmvae_model.py:311-412), quirk included: K shared samples per data point from the
        joint posterior, fresh private samples of every modality per chunk of `batch_size_K`, and `ln_prior` /
        `ln_posterior` that are NEVER reset (:352, :385-403 accumulate them over the K-chunks AND over the data points), so
        the log-weight of sample r of chunk c of data point i carries the prior / posterior log-densities of sample r of
        every earlier (data point, chunk).  Here the K axis is a kernel axis (mvk_iwae_sample, one decoder pass per
        modality and chunk of data points, mvk_recon_nll_fwd rows, mvk_iwae_logw for ln p - ln q of the concatenated
        [shared, private...] latent, mvk_iwae_reduce); the running sums are one cumulative sum over the (data point,
        chunk) sequence.  kwargs: noise = {"shared": [K,B,L], "private": {m: [K,B,S_m]}} (sample k = c * batch_size_K + r)."""
        self.eval()
        if hasattr(inputs, "masks"):
            raise AttributeError(self._NLL_INCOMPLETE)
        K, bK = int(K), min(int(batch_size_K), int(K))
        if K % bK:  # the reference adds a [K % bK] chunk to its [bK] running sums and fails
            raise RuntimeError(f"K = {K} is not a multiple of batch_size_K = {bK}: the running prior / posterior sums of "
                               "the reference's estimator have one entry per sample of a chunk")
        noise = kwargs.get("noise") or {}
        with torch.no_grad():
            mu, lv, _, private = self._infer_latent_parameters(inputs)
            B, L = mu.shape
            dev = mu.device
            names = list(inputs.data.keys())
            z = kernels.iwae_sample(mu, kernels.std_from_logvar(lv), self._noise((K, B, L), dev, noise.get("shared")))
            ws, locs, sds = {}, [mu], [kernels.std_from_logvar(lv)]
            for m in names:
                pm, pl = private[m]
                psd = kernels.std_from_logvar(pl)
                ws[m] = kernels.iwae_sample(pm, psd, self._noise((K, B, pm.shape[-1]), dev,
                                                                   (noise.get("private") or {}).get(m)))
                locs.append(pm)
                sds.append(psd)
            loc_all, sd_all = torch.cat(locs, dim=-1).contiguous(), torch.cat(sds, dim=-1).contiguous()
            xs = [inputs.data[m].float().contiguous() for m in names]
            dists = [self.recon_dists[m][0] for m in names]
            scales = [self.recon_dists[m][1] for m in names]
            order = self._branch_order(inputs, names)
            nc = K // bK
            ll = torch.empty(B, dtype=torch.float32, device=dev)
            carry = torch.zeros(bK, dtype=torch.float32, device=dev)
            step = max(1, kernels.IWAE_ROWS_BUDGET // K)
            for b0 in range(0, B, step):
                b1 = min(B, b0 + step)
                b = b1 - b0
                zc = z[:, b0:b1]
                wc = {m: ws[m][:, b0:b1] for m in names}
                rec = kernels.run_branches(order, lambda m: self.decoders[m](
                    torch.cat([zc, wc[m]], dim=-1).contiguous()).reconstruction, dev)
                rows = kernels.recon_nll_rows([rec[m] for m in names], [x[b0:b1] for x in xs], dists, scales, K, b)
                z_all = torch.cat([zc] + [wc[m] for m in names], dim=-1).contiguous()
                ratio = kernels.iwae_logw(z_all, [], [loc_all[b0:b1]], [sd_all[b0:b1]])  # ln p - ln q, [K, b]
                # running sums over the (data point, chunk) sequence, per sample slot r of a chunk
                t = ratio.view(nc, bK, b).permute(2, 0, 1).reshape(b * nc, bK)
                run = torch.cumsum(t, dim=0) + carry
                carry = run[-1].clone()
                lw = run.view(b, nc, bK).permute(1, 2, 0).reshape(K, b).contiguous()
                for r in rows:
                    lw = kernels.axpby(lw, 1.0, r, -1.0)
                kernels.iwae_reduce([lw], out=ll[b0:b1])
            return -ll.sum()
