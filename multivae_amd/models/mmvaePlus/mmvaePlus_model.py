"""MMVAE+ (Palumbo et al. 2023) on the HIP kernels.  Mirrors `multivae/models/mmvaePlus/mmvaePlus_model.py`:
_compute_posteriors_and_embeddings :122-187, _compute_k_lws :214-270, _dreg_looser :272-342, _iwae_looser :344-362.

Every modality has a shared latent u (latent_dim) and a private latent w (modalities_specific_dim); they are handled
as ONE concatenated latent z = [u, w] by the MMVAE kernels (csrc/mmvae.hip): the mixture-of-experts density covers
the first latent_dim dimensions, the private ones are scored by the modality's own posterior, the prior terms are
weighted by beta, and cross-modal decoder inputs take their private part from the target modality's learnable prior.
"""
import torch
from torch import nn

from ... import _lib, kernels
from ...data.utils import drop_unused_modalities
from ..base import BaseMultiVAE
from ..base.base_utils import ModelOutput
from ..nn.default_architectures import BaseDictDecodersMultiLatents, BaseDictEncoders_MultiLatents
from .mmvaePlus_config import MMVAEPlusConfig


class MMVAEPlus(BaseMultiVAE):
    def __init__(self, model_config: MMVAEPlusConfig, encoders: dict = None, decoders: dict = None):
        if model_config.modalities_specific_dim is None:
            raise AttributeError("The modalities_specific_dim attribute must be provided in the model config.")
        super().__init__(model_config, encoders, decoders)
        if model_config.prior_and_posterior_dist not in ("laplace_with_softmax", "normal", "normal_with_softplus"):
            raise AttributeError(" The posterior_dist parameter must be  either 'laplace_with_softmax','normal' or "
                                 f"'normal_with_softplus'.  {model_config.prior_and_posterior_dist} was provided.")
        S, L = model_config.modalities_specific_dim, model_config.latent_dim
        self.mean_priors = nn.ParameterDict()
        self.logvars_priors = nn.ParameterDict()
        self.beta = model_config.beta
        self.modalities_specific_dim = S
        self.reconstruction_option = model_config.reconstruction_option
        self.multiple_latent_spaces = True
        self.style_dims = {m: S for m in self.encoders}
        for mod in list(self.encoders.keys()):
            self.mean_priors[mod] = nn.Parameter(torch.zeros(1, S), requires_grad=False)
            self.logvars_priors[mod] = nn.Parameter(torch.zeros(1, S), requires_grad=model_config.learn_modality_prior)
        self.mean_priors["shared"] = nn.Parameter(torch.zeros(1, L + S), requires_grad=False)
        self.logvars_priors["shared"] = nn.Parameter(torch.zeros(1, L + S), requires_grad=model_config.learn_shared_prior)
        self.model_name = "MMVAEPlus"
        self.objective = model_config.loss

    def default_encoders(self, model_config) -> nn.ModuleDict:
        return BaseDictEncoders_MultiLatents(input_dims=model_config.input_dims, latent_dim=model_config.latent_dim,
                                             modality_dims={m: model_config.modalities_specific_dim
                                                            for m in model_config.input_dims})

    def default_decoders(self, model_config) -> nn.ModuleDict:
        return BaseDictDecodersMultiLatents(input_dims=model_config.input_dims, latent_dim=model_config.latent_dim,
                                            modality_dims={m: model_config.modalities_specific_dim
                                                           for m in model_config.input_dims})

    @property
    def _std_family(self):
        return _lib.FAMILY[self.model_config.prior_and_posterior_dist]

    @property
    def _family(self):  # density / sampling family of the latent kernels
        return 1 if self.model_config.prior_and_posterior_dist == "laplace_with_softmax" else 0

    def _log_var_to_std(self, log_var):
        return kernels.MMVAEStdFn.apply(log_var, self._std_family)

    @property
    def pz_params(self):
        return self.mean_priors["shared"], self._log_var_to_std(self.logvars_priors["shared"])

    def forward(self, inputs, **kwargs):
        """kwargs: K, noise = {cond: {"u": [K,B,L], "w": [K,B,S], other modality: [K,B,S]}} (explicit noise in the
        reference's draw order), detailed_output."""
        inputs = drop_unused_modalities(inputs)
        K = int(kwargs.pop("K", self.model_config.K))
        noise = kwargs.pop("noise", None)
        if self.objective not in ("dreg_looser", "iwae_looser"):
            raise NotImplementedError()
        dreg = self.objective == "dreg_looser"
        mods = list(inputs.data.keys())
        M = len(mods)
        L, S = self.latent_dim, self.modalities_specific_dim
        D = L + S
        family = self._family
        uniform = family == 1

        def encode_one(m):
            out = self.encoders[m](inputs.data[m])
            mu = torch.cat([out.embedding, out.style_embedding], dim=-1)
            sd = torch.cat([self._log_var_to_std(out.log_covariance),
                            self._log_var_to_std(out.style_log_covariance)], dim=-1)
            return mu, sd

        order = self._branch_order(inputs, mods)
        enc = kernels.run_branches(order, encode_one, inputs.data[order[0]].device)
        mus = [enc[m][0] for m in mods]
        sds = [enc[m][1] for m in mods]
        B = mus[0].shape[0]
        device = mus[0].device
        noises, cross_noise = [], {}
        for c in mods:  # draw order of the reference: u, w, then one private-prior draw per other modality
            nu = self._noise((K, B, L), device, None if noise is None else noise[c]["u"], uniform=uniform)
            nw = self._noise((K, B, S), device, None if noise is None else noise[c]["w"], uniform=uniform)
            noises.append(torch.cat([nu, nw], dim=-1))
            for r in mods:
                if r != c:
                    cross_noise[(c, r)] = self._noise((K, B, S), device, None if noise is None else noise[c][r],
                                                      uniform=uniform)
        masks = None
        if hasattr(inputs, "masks"):
            masks = [inputs.masks[m].to(torch.bool).contiguous() for m in mods]
        state = kernels.MMVAEState()
        state.shared_dims, state.beta = L, float(self.beta)
        prior_mean, prior_std = self.pz_params
        zs = kernels.MMVAELatentFn.apply(state, noises, masks, prior_mean.detach(), family, int(dreg), prior_std,
                                         *mus, *sds)
        mod_prior_std = {r: self._log_var_to_std(self.logvars_priors[r]) for r in mods}

        def decode_all(r):  # every conditioning modality's latent through decoder r: ONE pass over the M * K * B stacked rows
            zins = []
            for ci, c in enumerate(mods):
                if r == c:
                    zin = zs[ci]
                else:
                    zin = kernels.MMVAEPlusCrossLatentFn.apply(zs[ci], mod_prior_std[r], cross_noise[(c, r)], L, family)
                zins.append(zin.reshape(-1, D))
            # (the reference decodes every (conditioning, target) pair on its own, mmvaePlus_model.py:172-186: M^2 decoder passes
            # of K * B rows; the rows are independent, so M passes of M * K * B rows give the same reconstructions with 1 / M of
            # the launches and M times longer weight-gradient reductions per launch)
            # — for decoders that declare their rows independent (BaseDecoder.rows_independent: every in-package one); a
            # user-written decoder is run pair by pair like in the reference
            if not getattr(self.decoders[r], "rows_independent", False):
                return [self.decoders[r](zin).reconstruction for zin in zins]
            rec = self.decoders[r](torch.cat(zins, dim=0)).reconstruction
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

    def encode(self, inputs, cond_mod="all", N: int = 1, return_mean=False, **kwargs):
        """mmvaePlus_model.py:364-456: shared latent from a randomly chosen conditioning modality (or the mean of the
        means), private latents from the own posterior for conditioning modalities, from the priors for the others."""
        cond_mod = super().encode(inputs, cond_mod, N, **kwargs).cond_mod
        flatten = kwargs.pop("flatten", False)
        family = self.model_config.prior_and_posterior_dist
        with torch.no_grad():
            outs = {m: self.encoders[m](inputs.data[m]) for m in cond_mod}
            B = len(list(inputs.data.values())[0])
            device = next(iter(outs.values())).embedding.device
            shape = (N,) if N > 1 else ()

            def sample(mu, sd):
                if family == "laplace_with_softmax":
                    return torch.distributions.Laplace(mu, sd).rsample(shape)
                return torch.distributions.Normal(mu, sd).rsample(shape)

            if return_mean:
                emb = torch.mean(torch.stack([o.embedding for o in outs.values()]), dim=0)
                z = torch.stack([emb] * N) if N > 1 else emb
            else:
                pick = cond_mod[int(torch.randint(len(cond_mod), (1,)))]
                z = sample(outs[pick].embedding, self._log_var_to_std(outs[pick].log_covariance))
            style = {}
            for m in self.encoders:
                if m in cond_mod and not return_mean:
                    style[m] = sample(outs[m].style_embedding, self._log_var_to_std(outs[m].style_log_covariance))
                elif m in cond_mod:
                    style[m] = torch.stack([outs[m].style_embedding] * N) if N > 1 else outs[m].style_embedding
                else:
                    mu = torch.cat([self.mean_priors[m]] * B, dim=0).to(device)
                    sd = torch.cat([self._log_var_to_std(self.logvars_priors[m])] * B, dim=0).to(device)
                    style[m] = sample(mu, sd)
            if flatten and N > 1:
                z = z.reshape(-1, z.shape[-1])
                style = {m: v.reshape(-1, v.shape[-1]) for m, v in style.items()}
        return ModelOutput(z=z, one_latent_space=False, modalities_z=style)

    def compute_joint_nll(self, inputs, K: int = 1000, batch_size_K: int = 100, **kwargs):
        """-sum_b ln p(x_b) from the pooled importance weights of the forward pass (mmvaePlus_model.py:477-531):
        K // n_modalities samples per conditioning modality, rescale factors and beta forced to 1, and
        ln p(x_b) ~= logsumexp over all conditioning modalities and samples - ln(count).

        The reference evaluates this on the inputs WITHOUT their last modality (`inputs.data.popitem()`, :497, also
        removes it from the caller's dataset) while the mixture normaliser stays ln(n_modalities); the default here
        returns that number (without touching the caller's inputs).  kwargs: all_modalities=True conditions on and
        scores every modality instead; noise as in forward() with K -> K // n_modalities."""
        self.eval()
        if hasattr(inputs, "masks"):
            raise AttributeError(self._NLL_INCOMPLETE)
        from ...data.datasets.base import DatasetOutput
        import math

        mods = list(inputs.data.keys())
        if not kwargs.get("all_modalities", False):
            mods = mods[:-1]
        if not mods:
            raise AttributeError("compute_joint_nll needs at least two modalities (the reference drops the last one).")
        k = int(K) // self.n_modalities
        noise = kwargs.get("noise")
        B = inputs.data[mods[0]].shape[0]
        device = inputs.data[mods[0]].device
        ll = torch.empty(B, dtype=torch.float32, device=device)
        step = max(1, kernels.IWAE_ROWS_BUDGET // max(k, 1))
        rescale, beta = self.rescale_factors, self.beta
        self.rescale_factors, self.beta = {m: 1.0 for m in rescale}, 1.0
        try:
            with torch.no_grad():
                for b0 in range(0, B, step):
                    b1 = min(B, b0 + step)
                    part = DatasetOutput(data={m: inputs.data[m][b0:b1] for m in mods})
                    nz = None
                    if noise is not None:
                        nz = {c: {key: v[:, b0:b1] for key, v in noise[c].items()} for c in mods}
                    out = self.forward(part, K=k, noise=nz, detailed_output=True)
                    kernels.iwae_reduce([out["lws"][m] for m in mods], out=ll[b0:b1])
        finally:
            self.rescale_factors, self.beta = rescale, beta
        # the latent kernel normalises the mixture by the number of modalities it was given; the reference by
        # n_modalities (mmvaePlus_model.py:239): a constant shift of every log-weight
        shift = math.log(self.n_modalities) - math.log(len(mods))
        return -(ll.sum() + B * shift)

    def generate_from_prior(self, n_samples, **kwargs):
        """n samples of the (learnable) prior `prior_dist(*pz_params)` (mmvaePlus_model.py:453-456).  kwargs: noise [n, D]."""
        with torch.no_grad():
            mean, std = self.pz_params
            D = mean.shape[-1]
            n = max(int(n_samples), 1)
            noise = kwargs.get("noise")
            noise = self._noise((n, 1, D), mean.device, None if noise is None else noise.reshape(n, 1, D),
                                uniform=self._family == 1)
            z = kernels.iwae_sample(mean.detach().reshape(1, D), std.detach().reshape(1, D), noise, self._family)
        return ModelOutput(z=z.reshape(n, D).squeeze() if n_samples > 1 else z.reshape(D), one_latent_space=True)
