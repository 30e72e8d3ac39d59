"""MVTCAE on the HIP kernels.  Mirrors `multivae/models/mvtcae/mvtcae_model.py:42-169` (Appendix A.3):
PoE over the modalities (no prior expert), joint KL, per-modality conditional KLs, reconstruction SUMS;
loss = total / B, loss_sum = total, metrics are sums.  The three `assert not isnan` host syncs of the
reference (:55,75,97) are dropped (SURVEY.md Appendix D "may drop")."""
from typing import Union

import torch

from ... import kernels
from ..base import BaseMultiVAE
from ..base.base_utils import ModelOutput
from .mvtcae_config import MVTCAEConfig


class MVTCAE(BaseMultiVAE):
    def __init__(self, model_config: MVTCAEConfig, encoders: dict = None, decoders: dict = None):
        super().__init__(model_config, encoders, decoders)
        self.alpha = model_config.alpha
        self.beta = model_config.beta
        self.model_name = "MVTCAE"

    def _posterior(self, inputs, K, noise=None, mods=None):
        mods = list(inputs.data.keys()) if mods is None else mods
        order = self._branch_order(inputs, mods)
        enc = kernels.run_branches(order, lambda m: self.encoders[m](inputs.data[m]), inputs.data[order[0]].device)
        mus = [enc[m].embedding for m in mods]
        lvs = [enc[m].log_covariance for m in mods]
        if mus[0].dim() == 1:
            mus = [t.unsqueeze(0) for t in mus]
            lvs = [t.unsqueeze(0) for t in lvs]
        B, L = mus[0].shape
        device = mus[0].device
        masks = None
        if hasattr(inputs, "masks"):
            masks = [inputs.masks[m].to(torch.bool).contiguous() for m in mods]
        eps = self._noise((K, B, L), device, noise)
        outs = kernels.MVTCAEPosteriorFn.apply(eps, masks, *mus, *lvs)
        return enc, outs, (B, L, device)

    def forward(self, inputs, **kwargs) -> ModelOutput:
        K = int(kwargs.pop("K", self.model_config.K))
        noise = kwargs.pop("noise", None)
        if noise is not None and noise.dim() == 2:
            noise = noise.unsqueeze(0)
        mods = list(inputs.data.keys())
        _, outs, (B, L, device) = self._posterior(inputs, K, noise=noise)
        z, jkl, ckl = outs[0], outs[1], outs[2]
        names = list(self.encoders.keys())
        z_in = z[0] if K == 1 else z
        rec = kernels.run_branches(self._branch_order(inputs, names), lambda m: self.decoders[m](z_in).reconstruction, device)
        recons = [rec[m] for m in names]
        masks = inputs.masks if hasattr(inputs, "masks") else None
        spec = self._recon_spec(names, inputs.data, masks, K, B)
        M, Mn = len(mods), len(names)
        rec_w = (self.n_modalities - self.alpha) / self.n_modalities
        cvib_w = self.alpha / self.n_modalities
        vib_w = 1 - self.alpha
        spec.update(coef=[1.0 / K] * Mn, lossw=[rec_w / B] * Mn,
                    extra_coef=[1.0, 1.0], extra_lossw=[self.beta * vib_w / B, self.beta * cvib_w / B],
                    extra_split=[1, M], loss_sum_scale=float(B))
        loss, terms = kernels.ReconLossFn.apply(spec, Mn, *recons, jkl, ckl)
        metrics = {"joint_divergence": terms[Mn]}
        for i, m in enumerate(names):
            metrics[m] = terms[i]
        for j, m in enumerate(mods):
            metrics["kld_" + m] = terms[Mn + 1 + j]
        n_terms = Mn + 1 + M
        return ModelOutput(loss=loss, loss_sum=terms[n_terms + 1], metrics=metrics)

    def compute_joint_nll(self, inputs, K: int = 1000, batch_size_K: int = 100, **kwargs):
        """-sum_b ln p(x_b) by importance sampling from the PoE joint posterior (mvtcae_model.py:213-291).
        kwargs: noise [K,B,L]."""
        self.eval()
        if hasattr(inputs, "masks"):
            raise AttributeError(self._NLL_INCOMPLETE)
        with torch.no_grad():
            _, outs, _ = self._posterior(inputs, int(K), noise=kwargs.get("noise"))
            z, mu, lv = outs[0], outs[3], outs[4]
            return self._joint_nll(inputs, z, [mu], [kernels.std_from_logvar(lv)])

    def encode(self, inputs, cond_mod: Union[list, str] = "all", N: int = 1, return_mean=False, **kwargs):
        cond_mod = BaseMultiVAE.encode(self, inputs, cond_mod, N, **kwargs).cond_mod  # (CRMVAE borrows this method)
        from ...data.datasets.base import MultimodalBaseDataset

        cond_inputs = MultimodalBaseDataset(data={k: inputs.data[k] for k in cond_mod})
        with torch.no_grad():
            _, outs, _ = self._posterior(cond_inputs, 1, mods=list(cond_mod))
            mu, log_var = outs[3], outs[4]
            z = self._gaussian_encoding(mu, log_var, N, return_mean, kwargs.pop("flatten", False), kwargs.get("noise"))
        return ModelOutput(z=z, one_latent_space=True)
