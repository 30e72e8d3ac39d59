"""CRMVAE (Suzuki & Matsuo 2023) on the HIP kernels.  Mirrors `multivae/models/crmvae/crmvae_model.py`:
forward :37-105, _modality_encode :107-131, _infer_all_latent_parameters :133-165, encode :167-201,
compute_joint_nll :203-295.

The aggregation is MVTCAE's (PoE of the available experts without a prior expert, KL(joint || prior) and
KL(joint || q_m) per modality: `mvk_mvtcae_posterior_fwd/bwd`); on top of it every modality is also reconstructed from a
sample of its OWN posterior (`mvk_gauss_sample_kl_fwd/bwd`).  Each decoder runs once on the two samples stacked, and the
reconstruction kernel scores the two slabs as separate terms.
"""
import torch

from ... import kernels
from ..base import BaseMultiVAE
from ..base.base_utils import ModelOutput
from ..mvtcae.mvtcae_model import MVTCAE
from .crmvae_config import CRMVAEConfig


class CRMVAE(BaseMultiVAE):
    def __init__(self, model_config: CRMVAEConfig, encoders: dict = None, decoders: dict = None):
        super().__init__(model_config, encoders, decoders)
        self.model_name = "CRMVAE"

    # the joint posterior, encode() and the likelihood estimator are MVTCAE's (same `poe` over the available experts)
    _posterior = MVTCAE._posterior
    encode = MVTCAE.encode
    compute_joint_nll = MVTCAE.compute_joint_nll

    def forward(self, inputs, **kwargs) -> ModelOutput:
        """kwargs: noise [B,L] (joint sample), modality_noise {m: [B,L]} (unimodal samples, drawn after the joint one
        in `inputs.data` order)."""
        noise = kwargs.pop("noise", None)
        mod_noise = kwargs.pop("modality_noise", None)
        if noise is not None and noise.dim() == 2:
            noise = noise.unsqueeze(0)
        mods = list(inputs.data.keys())
        enc, outs, (B, L, device) = self._posterior(inputs, 1, noise=noise)
        z, jkl, ckl = outs[0], outs[1], outs[2]
        M = self.n_modalities
        z_m = {}
        for m in mods:  # z_m ~ q(z | x_m) from the UNMASKED encoder output (:73-75)
            mu, lv = enc[m].embedding, enc[m].log_covariance
            if mu.dim() == 1:
                mu, lv = mu.unsqueeze(0), lv.unsqueeze(0)
            nz = None if mod_noise is None else mod_noise[m].reshape(1, B, L)
            z_m[m], _ = kernels.GaussSampleKLFn.apply(self._noise((1, B, L), device, nz), mu, lv)
        dnames = [m for m in self.decoders.keys() if m in z_m]
        rec = kernels.run_branches(self._branch_order(inputs, dnames),
                                   lambda m: self.decoders[m](torch.cat([z, z_m[m]], dim=0)).reconstruction, device)
        masks = inputs.masks if hasattr(inputs, "masks") else None
        pairs, pair_mod = [], []
        for i, m in enumerate(dnames):  # order of the reference's loops: for gen_mod: for m in ["joint", gen_mod]
            pairs += [(i, 0), (i, 1)]
            pair_mod += [m, m]
        spec = self._recon_spec(pair_mod, inputs.data, masks, 1, B)
        P = len(pairs)
        beta = float(self.model_config.beta)
        # term = mean over the batch (the metric); loss = sum_b [ sum recon / (2 (M+1)) + beta (sum KL) / (M+1) ]
        spec.update(pairs=pairs, coef=[1.0 / B] * P, lossw=[B / (2.0 * (M + 1))] * P, extra_coef=[1.0 / B, 1.0 / B],
                    extra_lossw=[beta * B / (M + 1)] * 2, extra_split=[1, len(mods)], loss_sum_scale=1.0)
        loss, terms = kernels.ReconLossFn.apply(spec, len(dnames), *[rec[m] for m in dnames], jkl, ckl)
        metrics = {"joint_divergence": terms[P]}
        for j, m in enumerate(mods):
            metrics[f"kl_{m}"] = terms[P + 1 + j]
        for i, m in enumerate(dnames):
            metrics[f"recon_{m}_from_joint"] = terms[2 * i]
            metrics[f"recon_{m}_from_{m}"] = terms[2 * i + 1]
        return ModelOutput(loss=loss, loss_sum=loss, metrics=metrics)
