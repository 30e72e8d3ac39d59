"""JMVAE (Suzuki et al. 2016) on the fused HIP path.  Mirrors `multivae/models/jmvae/jmvae_model.py`:
forward :116-192 (joint encoder -> z -> decoders -> recon + annealed (beta KLD + alpha LJM)), encode :58-114."""
from typing import Union

import torch

from ... import kernels
from ..base.base_utils import ModelOutput
from ..joint_models import BaseJointModel
from ..nn.base_architectures import BaseJointEncoder
from .jmvae_config import JMVAEConfig


class JMVAE(BaseJointModel):
    def __init__(self, model_config: JMVAEConfig, encoders: dict = None, decoders: dict = None,
                 joint_encoder: Union[BaseJointEncoder, None] = None, **kwargs):
        super().__init__(model_config, encoders, decoders, joint_encoder, **kwargs)
        self.model_name = "JMVAE"
        self.alpha = model_config.alpha
        self.warmup = model_config.warmup
        self.start_keep_best_epoch = model_config.warmup + 1
        self.beta = model_config.beta

    def graph_key(self, epoch=1, **kwargs):
        """What a captured training graph of this model depends on besides the batch shape: the annealing factor."""
        return min(int(epoch), int(self.warmup))

    def forward(self, inputs, **kwargs) -> ModelOutput:
        """kwargs: epoch (annealing factor = min(1, epoch / warmup)), noise [B,L] (explicit eps of the joint sample)."""
        super().forward(inputs)
        epoch = kwargs.pop("epoch", 1)
        noise = kwargs.pop("noise", None)
        names = list(self.encoders.keys())
        joint = self.joint_encoder(inputs.data)
        mu, lv = joint.embedding, joint.log_covariance
        B, L = mu.shape
        device = mu.device
        enc = kernels.run_branches(self._branch_order(inputs, names), lambda m: self.encoders[m](inputs.data[m]), device)
        mus = [enc[m].embedding for m in names]
        lvs = [enc[m].log_covariance for m in names]
        eps = self._noise((1, B, L), device, None if noise is None else noise.reshape(1, B, L))
        z, kld_rows, ljm_rows = kernels.JMVAEPosteriorFn.apply(eps, mu, lv, *mus, *lvs)
        dnames = list(self.decoders.keys())
        rec = kernels.run_branches(self._branch_order(inputs, dnames), lambda m: self.decoders[m](z[0]).reconstruction,
                                   device)
        recons = [rec[m] for m in dnames]
        spec = self._recon_spec(dnames, inputs.data, None, 1, B)
        a = 1.0 if epoch >= self.warmup else epoch / self.warmup
        M = len(dnames)
        spec.update(coef=[1.0 / B] * M, lossw=[1.0] * M, extra_coef=[1.0 / B, 1.0 / B],
                    extra_lossw=[a * float(self.beta), a * float(self.alpha)], loss_sum_scale=float(B))
        loss, terms = kernels.ReconLossFn.apply(spec, M, *recons, kld_rows, ljm_rows)
        # terms[i] = term_i / B.  Metrics (detached scalars, outside the hot path):
        recon = terms[:M].sum()
        kld, ljm = terms[M] * float(self.beta), terms[M + 1] * float(self.alpha)
        metrics = dict(loss_no_ponderation=(recon + kld + ljm) * B, beta=a, elbo=recon + kld)
        return ModelOutput(loss=loss, loss_sum=terms[M + 3], metrics=metrics)

    def encode(self, inputs, cond_mod: Union[list, str] = "all", N: int = 1, return_mean=False, **kwargs):
        self.eval()
        cond_mod = super().encode(inputs, cond_mod, N, **kwargs).cond_mod
        flatten = kwargs.pop("flatten", False)
        with torch.no_grad():
            if len(cond_mod) == self.n_modalities:
                out = self.joint_encoder(inputs.data)
                mu, lv = out.embedding, out.log_covariance
            elif len(cond_mod) != 1:
                mu, lv = self._poe_subset(cond_mod, inputs.data)
            else:
                out = self.encoders[cond_mod[0]](inputs.data[cond_mod[0]])
                mu, lv = out.embedding, out.log_covariance
            z = self._gaussian_encoding(mu, lv, N, return_mean, flatten, kwargs.get("noise"))
        return ModelOutput(z=z, one_latent_space=True)

    def _poe_subset(self, subset, data):
        """Product of the unimodal experts of a subset (stable_poe, base_utils.py:133-147): inference helper."""
        mus, lvs = [], []
        for mod in subset:
            o = self.encoders[mod](data[mod])
            mus.append(o.embedding)
            lvs.append(o.log_covariance)
        from ..base.base_utils import stable_poe

        return stable_poe(mus, lvs)  # mvk_poe_fwd (csrc/utils.hip)
