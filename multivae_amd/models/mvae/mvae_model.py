"""MVAE (Wu & Goodman 2018) on the HIP kernels.  Mirrors `multivae/models/mvae/mvae_model.py`:
subsets :47-51, compute_mu_log_var_subset :53-84, _compute_elbo_subset :86-118, forward :145-228, encode :230-264,
compute_joint_nll :266-340.

forward = M encoder nodes (ONCE, the reference re-runs them for every subset) -> ONE posterior kernel for all subsets
of the objective (product of the available experts and the prior, one sample and one KL per subset) -> M decoder nodes,
each over the samples of all the subsets it belongs to -> the fused reconstruction-NLL kernel with one term per
(modality, subset) -> ONE scalar assembly kernel.
"""
from itertools import combinations
from typing import Union

import numpy as np
import torch

from ... import kernels
from ..base import BaseMultiVAE
from ..base.base_utils import ModelOutput
from .mvae_config import MVAEConfig


class MVAE(BaseMultiVAE):
    def __init__(self, model_config: MVAEConfig, encoders: dict = None, decoders: dict = None):
        super().__init__(model_config, encoders, decoders)
        self.subsampling = model_config.use_subsampling
        self.k = model_config.k
        if self.n_modalities <= 2:
            self.k = 0
        self._set_subsets()
        self.warmup = model_config.warmup
        self.start_keep_best_epoch = model_config.warmup + 1
        self.beta = model_config.beta
        self.model_name = "MVAE"

    def _set_subsets(self):
        self.subsets = []
        for i in range(2, self.n_modalities):
            self.subsets += combinations(list(self.encoders.keys()), r=i)

    def graph_key(self, epoch=1, batch_ratio=0, **kwargs):
        """The annealing factor is a host constant of the captured kernels and the random subsets change the launch
        sequence: only the post-warm-up, k = 0 step is replayable (False = run eagerly)."""
        if epoch >= self.warmup and self.k == 0:
            return "annealed"
        return False

    # -- posterior of a list of subsets -----------------------------------------------------------------------------
    def _bits(self, subset):
        pos = {m: i for i, m in enumerate(self.encoders.keys())}
        return sum(1 << pos[m] for m in subset)

    def _posterior(self, inputs, subsets, noise=None, want_stats=False):
        names = list(self.encoders.keys())
        used = [m for m in names if any(m in s for s in subsets)]
        order = self._branch_order(inputs, used)
        enc = kernels.run_branches(order, lambda m: self.encoders[m](inputs.data[m]), inputs.data[order[0]].device)
        ref = enc[used[0]].embedding
        if ref.dim() == 1:
            ref = ref.unsqueeze(0)
        B, L = ref.shape
        device = ref.device
        mus, lvs = [], []
        for m in names:  # modalities outside every subset still need a slot: they are never read
            if m in enc:
                mu, lv = enc[m].embedding, enc[m].log_covariance
                mus.append(mu if mu.dim() == 2 else mu.unsqueeze(0))
                lvs.append(lv if lv.dim() == 2 else lv.unsqueeze(0))
            else:
                mus.append(torch.zeros_like(ref))
                lvs.append(torch.zeros_like(ref))
        masks = None
        if hasattr(inputs, "masks"):
            masks = [inputs.masks[m].to(torch.bool).contiguous() if m in inputs.masks else None for m in names]
        eps = self._noise((len(subsets), B, L), device, noise)
        bits = [self._bits(s) for s in subsets]
        outs = kernels.MVAEPosteriorFn.apply(eps, masks, bits, want_stats, *mus, *lvs)
        return outs, (names, used, B, L, device)

    def compute_mu_log_var_subset(self, inputs, subset):
        """Posterior parameters when conditioning on `subset` (:53-84): stable_poe of its available experts and the
        prior."""
        outs, _ = self._posterior(inputs, [list(subset)], want_stats=True)
        return outs[-2][0], outs[-1][0]

    # -- forward ----------------------------------------------------------------------------------------------------
    def forward(self, inputs, **kwargs) -> ModelOutput:
        """kwargs: epoch, batch_ratio (annealing), noise [S,B,L] (one draw per subset in objective order: joint,
        unimodal ones, random ones), random_subsets (indices into self.subsets replacing the np.random.choice)."""
        epoch = kwargs.pop("epoch", 1)
        batch_ratio = kwargs.pop("batch_ratio", 0)
        noise = kwargs.pop("noise", None)
        if epoch >= self.warmup:
            beta = 1 * self.beta
        else:
            beta = (epoch - 1 + batch_ratio) / self.warmup * self.beta
        names = list(self.encoders.keys())
        subsets = [list(names)]
        if self.subsampling:
            subsets.extend([[m] for m in names])
            if self.k > 0 and self.training:
                idx = kwargs.pop("random_subsets", None)
                if idx is None:
                    idx = np.random.choice(np.arange(len(self.subsets)), size=self.k, replace=False)
                subsets.extend([list(self.subsets[int(i)]) for i in idx])
        S = len(subsets)
        outs, (names, used, B, L, device) = self._posterior(inputs, subsets, noise=noise)
        zs = dict(zip(names, outs[:len(names)]))  # every modality is in the joint subset
        kld_rows = outs[len(names)]
        dnames = [m for m in self.decoders.keys() if m in zs]
        rec = kernels.run_branches(self._branch_order(inputs, dnames), lambda m: self.decoders[m](zs[m]).reconstruction,
                                   device)
        # rows kept per subset (the reference filters the batch to samples with at least one modality of the subset,
        # :120-143) -- masks are inputs, one small host read per step in the incomplete-data case only
        masked = hasattr(inputs, "masks")
        if masked:
            valid = torch.stack([torch.stack([inputs.masks[m].bool() for m in s]).any(0) for s in subsets])
            n_rows = [int(v) for v in valid.sum(1).tolist()]
        else:
            n_rows = [B] * S
        inv = [1.0 / n if n > 0 else 0.0 for n in n_rows]
        slot = {m: 0 for m in dnames}
        pairs, pair_subset, pair_mod = [], [], []
        for si, s in enumerate(subsets):
            for m in dnames:
                if m in s:
                    pairs.append((dnames.index(m), slot[m]))
                    pair_subset.append(si)
                    pair_mod.append(m)
                    slot[m] += 1
        spec = self._recon_spec(pair_mod, inputs.data, inputs.masks if masked else None, 1, B)
        P = len(pairs)
        spec.update(pairs=pairs, coef=[inv[si] for si in pair_subset], lossw=[1.0] * P, extra_coef=[inv],
                    extra_lossw=[float(beta)], extra_split=[S], loss_sum_scale=float(n_rows[-1]))
        loss, terms = kernels.ReconLossFn.apply(spec, len(dnames), *[rec[m] for m in dnames], kld_rows)
        metrics = {}
        for si, s in enumerate(subsets):
            if n_rows[si] == 0:
                continue
            key = "_".join(sorted(s))
            recon = sum(terms[i] for i in range(P) if pair_subset[i] == si)
            kld = terms[P + si]
            metrics[key] = recon + beta * kld
            metrics["beta"] = beta
            metrics["kld" + key] = kld
            # the reference's `recon` aliases the tensor its in-place `elbo_sub += KLD * beta` updates
            # (mvae_model.py:99-102): its "recon" metric is the subset ELBO; same here
            metrics["recon" + key] = metrics[key]
        return ModelOutput(loss=loss, loss_sum=terms[P + S + 1], metrics=metrics)

    # -- inference helpers ------------------------------------------------------------------------------------------
    def encode(self, inputs, cond_mod: Union[list, str] = "all", N: int = 1, return_mean=False, **kwargs):
        cond_mod = super().encode(inputs, cond_mod, N, **kwargs).cond_mod
        flatten = kwargs.pop("flatten", False)
        with torch.no_grad():
            mu, log_var = self.compute_mu_log_var_subset(inputs, cond_mod)
            z = self._gaussian_encoding(mu, log_var, N, return_mean, flatten, kwargs.get("noise"))
        return ModelOutput(z=z, one_latent_space=True)

    def compute_joint_nll(self, inputs, K: int = 1000, batch_size_K: int = 100, **kwargs):
        """-sum_b ln p(x_b) by importance sampling from the joint posterior (:266-340).  kwargs: noise [K,B,L]."""
        self.eval()
        if hasattr(inputs, "masks"):
            raise AttributeError(self._NLL_INCOMPLETE)
        with torch.no_grad():
            mu, lv = self.compute_mu_log_var_subset(inputs, list(self.encoders.keys()))
            B, L = mu.shape
            sd = kernels.std_from_logvar(lv)
            z = kernels.iwae_sample(mu, sd, self._noise((int(K), B, L), mu.device, kwargs.get("noise")))
            return self._joint_nll(inputs, z, [mu], [sd])
