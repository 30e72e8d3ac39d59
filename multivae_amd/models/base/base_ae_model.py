"""BaseMultiVAE: the plugin contract of `multivae/models/base/base_ae_model.py:24-443` (constructor checks,
encoders / decoders ModuleDicts, rescale factors, decoder distributions, encode / decode / predict).

The training arithmetic of the subclasses runs in HIP kernels (multivae_amd.kernels); the helper methods
here (encode / decode / predict / generate_from_prior) are evaluation conveniences outside the hot path.
"""
from copy import deepcopy
from typing import Union

import numpy as np
import torch
import torch.nn as nn

from ... import kernels
from ..nn.base_architectures import BaseDecoder, BaseEncoder
from ..nn.default_architectures import BaseDictDecoders, BaseDictEncoders
from .base_config import BaseMultiVAEConfig
from .base_model import BaseModel
from .base_utils import ModelOutput, decoder_dist_code


class BaseMultiVAE(BaseModel):
    def __init__(self, model_config: BaseMultiVAEConfig, encoders: dict = None, decoders: dict = None):
        super().__init__(model_config)
        self.model_name = "BaseMultiVAE"
        self.n_modalities = model_config.n_modalities
        self.input_dims = model_config.input_dims
        self.latent_dim = model_config.latent_dim
        self.device = None
        self.multiple_latent_spaces = False
        self.use_likelihood_rescaling = model_config.uses_likelihood_rescaling
        self.check_input_dims(model_config)

        if encoders is None:
            if self.input_dims is None:
                raise AttributeError("Please provide encoders or input dims for the modalities in the model_config.")
            encoders = self.default_encoders(model_config)
        else:
            self.model_config.custom_architectures.append("encoders")
        if decoders is None:
            if self.input_dims is None:
                raise AttributeError("Please provide decoders or input dims for the modalities in the model_config.")
            decoders = self.default_decoders(model_config)
        else:
            self.model_config.custom_architectures.append("decoders")

        self.sanity_check(encoders, decoders)
        self.set_decoders(decoders)  # decoders first: parameter order of the reference (base_ae_model.py:86-87)
        self.set_encoders(encoders)
        self.modalities_name = list(self.decoders.keys())
        self.rescale_factors = self.set_rescale_factors()

        if model_config.decoders_dist is None:
            model_config.decoders_dist = {k: "normal" for k in self.encoders}
        if model_config.decoder_dist_params is None:
            model_config.decoder_dist_params = {}
        self.set_decoders_dist(model_config.decoders_dist, deepcopy(model_config.decoder_dist_params))

    # -- configuration -------------------------------------------------------------------------------

    def _branch_order(self, inputs=None, names=None):
        """Modalities by decreasing input size: the first one keeps the caller's stream, the others get side streams
        (kernels.run_branches), so the large modality's kernels are never queued behind the small ones."""
        names = list(self.encoders.keys()) if names is None else list(names)
        dims = self.input_dims or {}

        def size(m):
            d = dims.get(m)
            if d is None and inputs is not None and m in inputs.data:
                d = tuple(inputs.data[m].shape[1:])
            n = 1
            for v in (d or ()):
                n *= int(v)
            return n

        return sorted(names, key=lambda m: -size(m))

    def set_decoders_dist(self, recon_dict, dist_params_dict):
        """Per-modality (distribution code, scale) consumed by mvk_recon_nll_* (base_utils.py:62-87)."""
        self.recon_dists = {}
        for k in recon_dict:
            params = dist_params_dict.get(k, {})
            code = decoder_dist_code(recon_dict[k])
            scale = float(params.get("scale", 1.0)) if recon_dict[k] in ("normal", "laplace") else 1.0
            self.recon_dists[k] = (code, scale)

    def check_input_dims(self, model_config):
        if model_config.input_dims is not None:
            if len(model_config.input_dims.keys()) != model_config.n_modalities:
                raise AttributeError(
                    f"The provided number of input_dims {len(model_config.input_dims)} doesn't"
                    f"match the number of modalities ({model_config.n_modalities} in model config ")

    def set_rescale_factors(self):
        if self.use_likelihood_rescaling:
            if self.model_config.rescale_factors is not None:
                return self.model_config.rescale_factors
            if self.input_dims is None:
                raise AttributeError(
                    " inputs_dim is None but (use_likelihood_rescaling = True in model_config)"
                    " To compute default likelihood rescalings we need the input dimensions."
                    " Please provide a valid dictionary for input_dims or provide rescale_factors"
                    " in the model_config.")
            max_dim = max(*[np.prod(self.input_dims[k]) for k in self.input_dims])
            return {k: max_dim / np.prod(self.input_dims[k]) for k in self.input_dims}
        return {k: 1 for k in self.encoders}

    def sanity_check(self, encoders, decoders):
        if self.n_modalities != len(encoders.keys()):
            raise AttributeError(
                f"The provided number of encoders {len(encoders.keys())} doesn't"
                f"match the number of modalities ({self.n_modalities} in model config ")
        if self.n_modalities != len(decoders.keys()):
            raise AttributeError(
                f"The provided number of decoders {len(decoders.keys())} doesn't"
                f"match the number of modalities ({self.n_modalities} in model config ")
        if encoders.keys() != decoders.keys():
            raise AttributeError("The names of the modalities in the encoders dict doesn't match the names of the "
                                 "modalities in the decoders dict.")
        if self.input_dims is not None and self.input_dims.keys() != encoders.keys():
            raise KeyError(
                f"Warning! : The modalities names in model_config.input_dims : {list(self.input_dims.keys())}"
                f" do not match the modalities names in encoders : {list(encoders.keys())}")

    def default_encoders(self, model_config) -> nn.ModuleDict:
        return BaseDictEncoders(self.input_dims, model_config.latent_dim)

    def default_decoders(self, model_config) -> nn.ModuleDict:
        return BaseDictDecoders(self.input_dims, model_config.latent_dim)

    def set_encoders(self, encoders: dict) -> None:
        self.encoders = nn.ModuleDict()
        for modality in encoders:
            if not issubclass(type(encoders[modality]), BaseEncoder):
                raise AttributeError(f"For modality {modality}, encoder must inherit from BaseEncoder class "
                                     "(multivae_amd.models.nn.base_architectures.BaseEncoder).")
            self.encoders[modality] = encoders[modality]

    def set_decoders(self, decoders: dict) -> None:
        self.decoders = nn.ModuleDict()
        for modality in decoders:
            if not issubclass(type(decoders[modality]), BaseDecoder):
                raise AttributeError(f"For modality {modality}, decoder must inherit from BaseDecoder class "
                                     "(multivae_amd.models.nn.base_architectures.BaseDecoder).")
            self.decoders[modality] = decoders[modality]

    # -- evaluation helpers (outside the training hot path) -----------------------------------------------
    def encode(self, inputs, cond_mod: Union[list, str] = "all", N: int = 1, return_mean=False, **kwargs):
        if isinstance(cond_mod, str):
            if cond_mod == "all":
                cond_mod = list(self.encoders.keys())
            elif cond_mod in self.encoders.keys():
                cond_mod = [cond_mod]
            else:
                raise AttributeError('If cond_mod is a string, it must either be "all" or a modality name'
                                     f" The provided string {cond_mod} is neither.")
        ignore_incomplete = kwargs.pop("ignore_incomplete", False)
        if hasattr(inputs, "masks") and not ignore_incomplete:
            avail = None
            for m in cond_mod:
                avail = inputs.masks[m] if avail is None else torch.logical_and(avail, inputs.masks[m])
            if not bool(torch.all(avail)):
                raise AttributeError("You tried to encode a incomplete dataset conditioning on",
                                     f"modalities {cond_mod}, but some samples are not available"
                                     "in all those modalities.")
        return ModelOutput(cond_mod=cond_mod, z=None, one_latent_space=None)

    def decode(self, embedding: ModelOutput, modalities: Union[list, str] = "all"):
        self.eval()
        with torch.no_grad():
            if modalities == "all":
                modalities = list(self.decoders.keys())
            elif isinstance(modalities, str):
                modalities = [modalities]
            try:
                outputs = ModelOutput()
                for m in modalities:
                    z = embedding.z
                    if not embedding.one_latent_space:
                        z = torch.cat([z, embedding.modalities_z[m]], dim=-1)
                    outputs[m] = self.decoders[m](z).reconstruction
                return outputs
            except Exception as e:
                raise ValueError("There was an error during decode. Check that the format for the embedding is "
                                 "correct: it must be a ModelOuput instance and embedding.z must be a Tensor of "
                                 "shape (batch_size, *latent_shape). If you used the encode function with N>1 you "
                                 "need to pass flatten=True to have the right format for decoding.") from e

    def predict(self, inputs, cond_mod="all", gen_mod="all", N: int = 1, flatten: bool = False, **kwargs):
        self.eval()
        ignore_incomplete = kwargs.pop("ignore_incomplete", False)
        z = self.encode(inputs, cond_mod, N=N, flatten=True, ignore_incomplete=ignore_incomplete, **kwargs)
        output = self.decode(z, gen_mod)
        n_data = len(z.z) // N
        if not flatten and N > 1:
            for m in output.keys():
                output[m] = output[m].reshape(N, n_data, *output[m].shape[1:])
        return output

    def forward(self, inputs, **kwargs) -> ModelOutput:
        raise NotImplementedError()

    def compute_joint_nll(self, inputs, K: int = 1000, batch_size_K: int = 100):
        raise NotImplementedError

    def compute_cond_nll(self, inputs, subset, pred_mods, k_iwae=1000, **kwargs):
        """-mean_b [ logsumexp_k ln p(x_pred,b | z_kb) - ln K ], z_kb ~ q(z | x_subset,b), per modality of pred_mods
        (base_ae_model.py:396-442).  The reference calls encode() + decode() k_iwae times; here the k_iwae samples of
        encode(N = k_iwae) are the leading axis of one decoder pass per chunk of data points (mvk_recon_nll_fwd rows,
        mvk_iwae_reduce).  kwargs: noise [K,B,L] (forwarded to encode)."""
        if isinstance(pred_mods, str):
            pred_mods = [pred_mods]
        K = int(k_iwae)
        with torch.no_grad():
            enc = self.encode(inputs, subset, N=K, **kwargs)
            z = enc.z if K > 1 else enc.z.unsqueeze(0)
            B = z.shape[1]
            xs = {m: inputs.data[m].float().contiguous() for m in pred_mods}
            ll = {m: torch.empty(B, dtype=torch.float32, device=z.device) for m in pred_mods}
            step = max(1, kernels.IWAE_ROWS_BUDGET // K)
            for b0 in range(0, B, step):
                b1 = min(B, b0 + step)
                for m in pred_mods:
                    zc = z[:, b0:b1]
                    if not enc.one_latent_space:
                        w = enc.modalities_z[m] if K > 1 else enc.modalities_z[m].unsqueeze(0)
                        zc = torch.cat([zc, w[:, b0:b1]], dim=-1)
                    rec = self.decoders[m](zc.contiguous()).reconstruction
                    rows = kernels.recon_nll_rows([rec], [xs[m][b0:b1]], [self.recon_dists[m][0]],
                                                  [self.recon_dists[m][1]], K, b1 - b0)
                    kernels.iwae_reduce([kernels.axpby(rows[0], -1.0, None, 0.0)], out=ll[m][b0:b1])
            return {m: -ll[m].sum() / B for m in pred_mods}

    _NLL_INCOMPLETE = "The compute_joint_nll method is not yet implemented for incomplete datasets."

    def _joint_nll(self, inputs, z, locs, sds, family=0, prior_loc=None, prior_sd=None, private=None):
        """Importance-sampled -ln p(x) summed over the batch: the shared body of the reference's `compute_joint_nll`
        methods (mopoe_model.py:522-592, mmvae_model.py:399-441, mvtcae_model.py:249-289, joint_model.py:111-152).
        z [K,B,L] importance samples; q = uniform mixture of the experts (locs[e], sds[e]) [B,L].  The K axis is a
        kernel axis here (decoders on [K,b,L], one mvk_recon_nll_fwd, mvk_iwae_logw, mvk_iwae_reduce per chunk of
        data points) instead of a Python loop per data point and per `batch_size_K` samples; the likelihoods are
        NOT rescaled, as in the reference.
        private = {m: (w [K,B,S_m], loc [B,S_m], sd [B,S_m])}: modality-specific latents (mopoe_model.py:507-521,
        :560-567): decoder m sees [z, w_m], and ln N(w_m; 0, I) - ln q(w_m | x_m) joins the log-weight."""
        names = list(inputs.data.keys())
        xs = [inputs.data[m].float().contiguous() for m in names]
        dists = [self.recon_dists[m][0] for m in names]
        scales = [self.recon_dists[m][1] for m in names]
        device = z.device
        order = self._branch_order(inputs, names)

        def decode_rows(zc, b0, b1):
            K, b = zc.shape[0], zc.shape[1]
            if private is None:
                rec = kernels.run_branches(order, lambda m: self.decoders[m](zc).reconstruction, device)
                return kernels.recon_nll_rows([rec[m] for m in names], [x[b0:b1] for x in xs], dists, scales, K, b)
            ws = {m: private[m][0][:, b0:b1].contiguous() for m in names}
            rec = kernels.run_branches(order, lambda m: self.decoders[m](torch.cat([zc, ws[m]], dim=-1)).reconstruction,
                                       device)
            rows = kernels.recon_nll_rows([rec[m] for m in names], [x[b0:b1] for x in xs], dists, scales, K, b)
            for m in names:  # -(ln p(w_m) - ln q(w_m | x_m)) as one more "row" of the log-weight
                ratio = kernels.iwae_logw(ws[m], [], [private[m][1][b0:b1]], [private[m][2][b0:b1]])
                rows.append(kernels.axpby(ratio, -1.0, None, 0.0))
            return rows

        with torch.no_grad():
            return kernels.joint_nll(decode_rows, z, locs, sds, family, prior_loc, prior_sd)

    def generate_from_prior(self, n_samples, **kwargs):
        shape = [n_samples, self.latent_dim] if n_samples > 1 else [self.latent_dim]
        dev = next(self.parameters()).device
        return ModelOutput(z=torch.randn(shape, device=dev), one_latent_space=True)

    # -- shared helpers for the HIP forward passes ------------------------------------------------------------
    def _gaussian_encoding(self, mu, log_var, N, return_mean, flatten, noise=None):
        """`rsample_from_gaussian(mu, log_var, N, return_mean, flatten)` (base_utils.py:150-172) for the encode()
        helpers, on mvk_iwae_sample; noise [N,B,L] (or [B,L] for N == 1) replaces the N(0,1) draw."""
        if return_mean:
            z = torch.stack([mu] * N) if N > 1 else mu
        else:
            B, L = mu.shape
            if noise is not None and noise.dim() == 2:
                noise = noise.unsqueeze(0)
            z = kernels.iwae_sample(mu, kernels.std_from_logvar(log_var), self._noise((N, B, L), mu.device, noise))
            if N == 1:
                z = z[0]
        if N > 1 and flatten:
            z = z.reshape(-1, *z.shape[2:])
        return z

    @staticmethod
    def _noise(shape, device, noise=None, uniform=False):
        """Noise is an explicit kernel input; by default it is drawn from torch's generator."""
        if noise is not None:
            if tuple(noise.shape) != tuple(shape):
                raise ValueError(f"noise has shape {tuple(noise.shape)}, expected {tuple(shape)}")
            return noise.to(device=device, dtype=torch.float32).contiguous()
        device = torch.device(device)
        lo = torch.finfo(torch.float32).eps - 1.0
        if device.type == "cuda" and kernels.DEVICE_RNG:  # generator state in device memory: graph replays need no host launch
            return kernels.device_randn(tuple(shape), device, uniform=uniform, lo=lo, hi=1.0)
        if uniform:
            return torch.empty(shape, device=device, dtype=torch.float32).uniform_(lo, 1.0)
        return torch.randn(shape, device=device, dtype=torch.float32)

    def _recon_spec(self, names, data, masks, K, B):
        xs, mks, dist, scale, resc = [], [], [], [], []
        for m in names:
            x = data[m]
            if x.dtype != torch.float32:
                x = x.float()
            xs.append(x.contiguous())
            mk = None
            if masks is not None:
                mk = masks[m].to(torch.bool).contiguous()
            mks.append(mk)
            code, sc = self.recon_dists[m]
            dist.append(code)
            scale.append(sc)
            resc.append(float(self.rescale_factors[m]))
        return dict(K=K, B=B, x=xs, masks=mks, dist=dist, scale=scale, rescale=resc)
