"""Default MLP architectures with the reference's parameter names and shapes
(`multivae/models/nn/default_architectures.py:21-72, 143-258`), computed by the fp32-MFMA GEMM kernels.

The `nn.Linear` / `nn.Sequential` objects are parameter containers only (they give the state_dict the keys
`layers.0.0.weight`, `embedding.weight`, ... and the default initialisation); `forward` never calls them:
the whole network is one autograd node (`kernels.MLPEncoderFn` / `kernels.MLPDecoderFn`).
"""
import math
from typing import List

import numpy as np
import torch
from torch import nn

from ... import kernels
from ..base.base_config import BaseAEConfig
from ..base.base_utils import ModelOutput
from .base_architectures import BaseDecoder, BaseEncoder, BaseJointEncoder, BaseMultilatentEncoder


class Encoder_VAE_MLP(BaseEncoder):
    def __init__(self, args, n_hidden=1):
        BaseEncoder.__init__(self)
        self.input_dim = args.input_dim
        self.latent_dim = args.latent_dim
        layers = nn.ModuleList()
        layers.append(nn.Sequential(nn.Linear(int(np.prod(args.input_dim)), 512), nn.ReLU()))
        for _ in range(n_hidden):
            layers.append(nn.Sequential(nn.Linear(512, 512), nn.ReLU()))
        self.layers = layers
        self.depth = len(layers)
        self.embedding = nn.Linear(512, self.latent_dim)
        self.log_var = nn.Linear(512, self.latent_dim)

    def forward(self, x, output_layer_levels: List[int] = None):
        if output_layer_levels is not None:
            raise NotImplementedError("output_layer_levels is not supported on the fused HIP path")
        params = []
        for seq in self.layers:
            params += [seq[0].weight, seq[0].bias]
        params += [self.embedding.weight, self.embedding.bias, self.log_var.weight, self.log_var.bias]
        mu, lv = kernels.MLPEncoderFn.apply(x, *params)
        return ModelOutput(embedding=mu, log_covariance=lv)


class Decoder_AE_MLP(BaseDecoder):
    """Accepts any input shape (*, latent_dim); output is (*, *input_dim)."""
    rows_independent = True


    def __init__(self, args):
        BaseDecoder.__init__(self)
        self.input_dim = tuple(args.input_dim)
        layers = nn.ModuleList()
        layers.append(nn.Sequential(nn.Linear(args.latent_dim, 512), nn.ReLU()))
        layers.append(nn.Sequential(nn.Linear(512, int(np.prod(args.input_dim))), nn.Sigmoid()))
        self.layers = layers
        self.depth = len(layers)

    def forward(self, z: torch.Tensor, **kwargs):
        l0, l1 = self.layers[0][0], self.layers[1][0]
        out = kernels.MLPDecoderFn.apply(z, l0.weight, l0.bias, l1.weight, l1.bias, self.input_dim)
        return ModelOutput(reconstruction=out)

    def late_leaf_params(self):
        """The parameters whose gradients are leaves of the backward pass that a rotated step (kernels.Rotation) produces at the
        head of the NEXT step: trainers.FlatParams keeps them together at the end of its buffers."""
        l0, l1 = self.layers[0][0], self.layers[1][0]
        return [l0.weight, l1.weight, l1.bias] if self.depth == 2 else []

    def early_work(self, x: torch.Tensor, rows: int):
        """What `reconstruction_nll` needs that does not depend on z — the fp16 pair planes of the output layer's weight and
        the bound of the targets — launched where the caller has an idle stream (MoPoE: the head of the short encoder's
        branch) instead of in the decoder's own chain.  A no-op when the fused tail would not take `rows` decoder rows."""
        l0, l1 = self.layers[0][0], self.layers[1][0]
        D = int(np.prod(self.input_dim))
        if (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x[0].numel() == D and torch.is_grad_enabled()
                and kernels.mlp_fused_tail_ok(rows, l0.weight.shape[1], l0.weight.shape[0], D)):
            kernels.dense16_pack(l1.weight)
            kernels.dense16_xamax(x.reshape(x.shape[0], D))

    def reconstruction_nll(self, z: torch.Tensor, x: torch.Tensor, dist: str = "normal", scale: float = 1.0, row_weight=None):
        """OPT-IN fast path for models that own the loss (MoPoE here; the contract of Decoder_VAE_SVHN.reconstruction_nll):
        -log p(x | decoder(z)) under Normal(scale), computed in the epilogue of the output layer's GEMM on fp16 pair planes
        (csrc/dense16.hip) so that neither the reconstruction nor its gradient travels through HBM as fp32.  Returns PARTIAL
        row sums [P, *z.shape[:-1]] whose sum over P is the NLL of the row — a caller that only sums rows uses them as they
        are —, or None when this decoder / likelihood / batch has no fused form (the caller then uses `forward` + the generic
        likelihood kernel).  x: [B, *input_dim]; the rows of z are scored against x[row % B].  row_weight: the weight the rows
        will enter the loss with (the expected d loss / d rows, e.g. rescale / (K B)): the stored gradient is pre-multiplied by
        it, and a backward pass that receives exactly that constant rescales nothing."""
        l0, l1 = self.layers[0][0], self.layers[1][0]
        n = z.reshape(-1, z.shape[-1]).shape[0]
        D = int(np.prod(self.input_dim))
        if (dist != "normal" or not torch.is_grad_enabled() or z.shape[-1] != l0.weight.shape[1] or not x.is_cuda
                or x.device != z.device or x.shape[0] == 0 or n % x.shape[0] != 0 or x[0].numel() != D
                or not kernels.mlp_fused_tail_ok(n, l0.weight.shape[1], l0.weight.shape[0], D)):
            return None
        if row_weight is not None and not (math.isfinite(float(row_weight)) and float(row_weight) != 0.0):
            return None  # a zero / non-finite weight has no pre-multiplied gradient form (its bound would be 0): generic path
        x2 = x.float().reshape(x.shape[0], D).contiguous()
        if x2.data_ptr() % 16:  # a view into a larger storage: the kernel reads the targets with 16-byte loads
            x2 = x2.clone()
        return kernels.MLPDecoderFn.apply(z, l0.weight, l0.bias, l1.weight, l1.bias, self.input_dim, x2, float(scale),
                                          1.0 if row_weight is None else float(row_weight))


class Encoder_VAE_MLP_Style(BaseMultilatentEncoder):
    """`default_architectures.py:75-141`: Linear(prod D, 512)+ReLU and four linear heads, one autograd node."""

    def __init__(self, args):
        BaseMultilatentEncoder.__init__(self)
        self.input_dim = args.input_dim
        self.latent_dim = args.latent_dim
        self.style_dim = args.style_dim
        layers = nn.ModuleList()
        layers.append(nn.Sequential(nn.Linear(int(np.prod(args.input_dim)), 512), nn.ReLU()))
        self.layers = layers
        self.depth = len(layers)
        self.embedding = nn.Linear(512, self.latent_dim)
        self.log_var = nn.Linear(512, self.latent_dim)
        self.style_embedding = nn.Linear(512, self.style_dim)
        self.style_log_var = nn.Linear(512, self.style_dim)

    def forward(self, x, output_layer_levels: List[int] = None):
        if output_layer_levels is not None:
            raise NotImplementedError("output_layer_levels is not supported on the fused HIP path")
        params = []
        for seq in self.layers:
            params += [seq[0].weight, seq[0].bias]
        for head in (self.embedding, self.log_var, self.style_embedding, self.style_log_var):
            params += [head.weight, head.bias]
        mu, lv, smu, slv = kernels.MLPHeadsFn.apply(x, 4, *params)
        return ModelOutput(embedding=mu, log_covariance=lv, style_embedding=smu, style_log_covariance=slv)


def BaseDictEncoders_MultiLatents(input_dims: dict, latent_dim: int, modality_dims: dict):
    encoders = nn.ModuleDict()
    for mod in input_dims:
        encoders[mod] = Encoder_VAE_MLP_Style(BaseAEConfig(input_dim=input_dims[mod], latent_dim=latent_dim,
                                                           style_dim=modality_dims[mod]))
    return encoders


def BaseDictDecodersMultiLatents(input_dims: dict, latent_dim: int, modality_dims: dict):
    decoders = nn.ModuleDict()
    for mod in input_dims:
        decoders[mod] = Decoder_AE_MLP(BaseAEConfig(input_dim=input_dims[mod],
                                                    latent_dim=latent_dim + modality_dims[mod]))
    return decoders


class MultipleHeadJointEncoder(BaseJointEncoder):
    """`default_architectures.py:261-322`: deep copies of the unimodal encoders, their embeddings concatenated in
    modality order, a unifying MLP ([Linear+ReLU] x n_hidden_layers) and two linear heads.  The MLP and the heads run
    as ONE autograd node on the GEMM engine (kernels.MLPEncoderFn); the concatenation is a device copy."""

    def __init__(self, dict_encoders: dict, args, hidden_dim=512, n_hidden_layers=2, **kwargs):
        BaseJointEncoder.__init__(self)
        from copy import deepcopy

        self.encoders = nn.ModuleDict()
        self.joint_input_dim = 0
        for modality in dict_encoders:
            self.encoders[modality] = deepcopy(dict_encoders[modality])
            self.joint_input_dim += self.encoders[modality].latent_dim
        modules = [nn.Sequential(nn.Linear(self.joint_input_dim, hidden_dim), nn.ReLU(True))]
        for _ in range(n_hidden_layers - 1):
            modules.append(nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.ReLU(True)))
        self.enc = nn.Sequential(*modules)
        self.fc1 = nn.Linear(hidden_dim, args.latent_dim)
        self.fc2 = nn.Linear(hidden_dim, args.latent_dim)
        self.latent_dim = args.latent_dim

    def forward(self, x: dict):
        assert list(x.keys()) == list(self.encoders.keys())
        names = list(self.encoders.keys())
        dev = x[names[0]].device
        outs = kernels.run_branches(names, lambda m: self.encoders[m](x[m])["embedding"], dev)
        h = torch.cat([outs[m] for m in names], dim=1)
        params = []
        for seq in self.enc:
            params += [seq[0].weight, seq[0].bias]
        params += [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias]
        mu, lv = kernels.MLPEncoderFn.apply(h, *params)
        return ModelOutput(embedding=mu, log_covariance=lv)


def BaseDictEncoders(input_dims: dict, latent_dim: int):
    encoders = nn.ModuleDict()
    for mod in input_dims:
        encoders[mod] = Encoder_VAE_MLP(BaseAEConfig(input_dim=tuple(input_dims[mod]), latent_dim=latent_dim))
    return encoders


def BaseDictDecoders(input_dims: dict, latent_dim: int):
    decoders = nn.ModuleDict()
    for mod in input_dims:
        decoders[mod] = Decoder_AE_MLP(BaseAEConfig(input_dim=tuple(input_dims[mod]), latent_dim=latent_dim))
    return decoders
