"""Default MLP architectures with the reference's parameter names and shapes
(`multivae/models/nn/default_architectures.py:21-72, 143-258`), computed by the fp32-MFMA GEMM kernels.

The `nn.Linear` / `nn.Sequential` objects are parameter containers only (they give the state_dict the keys
`layers.0.0.weight`, `embedding.weight`, ... and the default initialisation); `forward` never calls them:
the whole network is one autograd node (`kernels.MLPEncoderFn` / `kernels.MLPDecoderFn`).
"""
from typing import List

import numpy as np
import torch
from torch import nn

from ... import kernels
from ..base.base_config import BaseAEConfig
from ..base.base_utils import ModelOutput
from .base_architectures import BaseDecoder, BaseEncoder


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
