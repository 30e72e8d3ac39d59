"""CUB / 64x64-image ResNet encoder and decoder (`multivae/models/nn/cub.py:144-293`, adapted there from
github.com/epalu/mmvaeplus) on the HIP kernels.  Same building blocks as models/nn/mmnist.py; the ResnetBlock here
is the pre-activation variant  x_s + 0.1 * conv_1(lrelu(conv_0(lrelu(x))))  (`cub.py:274-280`).  Module structure and
parameter names follow the reference.  (The reference's CUB text networks are outside the hot-path scope.)"""
import numpy as np
from torch import nn

from ... import kernels
from ..base.base_utils import ModelOutput
from .base_architectures import BaseDecoder, BaseEncoder
from .mmnist import _add_block, _add_conv, _add_sequential


class ResnetBlock(nn.Module):
    order = "pre"

    def __init__(self, fin, fout, fhidden=None, is_bias=True):
        super().__init__()
        self.is_bias = is_bias
        self.learned_shortcut = fin != fout
        self.fin, self.fout = fin, fout
        self.fhidden = min(fin, fout) if fhidden is None else fhidden
        self.conv_0 = nn.Conv2d(self.fin, self.fhidden, 3, stride=1, padding=1)
        self.conv_1 = nn.Conv2d(self.fhidden, self.fout, 3, stride=1, padding=1, bias=is_bias)
        if self.learned_shortcut:
            self.conv_s = nn.Conv2d(self.fin, self.fout, 1, stride=1, padding=0, bias=False)

    def forward(self, x):  # NCHW in / out
        prog, params = [], []
        _add_block(prog, params, self)
        return kernels.ResnetStackFn.apply(x.permute(0, 2, 3, 1), prog, *params).permute(0, 3, 1, 2)


class CUB_Resnet_Encoder(BaseEncoder):
    """`cub.py:144-196`: conv_img -> ResnetBlock -> [AvgPool, ResnetBlock] x log2(64/s0) -> flatten -> fc(lrelu(.))."""

    def __init__(self, latent_dim, s0=16, nfilter=64, nfilter_max=1024):
        super().__init__()
        self.latent_dim = latent_dim
        size = 64
        self.s0 = s0
        nf = self.nf = nfilter
        nf_max = self.nf_max = nfilter_max
        nlayers = int(np.log2(size / s0))
        self.nf0 = min(nf_max, nf * 2 ** nlayers)
        blocks = [ResnetBlock(nf, nf)]
        for i in range(nlayers):
            nf0 = min(nf * 2 ** i, nf_max)
            nf1 = min(nf * 2 ** (i + 1), nf_max)
            blocks += [nn.AvgPool2d(3, stride=2, padding=1), ResnetBlock(nf0, nf1)]
        self.conv_img = nn.Conv2d(3, 1 * nf, 3, padding=1)
        self.resnet = nn.Sequential(*blocks)
        self.fc_mu = nn.Linear(self.nf0 * s0 * s0, self.latent_dim)
        self.fc_logvar = nn.Linear(self.nf0 * s0 * s0, self.latent_dim)

    def forward(self, x):
        prog, params = [], []
        _add_conv(prog, params, self.conv_img, kernels.NONE)
        _add_sequential(prog, params, self.resnet)
        h = kernels.ResnetStackFn.apply(x.permute(0, 2, 3, 1).contiguous(), prog, *params)
        # fc_mu(actvn(out)), fc_logvar(actvn(out)): the activation and the NCHW flatten in one pass
        flat = kernels.nhwc_to_flat_nchw(h, kernels.LEAKY)
        mu, lv = kernels.MLPHeadsFn.apply(flat, 2, self.fc_mu.weight, self.fc_mu.bias, self.fc_logvar.weight,
                                          self.fc_logvar.bias)
        return ModelOutput(embedding=mu, log_covariance=lv)


class CUB_Resnet_Decoder(BaseDecoder):
    """`cub.py:199-247`: fc -> [ResnetBlock, Upsample] x log2(64/s0) -> ResnetBlock -> conv_img(lrelu(.)) (logits)."""
    rows_independent = True


    def __init__(self, latent_dim, s0=16, nfilter=64, nfilter_max=512, **kwargs):
        super().__init__()
        size = 64
        self.latent_dim = latent_dim
        self.s0 = s0
        nf = self.nf = nfilter
        nf_max = self.nf_max = nfilter_max
        nlayers = int(np.log2(size / s0))
        self.nf0 = min(nf_max, nf * 2 ** nlayers)
        self.fc = nn.Linear(self.latent_dim, self.nf0 * s0 * s0)
        blocks = []
        for i in range(nlayers):
            nf0 = min(nf * 2 ** (nlayers - i), nf_max)
            nf1 = min(nf * 2 ** (nlayers - i - 1), nf_max)
            blocks += [ResnetBlock(nf0, nf1), nn.Upsample(scale_factor=2)]
        blocks += [ResnetBlock(nf, nf)]
        self.resnet = nn.Sequential(*blocks)
        self.conv_img = nn.Conv2d(nf, 3, 3, padding=1)
        self.sigmoid = nn.Sigmoid()

    def forward(self, z):
        # the reference's `view(z.size(0), ...)` only supports 2-D latents; leading dims are flattened here so that the
        # K-sample models can decode [K, B, L] as with every other in-package decoder
        z2 = z.reshape(-1, z.shape[-1])
        (h,) = kernels.MLPHeadsFn.apply(z2, 1, self.fc.weight, self.fc.bias)
        h = kernels.flat_nchw_to_nhwc(h, self.nf0, self.s0, self.s0)
        prog, params = [], []
        _add_sequential(prog, params, self.resnet)
        prog.append(("act", kernels.LEAKY))
        _add_conv(prog, params, self.conv_img, kernels.NONE)
        out = kernels.ResnetStackFn.apply(h, prog, *params).permute(0, 3, 1, 2)
        return ModelOutput(reconstruction=out.reshape(*z.shape[:-1], *out.shape[1:]))
