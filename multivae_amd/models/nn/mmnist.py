"""PolyMNIST ResNet architectures of MMVAE+ (`multivae/models/nn/mmnist.py:214-366`, adapted there from
github.com/epalu/mmvaeplus) on the HIP kernels: 3x3 convolutions on the implicit-GEMM engine, AvgPool / Upsample /
residual kernels, the whole convolutional stack of a network as ONE autograd node (kernels.ResnetStackFn).  Module
structure and parameter names follow the reference, so its state_dicts load unchanged."""
import numpy as np
import torch
from torch import nn

from ... import kernels
from ..base.base_utils import ModelOutput
from .base_architectures import BaseDecoder, BaseEncoder


class ResnetBlock(nn.Module):
    """x_s + 0.1 * lrelu(conv2(lrelu(conv1(x))))  (`mmnist.py:214-252`).  Parameter container: the arithmetic runs
    inside kernels.ResnetStackFn."""

    order = "post"

    def __init__(self, nb_channels_in, nb_channels_out, nb_channels_hidden=None, bias=True):
        super().__init__()
        self.learn_shortcut = nb_channels_in != nb_channels_out
        if nb_channels_hidden is None:
            nb_channels_hidden = min(nb_channels_in, nb_channels_out)
        self.conv_layers = nn.Sequential(
            nn.Conv2d(nb_channels_in, nb_channels_hidden, 3, stride=1, padding=1),
            nn.LeakyReLU(2e-1),
            nn.Conv2d(nb_channels_hidden, nb_channels_out, 3, stride=1, padding=1, bias=bias),
            nn.LeakyReLU(2e-1),
        )
        if self.learn_shortcut:
            self.shortcut_layer = nn.Conv2d(nb_channels_in, nb_channels_out, 1, stride=1, padding=0, bias=False)

    def forward(self, x):  # NCHW in / out, like the reference module
        prog, params = [], []
        _add_block(prog, params, self)
        y = kernels.ResnetStackFn.apply(x.permute(0, 2, 3, 1), prog, *params)
        return y.permute(0, 3, 1, 2)


def _add_conv(prog, params, conv, act):
    params.append(conv.weight)
    iw = len(params) - 1
    ib = None
    if conv.bias is not None:
        params.append(conv.bias)
        ib = len(params) - 1
    prog.append(("conv", iw, ib, act))


def _add_block(prog, params, blk):
    if hasattr(blk, "conv_layers"):  # mmnist.py naming
        c1, c2, sc = blk.conv_layers[0], blk.conv_layers[2], getattr(blk, "shortcut_layer", None)
    else:  # cub.py naming
        c1, c2, sc = blk.conv_0, blk.conv_1, getattr(blk, "conv_s", None)
    idx = []
    for t in (c1.weight, c1.bias, c2.weight, c2.bias):
        if t is None:
            idx.append(None)
        else:
            params.append(t)
            idx.append(len(params) - 1)
    isc = None
    if sc is not None:
        params.append(sc.weight)
        isc = len(params) - 1
    prog.append(("block", blk.order, idx[0], idx[1], idx[2], idx[3], isc))


def _add_sequential(prog, params, seq):
    for m in seq:
        if hasattr(m, "order") and (hasattr(m, "conv_layers") or hasattr(m, "conv_0")):
            _add_block(prog, params, m)
        elif isinstance(m, nn.AvgPool2d):
            prog.append(("pool",))
        elif isinstance(m, nn.Upsample):
            prog.append(("up",))
        else:
            raise TypeError(f"unsupported layer in a ResNet stack: {type(m).__name__}")


class EncoderResnetMMNIST(BaseEncoder):
    """`mmnist.py:255-321`: conv_img -> ResnetBlock(64,64) -> [AvgPool, ResnetBlock] x 2 -> flatten (NCHW order) ->
    fc_mu / fc_lv, once for the shared latent (u) and once for the private latent (w, if private_latent_dim > 0)."""

    def __init__(self, private_latent_dim, shared_latent_dim):
        super().__init__()
        self.latent_dim = shared_latent_dim
        self.style_dim = private_latent_dim
        s0 = self.s0 = 7
        nf = self.nf = 64
        nf_max = self.nf_max = 1024
        size = 28
        self.multiple_latent = private_latent_dim > 0
        nlayers = int(np.log2(size / s0))
        self.nf0 = min(nf_max, nf * 2 ** nlayers)
        blocks_w = [ResnetBlock(nf, nf)]
        blocks_u = [ResnetBlock(nf, nf)]
        for i in range(nlayers):
            nf0 = min(nf * 2 ** i, nf_max)
            nf1 = min(nf * 2 ** (i + 1), nf_max)
            blocks_w += [nn.AvgPool2d(3, stride=2, padding=1), ResnetBlock(nf0, nf1)]
            blocks_u += [nn.AvgPool2d(3, stride=2, padding=1), ResnetBlock(nf0, nf1)]
        if self.multiple_latent:
            self.conv_img_w = nn.Conv2d(3, 1 * nf, 3, padding=1)
            self.resnet_w = nn.Sequential(*blocks_w)
            self.fc_mu_w = nn.Linear(self.nf0 * s0 * s0, private_latent_dim)
            self.fc_lv_w = nn.Linear(self.nf0 * s0 * s0, private_latent_dim)
        self.conv_img_u = nn.Conv2d(3, 1 * nf, 3, padding=1)
        self.resnet_u = nn.Sequential(*blocks_u)
        self.fc_mu_u = nn.Linear(self.nf0 * s0 * s0, shared_latent_dim)
        self.fc_lv_u = nn.Linear(self.nf0 * s0 * s0, shared_latent_dim)

    def _branch(self, x_nhwc, conv_img, resnet, fc_mu, fc_lv):
        prog, params = [], []
        _add_conv(prog, params, conv_img, kernels.NONE)
        _add_sequential(prog, params, resnet)
        h = kernels.ResnetStackFn.apply(x_nhwc, prog, *params)  # [B, 7, 7, nf0]
        flat = kernels.nhwc_to_flat_nchw(h)  # the reference flattens NCHW
        return kernels.MLPHeadsFn.apply(flat, 2, fc_mu.weight, fc_mu.bias, fc_lv.weight, fc_lv.bias)

    def forward(self, x):
        x_nhwc = x.permute(0, 2, 3, 1).contiguous()
        mu, lv = self._branch(x_nhwc, self.conv_img_u, self.resnet_u, self.fc_mu_u, self.fc_lv_u)
        output = ModelOutput(embedding=mu, log_covariance=lv)
        if self.multiple_latent:
            smu, slv = self._branch(x_nhwc, self.conv_img_w, self.resnet_w, self.fc_mu_w, self.fc_lv_w)
            output["style_embedding"] = smu
            output["style_log_covariance"] = slv
        return output


class DecoderResnetMMNIST(BaseDecoder):
    """`mmnist.py:324-366`: fc -> [ResnetBlock, Upsample] x 2 -> ResnetBlock(64,64) -> Conv(64,3,3) + LeakyReLU(0.2).
    latent_dim is the total (shared + private) latent dimension."""
    rows_independent = True


    def __init__(self, latent_dim):
        super().__init__()
        s0 = self.s0 = 7
        nf = self.nf = 64
        nf_max = self.nf_max = 512
        size = 28
        nlayers = int(np.log2(size / s0))
        self.nf0 = min(nf_max, nf * 2 ** nlayers)
        self.fc = nn.Linear(latent_dim, self.nf0 * s0 * s0)
        blocks = []
        for i in range(nlayers):
            nf0 = min(nf * 2 ** (nlayers - i), nf_max)
            nf1 = min(nf * 2 ** (nlayers - i - 1), nf_max)
            blocks += [ResnetBlock(nf0, nf1), nn.Upsample(scale_factor=2)]
        blocks += [ResnetBlock(nf, nf)]
        self.resnet = nn.Sequential(*blocks)
        self.conv_img = nn.Sequential(nn.Conv2d(nf, 3, 3, padding=1), nn.LeakyReLU(2e-1))

    def forward(self, z):
        z2 = z.reshape(-1, z.shape[-1])
        (h,) = kernels.MLPHeadsFn.apply(z2, 1, self.fc.weight, self.fc.bias)  # [N, nf0*7*7] in NCHW order
        h = kernels.flat_nchw_to_nhwc(h, self.nf0, self.s0, self.s0)
        prog, params = [], []
        _add_sequential(prog, params, self.resnet)
        _add_conv(prog, params, self.conv_img[0], kernels.LEAKY)
        out = kernels.ResnetStackFn.apply(h, prog, *params).permute(0, 3, 1, 2)  # NCHW [N, 3, 28, 28]
        lead = z.shape[:1] if z.dim() == 2 else z.shape[:2]
        return ModelOutput(reconstruction=out.reshape(*lead, *out.shape[1:]))
