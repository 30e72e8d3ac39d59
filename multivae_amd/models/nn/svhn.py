"""SVHN convolutional encoder / decoder with the reference's parameter names and shapes
(`multivae/models/nn/svhn.py:7-70`), computed by the implicit-GEMM fp32-MFMA kernels on NHWC activations.
The `nn.Conv2d` / `nn.ConvTranspose2d` objects only hold the parameters (state_dict keys `enc.0.weight`,
`c1.weight`, `dec.0.weight`, ...)."""
import math

import torch
from torch import nn

from ... import kernels
from ..base.base_utils import ModelOutput
from .base_architectures import BaseDecoder, BaseEncoder


class Encoder_VAE_SVHN(BaseEncoder):
    def __init__(self, args):
        BaseEncoder.__init__(self)
        self.input_dim = args.input_dim
        self.latent_dim = args.latent_dim
        self.n_channels = args.input_dim[0]
        self.fBase = 32
        self.enc = nn.Sequential(
            nn.Conv2d(self.n_channels, self.fBase, 4, 2, 1, bias=True), nn.ReLU(True),
            nn.Conv2d(self.fBase, self.fBase * 2, 4, 2, 1, bias=True), nn.ReLU(True),
            nn.Conv2d(self.fBase * 2, self.fBase * 4, 4, 2, 1, bias=True), nn.ReLU(True),
        )
        self.c1 = nn.Conv2d(self.fBase * 4, self.latent_dim, 4, 2, 0)
        self.c2 = nn.Conv2d(self.fBase * 4, self.latent_dim, 4, 2, 0)

    def pack_jobs(self):
        """The weight packs SVHNEncoderFn asks for (kernels.pack_scope: one pack launch per forward pass of the model)."""
        e = self.enc
        # (the image-consuming first layer reads the reference layout where its kernel covers the shape: no pack)
        first = [] if (e[0].weight.shape[1] <= 4 and e[0].weight.shape[0] in (16, 32, 64)) else [(e[0].weight, True, False)]
        return first + [(e[2].weight, True, True), (e[4].weight, True, True),
                        (self.c1.weight, True, False), (self.c2.weight, True, False)]

    def forward(self, x: torch.Tensor):
        e = self.enc
        mu, lv = kernels.SVHNEncoderFn.apply(x, e[0].weight, e[0].bias, e[2].weight, e[2].bias, e[4].weight,
                                             e[4].bias, self.c1.weight, self.c1.bias, self.c2.weight, self.c2.bias)
        # the reference squeezes ALL unit dims of the [B,L,1,1] head outputs (svhn.py:35-36; Appendix D)
        return ModelOutput(embedding=mu.squeeze(), log_covariance=lv.squeeze())


class Decoder_VAE_SVHN(BaseDecoder):
    rows_independent = True

    def __init__(self, args):
        BaseDecoder.__init__(self)
        self.latent_dim = args.latent_dim
        self.fBase = 32
        self.nb_channels = args.input_dim[0]
        self.dec = nn.Sequential(
            nn.ConvTranspose2d(self.latent_dim, self.fBase * 4, 4, 1, 0, bias=True), nn.ReLU(True),
            nn.ConvTranspose2d(self.fBase * 4, self.fBase * 2, 4, 2, 1, bias=True), nn.ReLU(True),
            nn.ConvTranspose2d(self.fBase * 2, self.fBase, 4, 2, 1, bias=True), nn.ReLU(True),
            nn.ConvTranspose2d(self.fBase, self.nb_channels, 4, 2, 1, bias=True), nn.Sigmoid(),
        )

    def pack_jobs(self):
        """The weight packs SVHNDecoderFn asks for (kernels.pack_scope)."""
        d = self.dec
        return [(d[0].weight, "unflatten"), (d[2].weight, True, True), (d[4].weight, True, True), (d[6].weight, True, False)]

    def late_leaf_params(self):
        """The weights whose gradients are leaves of the backward pass that a rotated step (kernels.Rotation) produces at the
        head of the NEXT step: trainers.FlatParams keeps them together at the end of its buffers."""
        d = self.dec
        return [d[0].weight, d[2].weight, d[4].weight]

    def forward(self, z: torch.Tensor):
        d = self.dec
        out = kernels.SVHNDecoderFn.apply(z, d[0].weight, d[0].bias, d[2].weight, d[2].bias, d[4].weight, d[4].bias,
                                          d[6].weight, d[6].bias)
        return ModelOutput(reconstruction=out)

    def reconstruction_nll(self, z: torch.Tensor, x: torch.Tensor, dist: str = "normal", scale: float = 1.0, row_weight=None):
        """OPT-IN fast path for models that own the loss (MoPoE / MVTCAE here): -log p(x | decoder(z)) summed over the image,
        one value per latent row ([*z.shape[:-1]]), for a Normal(scale) likelihood — computed in the epilogue of the last
        ConvTranspose2d + Sigmoid, so neither the reconstruction nor its gradient travels through HBM (DESIGN.md section 4:
        `small_up_fwd_bf_kernel` fused tail).  Returns None when this decoder / likelihood has no fused form: the caller then
        uses `forward` + the generic likelihood kernel, as for every user-written decoder.  x: [B, C, 32, 32]; the rows of z are
        scored against x[row % B] (K samples per data point).  row_weight: the weight the rows will enter the loss with (the
        expected d loss / d rows): the stored gradient is pre-multiplied by it, and a backward pass that receives exactly that
        constant reads no row gradient (Decoder_AE_MLP.reconstruction_nll has the same contract)."""
        d = self.dec
        if dist != "normal" or x.dim() != 4 or z.shape[-1] != self.latent_dim or not torch.is_grad_enabled() \
                or not kernels.svhn_fused_tail_ok(d[6].weight.shape[1], d[6].weight.shape[0]):
            return None
        n = z.reshape(-1, z.shape[-1]).shape[0]
        # the kernel reads x as [x.shape[0], C, 32, 32] fp32 on z's device: anything else takes the generic path
        if (tuple(x.shape[1:]) != (d[6].weight.shape[1], 32, 32) or not x.is_cuda or x.device != z.device
                or x.shape[0] == 0 or n % x.shape[0] != 0):
            return None
        if row_weight is not None and not (math.isfinite(float(row_weight)) and float(row_weight) != 0.0):
            return None  # no pre-multiplied gradient form for a zero / non-finite weight: generic path
        x = x.float().contiguous()
        if x.data_ptr() % 16:
            x = x.clone()
        return kernels.SVHNDecoderFn.apply(z, d[0].weight, d[0].bias, d[2].weight, d[2].bias, d[4].weight, d[4].bias,
                                           d[6].weight, d[6].bias, x, float(scale),
                                           1.0 if row_weight is None else float(row_weight))
