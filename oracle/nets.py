"""Oracle: the in-package encoder/decoder architectures as pure functions of a state dict.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates
  /root/reference/src/multivae/models/nn/default_architectures.py:21-72   Encoder_VAE_MLP
  /root/reference/src/multivae/models/nn/default_architectures.py:225-258 Decoder_AE_MLP
  /root/reference/src/multivae/models/nn/svhn.py:7-38                     Encoder_VAE_SVHN
  /root/reference/src/multivae/models/nn/svhn.py:41-70                    Decoder_VAE_SVHN
  /root/reference/src/multivae/models/nn/default_architectures.py:261-322 MultipleHeadJointEncoder
  /root/reference/src/multivae/models/nn/mmnist.py:214-366                ResnetBlock, Encoder/DecoderResnetMMNIST
  /root/reference/src/multivae/models/nn/cub.py:144-293                   CUB_Resnet_Encoder / Decoder, ResnetBlock
with the reference's parameter names (`<prefix>layers.0.0.weight`, `<prefix>enc.0.weight`, ...), on
torch CPU ops (the reference itself is torch ops; F.linear / F.conv2d are the same aten kernels).
`conv2d_np` / `conv_transpose2d_np` are independent numpy restatements of the two convolution
definitions, used to pin the layout/stride/padding semantics without torch.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def mlp_encoder(sd, prefix, x, n_hidden=1):
    """Encoder_VAE_MLP.forward: reshape(-1, prod D) -> [Linear+ReLU]*(1+n_hidden) -> two heads."""
    w0 = sd[prefix + "layers.0.0.weight"]
    h = x.reshape(-1, w0.shape[1])
    for i in range(1 + n_hidden):
        h = F.relu(F.linear(h, sd[f"{prefix}layers.{i}.0.weight"], sd[f"{prefix}layers.{i}.0.bias"]))
    mu = F.linear(h, sd[prefix + "embedding.weight"], sd[prefix + "embedding.bias"])
    lv = F.linear(h, sd[prefix + "log_var.weight"], sd[prefix + "log_var.bias"])
    return mu, lv


def mlp_style_encoder(sd, prefix, x):
    """Encoder_VAE_MLP_Style.forward (default_architectures.py:75-141): Linear+ReLU -> four heads
    (embedding, log_var, style_embedding, style_log_var)."""
    w0 = sd[prefix + "layers.0.0.weight"]
    h = F.relu(F.linear(x.reshape(-1, w0.shape[1]), w0, sd[prefix + "layers.0.0.bias"]))
    head = lambda n: F.linear(h, sd[f"{prefix}{n}.weight"], sd[f"{prefix}{n}.bias"])
    return head("embedding"), head("log_var"), head("style_embedding"), head("style_log_var")


def build_default_mlp_multilatent(sd, input_dims):
    """BaseDictEncoders_MultiLatents / BaseDictDecodersMultiLatents (default_architectures.py:161-221)."""
    enc = {m: (lambda x, m=m: mlp_style_encoder(sd, f"encoders.{m}.", x)) for m in input_dims}
    dec = {m: (lambda z, m=m: mlp_decoder(sd, f"decoders.{m}.", z, tuple(input_dims[m]))) for m in input_dims}
    return enc, dec


def mlp_decoder(sd, prefix, z, input_dim):
    """Decoder_AE_MLP.forward: Linear(L,512)+ReLU -> Linear(512, prod D)+Sigmoid -> reshape(*z.shape[:-1], *D)."""
    h = F.relu(F.linear(z, sd[prefix + "layers.0.0.weight"], sd[prefix + "layers.0.0.bias"]))
    h = torch.sigmoid(F.linear(h, sd[prefix + "layers.1.0.weight"], sd[prefix + "layers.1.0.bias"]))
    return h.reshape(*z.shape[:-1], *input_dim)


def svhn_encoder(sd, prefix, x):
    """Encoder_VAE_SVHN.forward: 3x Conv(4,2,1)+ReLU, two Conv(4,2,0) heads, `.squeeze()`.

    `.squeeze()` removes ALL unit dims, so B == 1 collapses the batch dim like the reference (Appendix D).
    """
    h = x
    for i in (0, 2, 4):
        h = F.relu(F.conv2d(h, sd[f"{prefix}enc.{i}.weight"], sd[f"{prefix}enc.{i}.bias"], stride=2, padding=1))
    mu = F.conv2d(h, sd[prefix + "c1.weight"], sd[prefix + "c1.bias"], stride=2, padding=0).squeeze()
    lv = F.conv2d(h, sd[prefix + "c2.weight"], sd[prefix + "c2.bias"], stride=2, padding=0).squeeze()
    return mu, lv


def svhn_decoder(sd, prefix, z):
    """Decoder_VAE_SVHN.forward: z[...,L] -> ConvT(4,1,0)+ReLU -> 2x ConvT(4,2,1)+ReLU -> ConvT(4,2,1)+Sigmoid."""
    lead = z.shape[:-1]
    h = z.reshape(-1, z.shape[-1], 1, 1)
    h = F.relu(F.conv_transpose2d(h, sd[prefix + "dec.0.weight"], sd[prefix + "dec.0.bias"], stride=1, padding=0))
    h = F.relu(F.conv_transpose2d(h, sd[prefix + "dec.2.weight"], sd[prefix + "dec.2.bias"], stride=2, padding=1))
    h = F.relu(F.conv_transpose2d(h, sd[prefix + "dec.4.weight"], sd[prefix + "dec.4.bias"], stride=2, padding=1))
    h = torch.sigmoid(F.conv_transpose2d(h, sd[prefix + "dec.6.weight"], sd[prefix + "dec.6.bias"], stride=2, padding=1))
    return h.reshape(*lead, *h.shape[1:])


def build_mnist_svhn(sd, latent_dim=20):
    """Encoder / decoder callables for the MnistSvhn architecture of examples/distributed_training.py:42-50
    (MLP for mnist, conv for svhn) bound to state dict `sd` with the BaseMultiVAE key prefixes."""
    enc = {
        "mnist": lambda x: mlp_encoder(sd, "encoders.mnist.", x),
        "svhn": lambda x: svhn_encoder(sd, "encoders.svhn.", x),
    }
    dec = {
        "mnist": lambda z: mlp_decoder(sd, "decoders.mnist.", z, (1, 28, 28)),
        "svhn": lambda z: svhn_decoder(sd, "decoders.svhn.", z),
    }
    return enc, dec


def joint_mlp_encoder(sd, input_dims, data, prefix="joint_encoder."):
    """MultipleHeadJointEncoder.forward (default_architectures.py:303-322): embeddings of the copied unimodal encoders,
    concatenated in modality order -> [Linear+ReLU]* -> fc1 / fc2 heads.  -> (mu, log_var)."""
    embs = [mlp_encoder(sd, f"{prefix}encoders.{m}.", data[m])[0] for m in input_dims]
    h = torch.cat(embs, dim=1)
    i = 0
    while f"{prefix}enc.{i}.0.weight" in sd:
        h = F.relu(F.linear(h, sd[f"{prefix}enc.{i}.0.weight"], sd[f"{prefix}enc.{i}.0.bias"]))
        i += 1
    return (F.linear(h, sd[prefix + "fc1.weight"], sd[prefix + "fc1.bias"]),
            F.linear(h, sd[prefix + "fc2.weight"], sd[prefix + "fc2.bias"]))


def joint_encoder_generic(sd, enc_fns, data, prefix="joint_encoder."):
    """MultipleHeadJointEncoder.forward (default_architectures.py:303-322) over arbitrary unimodal encoders:
    enc_fns = {modality: f(sd, prefix, x) -> (mu, log_var, ...)} evaluated on the joint encoder's own copies
    (`joint_encoder.encoders.<m>.`), embeddings concatenated in modality order -> MLP -> fc1 / fc2."""
    embs = [enc_fns[m](sd, f"{prefix}encoders.{m}.", data[m])[0] for m in enc_fns]
    h = torch.cat(embs, dim=1)
    i = 0
    while f"{prefix}enc.{i}.0.weight" in sd:
        h = F.relu(F.linear(h, sd[f"{prefix}enc.{i}.0.weight"], sd[f"{prefix}enc.{i}.0.bias"]))
        i += 1
    return (F.linear(h, sd[prefix + "fc1.weight"], sd[prefix + "fc1.bias"]),
            F.linear(h, sd[prefix + "fc2.weight"], sd[prefix + "fc2.bias"]))


def resnet_block(sd, prefix, x, order="post"):
    """ResnetBlock.  "post" (mmnist.py:229-246): x_s + 0.1 * lrelu(conv2(lrelu(conv1(x)))); "pre" (cub.py:274-280):
    x_s + 0.1 * conv2(lrelu(conv1(lrelu(x)))).  x_s = 1x1 shortcut convolution when the channel counts differ."""
    lr = lambda t: F.leaky_relu(t, 0.2)
    c1 = lambda t: F.conv2d(t, sd[prefix + "conv_layers.0.weight"], sd[prefix + "conv_layers.0.bias"], 1, 1)
    c2 = lambda t: F.conv2d(t, sd[prefix + "conv_layers.2.weight"], sd.get(prefix + "conv_layers.2.bias"), 1, 1)
    xs = F.conv2d(x, sd[prefix + "shortcut_layer.weight"]) if (prefix + "shortcut_layer.weight") in sd else x
    dx = lr(c2(lr(c1(x)))) if order == "post" else c2(lr(c1(lr(x))))
    return xs + 0.1 * dx


def mmnist_resnet_encoder(sd, prefix, x):
    """EncoderResnetMMNIST.forward (mmnist.py:296-321) -> (mu_u, lv_u, mu_w, lv_w) (the last two None without w)."""
    outs = {}
    for tag in ("u", "w"):
        if f"{prefix}conv_img_{tag}.weight" not in sd:
            outs[tag] = (None, None)
            continue
        h = F.conv2d(x, sd[f"{prefix}conv_img_{tag}.weight"], sd[f"{prefix}conv_img_{tag}.bias"], 1, 1)
        h = resnet_block(sd, f"{prefix}resnet_{tag}.0.", h)
        h = resnet_block(sd, f"{prefix}resnet_{tag}.2.", F.avg_pool2d(h, 3, 2, 1))
        h = resnet_block(sd, f"{prefix}resnet_{tag}.4.", F.avg_pool2d(h, 3, 2, 1))
        h = h.reshape(h.shape[0], -1)
        outs[tag] = (F.linear(h, sd[f"{prefix}fc_mu_{tag}.weight"], sd[f"{prefix}fc_mu_{tag}.bias"]),
                     F.linear(h, sd[f"{prefix}fc_lv_{tag}.weight"], sd[f"{prefix}fc_lv_{tag}.bias"]))
    return outs["u"][0], outs["u"][1], outs["w"][0], outs["w"][1]


def mmnist_resnet_decoder(sd, prefix, z):
    """DecoderResnetMMNIST.forward (mmnist.py:355-366)."""
    lead = z.shape[:-1]
    h = F.linear(z.reshape(-1, z.shape[-1]), sd[prefix + "fc.weight"], sd[prefix + "fc.bias"]).view(-1, 256, 7, 7)
    h = F.interpolate(resnet_block(sd, prefix + "resnet.0.", h), scale_factor=2)
    h = F.interpolate(resnet_block(sd, prefix + "resnet.2.", h), scale_factor=2)
    h = resnet_block(sd, prefix + "resnet.4.", h)
    h = F.leaky_relu(F.conv2d(h, sd[prefix + "conv_img.0.weight"], sd[prefix + "conv_img.0.bias"], 1, 1), 0.2)
    return h.reshape(*lead, *h.shape[1:])


def _cub_block(sd, prefix, x):
    """cub.py ResnetBlock (conv_0 / conv_1 / conv_s naming, pre-activation order :274-280)."""
    lr = lambda t: F.leaky_relu(t, 0.2)
    xs = F.conv2d(x, sd[prefix + "conv_s.weight"]) if (prefix + "conv_s.weight") in sd else x
    dx = F.conv2d(lr(x), sd[prefix + "conv_0.weight"], sd[prefix + "conv_0.bias"], 1, 1)
    dx = F.conv2d(lr(dx), sd[prefix + "conv_1.weight"], sd[prefix + "conv_1.bias"], 1, 1)
    return xs + 0.1 * dx


def cub_resnet_encoder(sd, prefix, x):
    """CUB_Resnet_Encoder.forward (cub.py:186-196), default s0 = 16."""
    h = F.conv2d(x, sd[prefix + "conv_img.weight"], sd[prefix + "conv_img.bias"], 1, 1)
    h = _cub_block(sd, prefix + "resnet.0.", h)
    h = _cub_block(sd, prefix + "resnet.2.", F.avg_pool2d(h, 3, 2, 1))
    h = _cub_block(sd, prefix + "resnet.4.", F.avg_pool2d(h, 3, 2, 1))
    h = F.leaky_relu(h.reshape(h.shape[0], -1), 0.2)
    return (F.linear(h, sd[prefix + "fc_mu.weight"], sd[prefix + "fc_mu.bias"]),
            F.linear(h, sd[prefix + "fc_logvar.weight"], sd[prefix + "fc_logvar.bias"]))


def cub_resnet_decoder(sd, prefix, z):
    """CUB_Resnet_Decoder.forward (cub.py:238-247); leading dims flattened (the reference takes [B, L] only)."""
    lead = z.shape[:-1]
    h = F.linear(z.reshape(-1, z.shape[-1]), sd[prefix + "fc.weight"], sd[prefix + "fc.bias"]).view(-1, 256, 16, 16)
    h = F.interpolate(_cub_block(sd, prefix + "resnet.0.", h), scale_factor=2)
    h = F.interpolate(_cub_block(sd, prefix + "resnet.2.", h), scale_factor=2)
    h = _cub_block(sd, prefix + "resnet.4.", h)
    h = F.conv2d(F.leaky_relu(h, 0.2), sd[prefix + "conv_img.weight"], sd[prefix + "conv_img.bias"], 1, 1)
    return h.reshape(*lead, *h.shape[1:])


def build_default_mlp(sd, input_dims):
    """Default architectures (BaseDictEncoders / BaseDictDecoders, default_architectures.py:143-222)."""
    enc = {m: (lambda x, m=m: mlp_encoder(sd, f"encoders.{m}.", x)) for m in input_dims}
    dec = {m: (lambda z, m=m: mlp_decoder(sd, f"decoders.{m}.", z, tuple(input_dims[m]))) for m in input_dims}
    return enc, dec


# ----------------------------------------------------------------------------------------------
# numpy pins of the two convolution definitions (small inputs only: O(N*C*C*H*W*k*k) python-free einsum)
# ----------------------------------------------------------------------------------------------
def conv2d_np(x, w, b, stride, padding):
    """out[n,co,oh,ow] = b[co] + sum_{ci,kh,kw} x[n,ci,oh*s-p+kh, ow*s-p+kw] * w[co,ci,kh,kw]."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    N, C, H, W = x.shape
    CO, _, KH, KW = w.shape
    xp = np.zeros((N, C, H + 2 * padding, W + 2 * padding))
    xp[:, :, padding : padding + H, padding : padding + W] = x
    OH = (H + 2 * padding - KH) // stride + 1
    OW = (W + 2 * padding - KW) // stride + 1
    out = np.zeros((N, CO, OH, OW))
    for kh in range(KH):
        for kw in range(KW):
            patch = xp[:, :, kh : kh + stride * OH : stride, kw : kw + stride * OW : stride]
            out += np.einsum("nchw,oc->nohw", patch, w[:, :, kh, kw])
    return out + np.asarray(b, np.float64)[None, :, None, None]


def conv_transpose2d_np(x, w, b, stride, padding):
    """out[n,co,ih*s-p+kh, iw*s-p+kw] += x[n,ci,ih,iw] * w[ci,co,kh,kw]; weight layout [Cin,Cout,KH,KW]."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    N, C, H, W = x.shape
    _, CO, KH, KW = w.shape
    FH = (H - 1) * stride + KH
    FW = (W - 1) * stride + KW
    full = np.zeros((N, CO, FH, FW))
    for kh in range(KH):
        for kw in range(KW):
            full[:, :, kh : kh + stride * H : stride, kw : kw + stride * W : stride] += np.einsum(
                "nchw,co->nohw", x, w[:, :, kh, kw]
            )
    out = full[:, :, padding : FH - padding, padding : FW - padding]
    return out + np.asarray(b, np.float64)[None, :, None, None]


def count_params(sd):
    return int(sum(math.prod(v.shape) for v in sd.values()))
