"""Bit-exact procedural test data shared by the golden generator and the tests.

Weights and inputs of the golden cases are NOT stored in the fixtures (a MnistSvhn state dict is
6 MB); they are regenerated from an integer hash (splitmix64 on uint64, exact on every platform)
so the committed fixtures only hold noise tensors and expected outputs.
"""
from collections import OrderedDict

import numpy as np


def hash_uniform(n, seed):
    """n float64 values in [0,1), pure integer arithmetic (splitmix64 finaliser)."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = z + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(shape, seed, lo=0.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (hi - lo) * hash_uniform(n, seed)).astype(np.float32).reshape(shape)


def name_seed(name):
    """Stable small integer derived from a parameter name (CRC32)."""
    import zlib

    return 777 + (zlib.crc32(name.encode()) % 100000)


def hash_indices(n, count, seed):
    """`count` pseudo-random flat indices into a tensor of n elements."""
    return (hash_uniform(count, seed) * n).astype(np.int64) % max(n, 1)


# ---- parameter shapes, reference key names -------------------------------
def mlp_encoder_shapes(prefix, in_features, latent_dim, n_hidden=1):
    s = OrderedDict()
    s[prefix + "layers.0.0.weight"] = (512, in_features)
    s[prefix + "layers.0.0.bias"] = (512,)
    for i in range(1, 1 + n_hidden):
        s[f"{prefix}layers.{i}.0.weight"] = (512, 512)
        s[f"{prefix}layers.{i}.0.bias"] = (512,)
    for h in ("embedding", "log_var"):
        s[f"{prefix}{h}.weight"] = (latent_dim, 512)
        s[f"{prefix}{h}.bias"] = (latent_dim,)
    return s


def mlp_decoder_shapes(prefix, latent_dim, out_features):
    s = OrderedDict()
    s[prefix + "layers.0.0.weight"] = (512, latent_dim)
    s[prefix + "layers.0.0.bias"] = (512,)
    s[prefix + "layers.1.0.weight"] = (out_features, 512)
    s[prefix + "layers.1.0.bias"] = (out_features,)
    return s


def svhn_encoder_shapes(prefix, latent_dim):
    s = OrderedDict()
    for i, (co, ci) in zip((0, 2, 4), ((32, 3), (64, 32), (128, 64))):
        s[f"{prefix}enc.{i}.weight"] = (co, ci, 4, 4)
        s[f"{prefix}enc.{i}.bias"] = (co,)
    for h in ("c1", "c2"):
        s[f"{prefix}{h}.weight"] = (latent_dim, 128, 4, 4)
        s[f"{prefix}{h}.bias"] = (latent_dim,)
    return s


def svhn_decoder_shapes(prefix, latent_dim):
    s = OrderedDict()
    for i, (ci, co) in zip((0, 2, 4, 6), ((latent_dim, 128), (128, 64), (64, 32), (32, 3))):
        s[f"{prefix}dec.{i}.weight"] = (ci, co, 4, 4)
        s[f"{prefix}dec.{i}.bias"] = (co,)
    return s


def mnist_svhn_shapes(latent_dim=20):
    """Parameter order of BaseMultiVAE: decoders are registered before encoders (base_ae_model.py:86-87)."""
    s = OrderedDict()
    s.update(mlp_decoder_shapes("decoders.mnist.", latent_dim, 784))
    s.update(svhn_decoder_shapes("decoders.svhn.", latent_dim))
    s.update(mlp_encoder_shapes("encoders.mnist.", 784, latent_dim))
    s.update(svhn_encoder_shapes("encoders.svhn.", latent_dim))
    return s


def default_mlp_shapes(input_dims, latent_dim):
    s = OrderedDict()
    for m, d in input_dims.items():
        s.update(mlp_decoder_shapes(f"decoders.{m}.", latent_dim, int(np.prod(d))))
    for m, d in input_dims.items():
        s.update(mlp_encoder_shapes(f"encoders.{m}.", int(np.prod(d)), latent_dim))
    return s


def joint_mlp_encoder_shapes(input_dims, latent_dim, hidden_dim=512, n_hidden_layers=2):
    """MultipleHeadJointEncoder built from default MLP encoders (default_architectures.py:261-322): copies of the
    unimodal encoders, the unifying MLP and its two heads, in state_dict order."""
    s = OrderedDict()
    for m, d in input_dims.items():
        s.update(mlp_encoder_shapes(f"joint_encoder.encoders.{m}.", int(np.prod(d)), latent_dim))
    s["joint_encoder.enc.0.0.weight"] = (hidden_dim, latent_dim * len(input_dims))
    s["joint_encoder.enc.0.0.bias"] = (hidden_dim,)
    for i in range(1, n_hidden_layers):
        s[f"joint_encoder.enc.{i}.0.weight"] = (hidden_dim, hidden_dim)
        s[f"joint_encoder.enc.{i}.0.bias"] = (hidden_dim,)
    for h in ("fc1", "fc2"):
        s[f"joint_encoder.{h}.weight"] = (latent_dim, hidden_dim)
        s[f"joint_encoder.{h}.bias"] = (latent_dim,)
    return s


def jmvae_mlp_shapes(input_dims, latent_dim):
    """JMVAE with all-default architectures: decoders, encoders, joint encoder (BaseJointModel registration order)."""
    s = default_mlp_shapes(input_dims, latent_dim)
    s.update(joint_mlp_encoder_shapes(input_dims, latent_dim))
    return s


def mlp_style_encoder_shapes(prefix, in_features, latent_dim, style_dim):
    s = OrderedDict()
    s[prefix + "layers.0.0.weight"] = (512, in_features)
    s[prefix + "layers.0.0.bias"] = (512,)
    for h, d in (("embedding", latent_dim), ("log_var", latent_dim), ("style_embedding", style_dim),
                 ("style_log_var", style_dim)):
        s[f"{prefix}{h}.weight"] = (d, 512)
        s[f"{prefix}{h}.bias"] = (d,)
    return s


def mmvaeplus_mlp_shapes(input_dims, latent_dim, style_dim):
    """MMVAEPlus with default architectures: decoders take latent_dim + style_dim inputs, encoders have four heads."""
    s = OrderedDict()
    for m, d in input_dims.items():
        s.update(mlp_decoder_shapes(f"decoders.{m}.", latent_dim + style_dim, int(np.prod(d))))
    for m, d in input_dims.items():
        s.update(mlp_style_encoder_shapes(f"encoders.{m}.", int(np.prod(d)), latent_dim, style_dim))
    return s


def mopoe_style_mlp_shapes(input_dims, latent_dim, style_dims):
    """MoPoE with modality-specific latent spaces and default architectures (mopoe_model.py:58-75): like MMVAEPlus but
    with one private dimension per modality (`style_dims`: {modality: dim})."""
    s = OrderedDict()
    for m, d in input_dims.items():
        s.update(mlp_decoder_shapes(f"decoders.{m}.", latent_dim + style_dims[m], int(np.prod(d))))
    for m, d in input_dims.items():
        s.update(mlp_style_encoder_shapes(f"encoders.{m}.", int(np.prod(d)), latent_dim, style_dims[m]))
    return s


def resnet_block_shapes(prefix, cin, cout, chid=None):
    """ResnetBlock (mmnist.py:214-252 / cub.py:250-293): conv_layers.0, conv_layers.2 (3x3), optional 1x1 shortcut."""
    chid = min(cin, cout) if chid is None else chid
    s = OrderedDict()
    s[prefix + "conv_layers.0.weight"] = (chid, cin, 3, 3)
    s[prefix + "conv_layers.0.bias"] = (chid,)
    s[prefix + "conv_layers.2.weight"] = (cout, chid, 3, 3)
    s[prefix + "conv_layers.2.bias"] = (cout,)
    if cin != cout:
        s[prefix + "shortcut_layer.weight"] = (cout, cin, 1, 1)
    return s


def mmnist_resnet_encoder_shapes(private_dim, shared_dim, prefix=""):
    """EncoderResnetMMNIST (mmnist.py:255-321) in state_dict order: w branch (if any) first, then u."""
    s = OrderedDict()
    for tag, dim in (("w", private_dim), ("u", shared_dim)):
        if tag == "w" and private_dim <= 0:
            continue
        s[f"{prefix}conv_img_{tag}.weight"] = (64, 3, 3, 3)
        s[f"{prefix}conv_img_{tag}.bias"] = (64,)
        s.update(resnet_block_shapes(f"{prefix}resnet_{tag}.0.", 64, 64))
        s.update(resnet_block_shapes(f"{prefix}resnet_{tag}.2.", 64, 128))
        s.update(resnet_block_shapes(f"{prefix}resnet_{tag}.4.", 128, 256))
        for h in ("fc_mu", "fc_lv"):
            s[f"{prefix}{h}_{tag}.weight"] = (dim, 256 * 49)
            s[f"{prefix}{h}_{tag}.bias"] = (dim,)
    return s


def mmnist_resnet_decoder_shapes(latent_dim, prefix=""):
    """DecoderResnetMMNIST (mmnist.py:324-366)."""
    s = OrderedDict()
    s[prefix + "fc.weight"] = (256 * 49, latent_dim)
    s[prefix + "fc.bias"] = (256 * 49,)
    s.update(resnet_block_shapes(prefix + "resnet.0.", 256, 128))
    s.update(resnet_block_shapes(prefix + "resnet.2.", 128, 64))
    s.update(resnet_block_shapes(prefix + "resnet.4.", 64, 64))
    s[prefix + "conv_img.0.weight"] = (3, 64, 3, 3)
    s[prefix + "conv_img.0.bias"] = (3,)
    return s


def cub_block_shapes(prefix, fin, fout):
    s = OrderedDict()
    fh = min(fin, fout)
    s[prefix + "conv_0.weight"] = (fh, fin, 3, 3)
    s[prefix + "conv_0.bias"] = (fh,)
    s[prefix + "conv_1.weight"] = (fout, fh, 3, 3)
    s[prefix + "conv_1.bias"] = (fout,)
    if fin != fout:
        s[prefix + "conv_s.weight"] = (fout, fin, 1, 1)
    return s


def cub_resnet_encoder_shapes(latent_dim, prefix=""):
    """CUB_Resnet_Encoder (cub.py:144-196), defaults s0=16, nfilter=64: blocks (64,64) (64,128) (128,256)."""
    s = OrderedDict()
    s[prefix + "conv_img.weight"] = (64, 3, 3, 3)
    s[prefix + "conv_img.bias"] = (64,)
    s.update(cub_block_shapes(prefix + "resnet.0.", 64, 64))
    s.update(cub_block_shapes(prefix + "resnet.2.", 64, 128))
    s.update(cub_block_shapes(prefix + "resnet.4.", 128, 256))
    for h in ("fc_mu", "fc_logvar"):
        s[f"{prefix}{h}.weight"] = (latent_dim, 256 * 256)
        s[f"{prefix}{h}.bias"] = (latent_dim,)
    return s


def cub_resnet_decoder_shapes(latent_dim, prefix=""):
    """CUB_Resnet_Decoder (cub.py:199-247), defaults."""
    s = OrderedDict()
    s[prefix + "fc.weight"] = (256 * 256, latent_dim)
    s[prefix + "fc.bias"] = (256 * 256,)
    s.update(cub_block_shapes(prefix + "resnet.0.", 256, 128))
    s.update(cub_block_shapes(prefix + "resnet.2.", 128, 64))
    s.update(cub_block_shapes(prefix + "resnet.4.", 64, 64))
    s[prefix + "conv_img.weight"] = (3, 64, 3, 3)
    s[prefix + "conv_img.bias"] = (3,)
    return s


def make_state_dict(shapes, seed, gain=1.0):
    """name -> float32 ndarray, U(-b, b) with b = gain/sqrt(prod(shape[1:])) (bias: b of its weight)."""
    sd = OrderedDict()
    bound = 1.0
    for i, (name, shape) in enumerate(shapes.items()):
        if name.endswith(".weight"):
            bound = gain / float(np.sqrt(np.prod(shape[1:])))
        sd[name] = uniform(shape, seed * 1000 + i, -bound, bound)
    return sd


def mmvaeplus_resnet_shapes(names, private_dim, shared_dim):
    """MMVAEPlus with the PolyMNIST ResNets (BASELINE configs[3]): decoders first, then encoders (base_ae_model.py:86-87)."""
    s = OrderedDict()
    for m in names:
        s.update(mmnist_resnet_decoder_shapes(private_dim + shared_dim, f"decoders.{m}."))
    for m in names:
        s.update(mmnist_resnet_encoder_shapes(private_dim, shared_dim, f"encoders.{m}."))
    return s


def jmvae_cub_shapes(latent_dim, n_attr, hidden_dim=512, n_hidden_layers=2):
    """JMVAE on (image 3x64x64: CUB ResNets, attributes [n_attr]: default MLPs) with the default joint encoder
    (BASELINE configs[4]): decoders, encoders, then the joint encoder's copies + MLP + heads."""
    s = OrderedDict()
    s.update(cub_resnet_decoder_shapes(latent_dim, "decoders.image."))
    s.update(mlp_decoder_shapes("decoders.attributes.", latent_dim, n_attr))
    s.update(cub_resnet_encoder_shapes(latent_dim, "encoders.image."))
    s.update(mlp_encoder_shapes("encoders.attributes.", n_attr, latent_dim))
    s.update(cub_resnet_encoder_shapes(latent_dim, "joint_encoder.encoders.image."))
    s.update(mlp_encoder_shapes("joint_encoder.encoders.attributes.", n_attr, latent_dim))
    s["joint_encoder.enc.0.0.weight"] = (hidden_dim, 2 * latent_dim)
    s["joint_encoder.enc.0.0.bias"] = (hidden_dim,)
    for i in range(1, n_hidden_layers):
        s[f"joint_encoder.enc.{i}.0.weight"] = (hidden_dim, hidden_dim)
        s[f"joint_encoder.enc.{i}.0.bias"] = (hidden_dim,)
    for h in ("fc1", "fc2"):
        s[f"joint_encoder.{h}.weight"] = (latent_dim, hidden_dim)
        s[f"joint_encoder.{h}.bias"] = (latent_dim,)
    return s
