"""Plugin bases (what the reference takes from `pythae.models.nn.base_architectures`): a user encoder /
decoder is any `nn.Module` subclass of these whose forward returns a `ModelOutput` with `embedding`,
`log_covariance` (encoders) or `reconstruction` (decoders).  SURVEY.md §8(b1)."""
import torch.nn as nn


class BaseEncoder(nn.Module):
    def __init__(self):
        nn.Module.__init__(self)

    def forward(self, x):
        raise NotImplementedError()


class BaseDecoder(nn.Module):
    # True: forward treats the rows of z independently of one another (no batch statistics, no dropout / RNG, contiguous
    # output), so MMVAE / MMVAE+ may decode the latents of all conditioning modalities in ONE stacked pass.  The in-package
    # decoders say True; a user-written decoder keeps the reference's one pass per (conditioning, target) pair unless it does.
    rows_independent = False

    def __init__(self):
        nn.Module.__init__(self)

    def forward(self, z):
        raise NotImplementedError()


class BaseJointEncoder(nn.Module):
    """Joint encoder plugin base (`multivae/models/nn/base_architectures.py`): forward(x: dict of modality tensors)
    returns a ModelOutput with `embedding` and `log_covariance`."""

    def __init__(self):
        nn.Module.__init__(self)
        self.latent_dim = None

    def forward(self, x: dict):
        raise NotImplementedError()


class BaseMultilatentEncoder(BaseEncoder):
    """Encoder plugin base for models with a shared and a private latent per modality (MMVAE+): forward returns a
    ModelOutput with `embedding`, `log_covariance`, `style_embedding`, `style_log_covariance`."""

    def __init__(self):
        BaseEncoder.__init__(self)
        self.latent_dim = None
        self.style_dim = None
