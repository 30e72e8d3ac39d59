from .base_architectures import BaseDecoder, BaseEncoder
from .default_architectures import BaseDictDecoders, BaseDictEncoders, Decoder_AE_MLP, Encoder_VAE_MLP
from .svhn import Decoder_VAE_SVHN, Encoder_VAE_SVHN

__all__ = ["BaseDecoder", "BaseEncoder", "BaseDictDecoders", "BaseDictEncoders", "Decoder_AE_MLP",
           "Encoder_VAE_MLP", "Decoder_VAE_SVHN", "Encoder_VAE_SVHN"]
