from .base_architectures import BaseDecoder, BaseEncoder, BaseJointEncoder
from .default_architectures import (BaseDictDecoders, BaseDictEncoders, Decoder_AE_MLP, Encoder_VAE_MLP,
                                    MultipleHeadJointEncoder)
from .svhn import Decoder_VAE_SVHN, Encoder_VAE_SVHN

__all__ = ["BaseDecoder", "BaseEncoder", "BaseJointEncoder", "BaseDictDecoders", "BaseDictEncoders", "Decoder_AE_MLP",
           "Encoder_VAE_MLP", "MultipleHeadJointEncoder", "Decoder_VAE_SVHN", "Encoder_VAE_SVHN"]
