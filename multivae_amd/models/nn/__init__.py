from .base_architectures import BaseDecoder, BaseEncoder, BaseJointEncoder, BaseMultilatentEncoder
from .default_architectures import (BaseDictDecoders, BaseDictDecodersMultiLatents, BaseDictEncoders,
                                    BaseDictEncoders_MultiLatents, Decoder_AE_MLP, Encoder_VAE_MLP,
                                    Encoder_VAE_MLP_Style, MultipleHeadJointEncoder)
from .svhn import Decoder_VAE_SVHN, Encoder_VAE_SVHN

__all__ = ["BaseDecoder", "BaseEncoder", "BaseJointEncoder", "BaseMultilatentEncoder", "BaseDictDecoders",
           "BaseDictDecodersMultiLatents", "BaseDictEncoders", "BaseDictEncoders_MultiLatents", "Decoder_AE_MLP",
           "Encoder_VAE_MLP", "Encoder_VAE_MLP_Style", "MultipleHeadJointEncoder", "Decoder_VAE_SVHN",
           "Encoder_VAE_SVHN"]
