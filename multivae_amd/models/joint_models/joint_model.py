"""Base class of the models with a joint encoder (`multivae/models/joint_models/joint_model.py:20-84`)."""
from typing import Union

import torch

from ... import kernels
from ..base import BaseMultiVAE
from ..nn.base_architectures import BaseJointEncoder
from ..nn.default_architectures import MultipleHeadJointEncoder
from .joint_model_config import BaseJointModelConfig


class BaseJointModel(BaseMultiVAE):
    def __init__(self, model_config: BaseJointModelConfig, encoders: dict = None, decoders: dict = None,
                 joint_encoder: Union[BaseJointEncoder, None] = None, **kwargs):
        super().__init__(model_config, encoders, decoders)
        if joint_encoder is None:
            joint_encoder = self.default_joint_encoder(model_config)
        else:
            self.model_config.custom_architectures.append("joint_encoder")
        self.set_joint_encoder(joint_encoder)

    def default_joint_encoder(self, model_config):
        return MultipleHeadJointEncoder(self.encoders, model_config)

    def set_joint_encoder(self, joint_encoder):
        if not issubclass(type(joint_encoder), BaseJointEncoder):
            raise AttributeError("The joint encoder must inherit from "
                                 "~multivae.models.nn.default_architectures.BaseJointEncoder . Refer to documentation.")
        self.joint_encoder = joint_encoder

    def forward(self, inputs, **kwargs):
        if hasattr(inputs, "masks"):
            raise AttributeError("The inputs have masks but this model is not compatible with incomplete dataset.")

    def encode(self, inputs, cond_mod="all", N=1, return_mean=False, **kwargs):
        if hasattr(inputs, "masks"):
            raise AttributeError("The inputs have masks but this model is not compatible with incomplete dataset.")
        return super().encode(inputs, cond_mod, N, **kwargs)

    def compute_joint_nll(self, inputs, K: int = 1000, batch_size_K: int = 100, **kwargs):
        """-sum_b ln p(x_b) by importance sampling from the joint encoder's Gaussian (joint_model.py:82-154).
        kwargs: noise [K,B,L]."""
        self.eval()
        if hasattr(inputs, "masks"):
            raise AttributeError("The inputs contains masks but this model is not compatible with incomplete dataset.")
        with torch.no_grad():
            out = self.joint_encoder(inputs.data)
            mu, lv = out.embedding, out.log_covariance
            B, L = mu.shape
            sd = kernels.std_from_logvar(lv)
            z = kernels.iwae_sample(mu, sd, self._noise((int(K), B, L), mu.device, kwargs.get("noise")))
            return self._joint_nll(inputs, z, [mu], [sd])
