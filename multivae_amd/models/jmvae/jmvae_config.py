from pydantic.dataclasses import dataclass

from ..joint_models.joint_model_config import BaseJointModelConfig


@dataclass
class JMVAEConfig(BaseJointModelConfig):
    """`multivae/models/jmvae/jmvae_config.py`: alpha weights the unimodal/joint KL term (LJM), the regularisation is
    annealed linearly over `warmup` epochs, beta weights the KL to the prior (the reference's add-on)."""

    alpha: float = 0.1
    warmup: int = 10
    beta: float = 1.0
