from pydantic.dataclasses import dataclass

from ..base.base_config import BaseMultiVAEConfig


@dataclass
class BaseJointModelConfig(BaseMultiVAEConfig):
    """`multivae/models/joint_models/joint_model_config.py`: base config of the models with a joint encoder."""

    pass
