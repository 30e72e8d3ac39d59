from .base_ae_model import BaseMultiVAE
from .base_config import BaseAEConfig, BaseConfig, BaseMultiVAEConfig, EnvironmentConfig
from .base_model import BaseModel
from .base_utils import ModelOutput

__all__ = ["BaseMultiVAE", "BaseAEConfig", "BaseConfig", "BaseMultiVAEConfig", "EnvironmentConfig", "BaseModel",
           "ModelOutput"]
