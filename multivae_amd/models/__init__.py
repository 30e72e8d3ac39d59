from .base import BaseAEConfig, BaseMultiVAE, BaseMultiVAEConfig, ModelOutput
from .jmvae import JMVAE, JMVAEConfig
from .joint_models import BaseJointModel, BaseJointModelConfig
from .mmvae import MMVAE, MMVAEConfig
from .mmvaePlus import MMVAEPlus, MMVAEPlusConfig
from .mopoe import MoPoE, MoPoEConfig
from .mvae import MVAE, MVAEConfig
from .mvtcae import MVTCAE, MVTCAEConfig
from .crmvae import CRMVAE, CRMVAEConfig
from .dmvae import DMVAE, DMVAEConfig
from .auto_model import AutoConfig, AutoModel  # noqa: E402  (needs the model classes above)

__all__ = ["BaseAEConfig", "BaseMultiVAE", "BaseMultiVAEConfig", "ModelOutput", "MMVAE", "MMVAEConfig", "MoPoE",
           "MoPoEConfig", "MVTCAE", "MVTCAEConfig", "JMVAE", "JMVAEConfig", "BaseJointModel", "BaseJointModelConfig", "MMVAEPlus", "MMVAEPlusConfig", "AutoModel", "AutoConfig", "MVAE", "MVAEConfig", "CRMVAE", "CRMVAEConfig", "DMVAE", "DMVAEConfig"]
