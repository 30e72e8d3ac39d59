from pydantic.dataclasses import dataclass

from ..base.base_config import BaseMultiVAEConfig


@dataclass
class CRMVAEConfig(BaseMultiVAEConfig):
    """`multivae/models/crmvae/crmvae_config.py`: beta weights the sum of all KLs."""

    beta: float = 2.5
