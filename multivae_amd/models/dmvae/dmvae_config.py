from typing import Union

from pydantic.dataclasses import dataclass

from ..base.base_config import BaseMultiVAEConfig


@dataclass
class DMVAEConfig(BaseMultiVAEConfig):
    """`multivae/models/dmvae/dmvae_config.py`: private latent dimensions per modality, beta on the shared KL, one beta
    per private KL."""

    modalities_specific_dim: dict = None
    modalities_specific_betas: Union[dict, None] = None
    beta: float = 1.0
