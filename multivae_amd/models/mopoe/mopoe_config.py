from typing import Dict, List, Union

from pydantic.dataclasses import dataclass

from ..base.base_config import BaseMultiVAEConfig


@dataclass
class MoPoEConfig(BaseMultiVAEConfig):
    """`multivae/models/mopoe/mopoe_config.py:8-46` plus one extension field:

    K (int): number of latent samples of the Monte-Carlo reconstruction term.  K = 1 (default) is exactly
        the reference's ELBO; K > 1 averages the reconstruction term over K reparameterised samples while
        the analytic KL is unchanged (SURVEY.md §0 D1 — the reference's training forward has no K).
    """

    subsets: Union[List[list], Dict[str, list], None] = None
    beta: float = 1.0
    beta_style: float = 1.0
    modalities_specific_dim: Union[dict, None] = None
    K: int = 1
