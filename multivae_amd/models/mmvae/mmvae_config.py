from typing import Literal

from pydantic.dataclasses import dataclass

from ..base.base_config import BaseMultiVAEConfig


@dataclass
class MMVAEConfig(BaseMultiVAEConfig):
    """`multivae/models/mmvae/mmvae_config.py:9-52`.  `beta` is kept for layout compatibility; like in the
    reference it does not enter the loss."""

    K: int = 10
    prior_and_posterior_dist: Literal["laplace_with_softmax", "normal"] = "laplace_with_softmax"
    learn_prior: bool = True
    beta: float = 1.0
    loss: Literal["iwae_looser", "dreg_looser"] = "dreg_looser"
