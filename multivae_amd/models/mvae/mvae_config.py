from pydantic.dataclasses import dataclass

from ..base.base_config import BaseMultiVAEConfig


@dataclass
class MVAEConfig(BaseMultiVAEConfig):
    """`multivae/models/mvae/mvae_config.py`: sub-sampled training paradigm (joint + unimodal + k random subset ELBOs),
    KL annealing over `warmup` epochs, beta on the KL terms."""

    use_subsampling: bool = True
    k: int = 0
    warmup: int = 10
    beta: float = 1
