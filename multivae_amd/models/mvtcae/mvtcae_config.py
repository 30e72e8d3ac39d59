from pydantic.dataclasses import dataclass

from ..base.base_config import BaseMultiVAEConfig


@dataclass
class MVTCAEConfig(BaseMultiVAEConfig):
    """`multivae/models/mvtcae/mvtcae_config.py`: alpha weights the total-correlation ratio, beta all KLs.
    K is the same Monte-Carlo extension as in MoPoEConfig (K = 1 = reference)."""

    alpha: float = 0.1
    beta: float = 2.5
    K: int = 1
