from .base import BaseTrainer, BaseTrainerConfig, TrainingCallback
from .flat import FlatParams, FusedAdam

__all__ = ["BaseTrainer", "BaseTrainerConfig", "TrainingCallback", "FlatParams", "FusedAdam"]
