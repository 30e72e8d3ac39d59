from .base import BaseTrainer, BaseTrainerConfig, TrainingCallback
from .flat import FlatParams, FusedAdam
from .graph import GraphedStep

__all__ = ["BaseTrainer", "BaseTrainerConfig", "TrainingCallback", "FlatParams", "FusedAdam", "GraphedStep"]
