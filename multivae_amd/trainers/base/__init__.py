from .base_trainer import BaseTrainer, set_seed, shard_indices, update_dict
from .base_trainer_config import BaseTrainerConfig
from .callbacks import CallbackHandler, MetricConsolePrinterCallback, TrainingCallback

__all__ = ["BaseTrainer", "BaseTrainerConfig", "CallbackHandler", "MetricConsolePrinterCallback", "TrainingCallback",
           "set_seed", "shard_indices", "update_dict"]
