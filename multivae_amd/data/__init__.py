from .datasets import DatasetOutput, IncompleteDataset, MultimodalBaseDataset
from .utils import drop_unused_modalities, get_batch_size, set_inputs_to_device

__all__ = ["DatasetOutput", "IncompleteDataset", "MultimodalBaseDataset", "drop_unused_modalities",
           "get_batch_size", "set_inputs_to_device"]
