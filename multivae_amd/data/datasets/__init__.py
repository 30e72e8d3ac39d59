from .base import DatasetOutput, IncompleteDataset, MultimodalBaseDataset
from .mmnist import MMNISTDataset
from .mnist_svhn import MnistSvhn
from .utils import ResampleDataset

__all__ = ["DatasetOutput", "IncompleteDataset", "MultimodalBaseDataset", "MMNISTDataset", "MnistSvhn", "ResampleDataset"]
