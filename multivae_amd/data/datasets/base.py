"""Dataset containers with the reference's contract (`multivae/data/datasets/base.py:8-202`):
items are `DatasetOutput(data={mod: tensor}, [masks={mod: bool}], [labels])` with key AND attribute access,
and `hasattr(inputs, "masks")` is the incompleteness test."""
import torch

from ..._output import ModelOutput


class DatasetOutput(ModelOutput):
    pass


class MultimodalBaseDataset(torch.utils.data.Dataset):
    def __init__(self, data: dict, labels=None):
        self.labels = labels
        self.data = data

    def __len__(self):
        length = len(self.data[list(self.data)[0]])
        for m in self.data:
            if len(self.data[m]) != length:
                raise AttributeError("The size of the provided datasets doesn't correspond between modalities!")
        return length

    def __getitem__(self, index):
        X = {m: self.data[m][index] for m in self.data}
        if self.labels is not None:
            return DatasetOutput(data=X, labels=self.labels[index])
        return DatasetOutput(data=X)

    def transform_for_plotting(self, tensor, modality):
        return tensor


class IncompleteDataset(MultimodalBaseDataset):
    def __init__(self, data: dict, masks: dict, labels=None):
        self.data = data
        self.masks = masks
        self.labels = labels
        length = len(self.data[list(self.data)[0]])
        for m in self.data:
            if len(self.data[m]) != length or len(self.masks[m]) != length:
                raise AttributeError(
                    "The size of the provided datasets/masks doesn't correspond between modalities!")
        if self.labels is not None and len(self.labels) != length:
            raise AttributeError("The size of the provided datasets/masks doesn't correspond with the labels")

    def __len__(self):
        return len(self.data[list(self.data)[0]])

    def __getitem__(self, index):
        X = {m: self.data[m][index] for m in self.data}
        mk = {m: self.masks[m][index] for m in self.masks}
        if self.labels is not None:
            return DatasetOutput(data=X, labels=self.labels[index], masks=mk)
        return DatasetOutput(data=X, masks=mk)
