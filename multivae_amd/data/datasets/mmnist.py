"""`multivae/data/datasets/mmnist.py:22-172`: PolyMNIST (5 image modalities of the same digit, `MMNIST/<split>/m{i}.pt` +
`labels.pt`) with the `missing_ratio` switch that simulates missing-at-random modalities: the SAME masks as the reference
(Bernoulli(1 - missing_ratio) per modality from `torch.Generator().manual_seed(i)`, modality m0 always present), the
content of missing samples erased, and the reference's length rule for `keep_incomplete=False`.

`.data` / `.masks` are dicts of whole tensors, so the trainer's device-resident batch iterator applies (one index_select per
modality per batch)."""
import math
import os
from typing import Literal

import torch

from .base import DatasetOutput, MultimodalBaseDataset


class MMNISTDataset(MultimodalBaseDataset):
    def __init__(self, data_path: str, transform=None, target_transform=None, split: Literal["train", "test"] = "train",
                 download: bool = False, missing_ratio: float = 0, keep_incomplete: bool = True):
        if isinstance(data_path, str):
            data_path = os.path.expanduser(data_path)
        unimodal_datapaths = [os.path.join(data_path, "MMNIST", split, f"m{i}.pt") for i in range(5)]
        self.num_modalities = len(unimodal_datapaths)
        self.unimodal_datapaths = unimodal_datapaths
        self.transform = transform
        self.target_transform = target_transform
        self.download = download
        self.missing_ratio = missing_ratio
        self.keep_incomplete = keep_incomplete
        self.__check_or_download_data__(data_path, unimodal_datapaths)
        self.images_dict = {f"m{i}": torch.load(p, weights_only=True) for i, p in enumerate(unimodal_datapaths)}
        for k, v in self.images_dict.items():
            setattr(self, k, v)
        self.labels = torch.load(os.path.join(data_path, "MMNIST", split, "labels.pt"), weights_only=True)
        assert self.images_dict["m0"].shape[0] == self.labels.shape[0]
        self.num_files = self.labels.shape[0]
        self.data = self.images_dict
        if missing_ratio > 0 and self.keep_incomplete:
            self.masks = {}
            for i in range(5):  # the missing samples, reproducibly (mmnist.py:111-117)
                self.masks[f"m{i}"] = torch.bernoulli(torch.ones((self.num_files,)) * (1 - missing_ratio),
                                                      generator=torch.Generator().manual_seed(i)).bool()
            self.masks["m0"] = torch.ones((self.num_files,)).bool()  # at least one modality for every sample
            for k in self.masks:  # erase the content of the missing samples
                shape = (-1,) + (1,) * (self.images_dict[k].dim() - 1)
                self.images_dict[k] = self.images_dict[k] * self.masks[k].to(self.images_dict[k].dtype).view(shape)
            self.data = self.images_dict

    def __check_or_download_data__(self, data_path, unimodal_datapaths):
        if not os.path.exists(unimodal_datapaths[0]) and self.download:
            try:
                import tempfile

                from torchvision.datasets.utils import download_and_extract_archive

                download_and_extract_archive(url="https://zenodo.org/record/4899160/files/PolyMNIST.zip",
                                             download_root=tempfile.mkdtemp(), extract_root=data_path)
            except ImportError as e:
                raise AttributeError("download=True needs torchvision; place the PolyMNIST files under data_path") from e
        elif not os.path.exists(unimodal_datapaths[0]) and not self.download:
            raise AttributeError("The PolyMNIST dataset is not available at the given datapath and download is set to "
                                 "False.Set download to True or place the dataset in the data_path folder.")

    def __getitem__(self, index):
        images_dict = {k: self.images_dict[k][index] for k in self.images_dict}
        if self.missing_ratio == 0 or not self.keep_incomplete:
            return DatasetOutput(data=images_dict, labels=self.labels[index])
        masks_dict = {k: self.masks[k][index] for k in self.masks}
        return DatasetOutput(data=images_dict, labels=self.labels[index], masks=masks_dict)

    def __len__(self):
        if self.missing_ratio == 0 or self.keep_incomplete:
            return self.num_files
        # the reference keeps the first ceil((1 - r)^4 n) samples (the expected share of complete ones), mmnist.py:166-172
        return math.ceil((1 - self.missing_ratio) ** 4 * self.num_files)
