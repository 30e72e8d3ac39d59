"""`multivae/data/datasets/utils.py:10-47`: ResampleDataset — `dataset[sampler(dataset, idx)]`.

Here the sampler is an index tensor (what `MnistSvhn` needs): `base[index[i]]`.  The trainer's device-resident batch
iterator recognises it (`base`, `index`) and gathers `base.to(device)[index[batch]]` on the GPU instead of materialising
the resampled copy (5 x the paired MnistSvhn set would be 4 GB of host memory)."""
import torch


class ResampleDataset(torch.utils.data.Dataset):
    def __init__(self, base: torch.Tensor, index: torch.Tensor, transform=None):
        self.base = base
        self.index = torch.as_tensor(index, dtype=torch.long)
        self.transform = transform

    def __len__(self):
        return len(self.index)

    def __getitem__(self, idx):
        i = self.index[idx]
        if torch.is_tensor(i) and i.numel() and (int(i.min()) < 0 or int(i.max()) >= len(self.base)):
            raise IndexError("out of range")
        x = self.base[i]
        return self.transform(x) if self.transform is not None else x
