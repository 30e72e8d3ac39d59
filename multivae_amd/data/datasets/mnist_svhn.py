"""`multivae/data/datasets/mnist_svhn.py:19-136`: the paired MNIST / SVHN dataset (same digit, `data_multiplication`
random pairings per class), same pairing files (`<data_path>/mnist_svhn_idx_data_mul_<k>/<split>/{mnist,svhn}_idx.pt`, so an
existing pairing made by the reference is reused as is), same item contract.

The reference reads the two sets through torchvision; this build reads the same files torchvision leaves on disk
(`MNIST/raw/*-idx?-ubyte[.gz]`, `{train,test}_32x32.mat`) without it.  There is no downloader here (`download=True`
defers to torchvision when it is importable)."""
import gzip
import logging
import os
from pathlib import Path

import numpy as np
import torch

from .base import MultimodalBaseDataset
from .utils import ResampleDataset

logger = logging.getLogger(__name__)


def _read_idx(path):
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    magic = int.from_bytes(raw[:4], "big")
    nd = magic & 0xFF
    dims = [int.from_bytes(raw[4 + 4 * i: 8 + 4 * i], "big") for i in range(nd)]
    return torch.from_numpy(np.frombuffer(raw, dtype=np.uint8, offset=4 + 4 * nd).reshape(dims).copy())


def _find(root, names):
    for n in names:
        for cand in (os.path.join(root, n), os.path.join(root, n + ".gz")):
            if os.path.exists(cand):
                return cand
    raise AttributeError(f"none of {names} found under {root}: place the files there (no downloader in this build)")


def load_mnist(data_path, train):
    """-> (images uint8 [n, 28, 28], labels int64 [n]) from torchvision's on-disk layout."""
    raw = os.path.join(data_path, "MNIST", "raw")
    pre = "train" if train else "t10k"
    x = _read_idx(_find(raw, [f"{pre}-images-idx3-ubyte"]))
    y = _read_idx(_find(raw, [f"{pre}-labels-idx1-ubyte"])).long()
    return x, y


def load_svhn(data_path, split):
    """-> (images uint8 [n, 3, 32, 32], labels int64 [n]) from `<split>_32x32.mat` (labels as stored: 10 means 0)."""
    from scipy.io import loadmat

    m = loadmat(_find(data_path, [f"{split}_32x32.mat"]))
    x = torch.from_numpy(np.transpose(m["X"], (3, 2, 0, 1)).copy())  # torchvision: [n, C, H, W]
    y = torch.from_numpy(m["y"].astype(np.int64).squeeze())
    return x, y


class MnistSvhn(MultimodalBaseDataset):
    def __init__(self, data_path, split: str = "train", download=False, data_multiplication=5, **kwargs):
        if split not in ["train", "test"]:
            raise AttributeError("Possible values for split are 'train' or 'test'")
        data_path = str(data_path)
        if download:
            try:
                from torchvision.datasets import MNIST, SVHN

                MNIST(data_path, train=(split == "train"), download=True)
                SVHN(data_path, split=split, download=True)
            except ImportError as e:
                raise AttributeError("download=True needs torchvision; place the MNIST / SVHN files under data_path") from e
        mnist_x, mnist_y = load_mnist(data_path, split == "train")
        svhn_x, svhn_y = load_svhn(data_path, split)
        self.data_mul = data_multiplication
        self.path_to_idx = data_path + f"/mnist_svhn_idx_data_mul_{self.data_mul}/" + split
        if not self._check_pairing_exists():
            self.create_pairing(mnist_y, svhn_y)
        i_mnist = torch.load(f"{self.path_to_idx}/mnist_idx.pt", weights_only=True)
        i_svhn = torch.load(f"{self.path_to_idx}/svhn_idx.pt", weights_only=True)
        order = np.arange(len(i_mnist))
        np.random.shuffle(order)  # so that the samples are not ordered by label (mnist_svhn.py:65-68: the global numpy RNG)
        order = torch.from_numpy(order)
        labels = mnist_y[i_mnist][order]
        data_mnist = mnist_x.float().div(255).unsqueeze(1)
        data_svhn = svhn_x.float().div(255)
        data = dict(mnist=ResampleDataset(data_mnist, i_mnist[order]), svhn=ResampleDataset(data_svhn, i_svhn[order]))
        self.data_path = data_path
        super().__init__(data, labels)

    def _check_pairing_exists(self):
        for f in ("mnist_idx.pt", "svhn_idx.pt"):
            if not os.path.exists(f"{self.path_to_idx}/{f}"):
                logger.warning("Pairing not found.")
                return False
        return True

    def rand_match_on_idx(self, l1, idx1, l2, idx2, max_d=10000):
        """Pair the two sample sets digit by digit (reference `mnist_svhn.py:100-115`; pinned against the reference's own
        output by tests/golden/pairing_mnist_svhn.npz).  `l1 / l2` are the sorted labels, `idx1 / idx2` the matching
        sample positions.  For every digit the first `min(count_1, count_2, max_d)` candidates of each set are kept and
        shuffled `data_mul` times; the global torch generator is consumed in the order (digit, round, set 1, set 2), which is
        what makes the index files of the reference reproducible."""
        picked = ([], [])
        for digit in torch.unique(l1):  # sorted; both sets hold every digit
            cand = (idx1[l1 == digit], idx2[l2 == digit])
            keep = min(cand[0].numel(), cand[1].numel(), max_d)
            for _round in range(self.data_mul):
                for side in (0, 1):
                    picked[side].append(cand[side][:keep][torch.randperm(keep)])
        return torch.cat(picked[0]), torch.cat(picked[1])

    def create_pairing(self, mnist_labels, svhn_labels, max_d=10000):
        logger.info(f"Creating indices in {self.path_to_idx}")
        svhn_labels = svhn_labels % 10  # SVHN stores the digit 0 as class 10
        mnist_l, mnist_li = mnist_labels.sort()
        svhn_l, svhn_li = svhn_labels.sort()
        idx1, idx2 = self.rand_match_on_idx(mnist_l, mnist_li, svhn_l, svhn_li, max_d=max_d)
        Path(self.path_to_idx).mkdir(parents=True, exist_ok=True)
        torch.save(idx1, f"{self.path_to_idx}/mnist_idx.pt")
        torch.save(idx2, f"{self.path_to_idx}/svhn_idx.pt")
