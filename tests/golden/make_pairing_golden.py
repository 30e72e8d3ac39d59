"""Fixture for the MnistSvhn pairing (pin `rand_match_on_idx` against the REFERENCE's own output).

Build container only (needs /root/reference):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_pairing_golden.py

Calls the reference's `MnistSvhn.rand_match_on_idx` (multivae/data/datasets/mnist_svhn.py:100-115) on procedural label vectors
under a fixed torch seed and stores inputs' parameters + the index vectors it returns.  Arrays and JSON only."""
import sys

sys.dont_write_bytecode = True
import json
import os
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import torch

import _reference_import as R

R.install()
import procedural as P
from multivae.data.datasets.mnist_svhn import MnistSvhn


def labels(n, seed):
    return torch.from_numpy((P.hash_uniform(n, seed) * 10).astype(np.int64) % 10)


def main():
    out = {}
    cases = [dict(name="a", n1=230, n2=310, data_mul=3, max_d=10000, seed=11), dict(name="b", n1=400, n2=180, data_mul=1, max_d=12, seed=12)]
    for c in cases:
        l1, i1 = labels(c["n1"], c["seed"]).sort()
        l2, i2 = labels(c["n2"], c["seed"] + 100).sort()
        torch.manual_seed(c["seed"])
        r1, r2 = MnistSvhn.rand_match_on_idx(types.SimpleNamespace(data_mul=c["data_mul"]), l1, i1, l2, i2, max_d=c["max_d"])
        out[c["name"] + "/idx1"], out[c["name"] + "/idx2"] = r1.numpy(), r2.numpy()
        print(c["name"], r1.shape, r2.shape)
    out["cfg_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "pairing_mnist_svhn.npz"), **out)


if __name__ == "__main__":
    main()
