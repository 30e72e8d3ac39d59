"""The dataset loaders of the reference's examples on synthetic files in the on-disk formats they read:
`MMNISTDataset` (`/root/reference/src/multivae/data/datasets/mmnist.py:22-172`: PolyMNIST `.pt` files, `missing_ratio` masks)
and `MnistSvhn` (`mnist_svhn.py:19-136`: torchvision's MNIST idx files + SVHN `.mat`, the pairing files)."""
import math
import os

import numpy as np
import pytest
import torch


def make_mmnist(root, split, n):
    d = os.path.join(root, "MMNIST", split)
    os.makedirs(d)
    g = torch.Generator().manual_seed(7)
    for i in range(5):
        torch.save(torch.rand(n, 3, 28, 28, generator=g) + 0.01, os.path.join(d, f"m{i}.pt"))
    torch.save(torch.randint(0, 10, (n,), generator=g), os.path.join(d, "labels.pt"))


def make_mnist_svhn(root, n_mnist=200, n_svhn=260):
    raw = os.path.join(root, "MNIST", "raw")
    os.makedirs(raw)
    rng = np.random.RandomState(3)
    for pre, n in (("train", n_mnist), ("t10k", 80)):
        x = rng.randint(0, 256, (n, 28, 28)).astype(np.uint8)
        y = (np.arange(n) % 10).astype(np.uint8)
        with open(os.path.join(raw, f"{pre}-images-idx3-ubyte"), "wb") as f:
            f.write((0x00000803).to_bytes(4, "big") + n.to_bytes(4, "big") + (28).to_bytes(4, "big") + (28).to_bytes(4, "big"))
            f.write(x.tobytes())
        with open(os.path.join(raw, f"{pre}-labels-idx1-ubyte"), "wb") as f:
            f.write((0x00000801).to_bytes(4, "big") + n.to_bytes(4, "big"))
            f.write(y.tobytes())
    from scipy.io import savemat

    for split, n in (("train", n_svhn), ("test", 90)):
        X = rng.randint(0, 256, (32, 32, 3, n)).astype(np.uint8)  # the .mat layout: H, W, C, n
        y = ((np.arange(n) % 10) + 1).astype(np.uint8).reshape(-1, 1)  # SVHN classes 1..10 (10 = digit 0)
        savemat(os.path.join(root, f"{split}_32x32.mat"), dict(X=X, y=y))


def test_mmnist_dataset_masks_and_length(tmp_path):
    from multivae_amd.data.datasets import MMNISTDataset

    n = 50
    make_mmnist(str(tmp_path), "train", n)
    full = MMNISTDataset(str(tmp_path), split="train")
    assert len(full) == n and set(full.data) == {f"m{i}" for i in range(5)} and not hasattr(full[0], "masks")
    assert full[3].data["m2"].shape == (3, 28, 28) and int(full[3].labels) == int(full.labels[3])
    r = 0.4
    inc = MMNISTDataset(str(tmp_path), split="train", missing_ratio=r)
    for i in range(5):  # the reference's masks: mmnist.py:111-121
        want = torch.bernoulli(torch.ones(n) * (1 - r), generator=torch.Generator().manual_seed(i)).bool()
        if i == 0:
            want = torch.ones(n).bool()
        assert torch.equal(inc.masks[f"m{i}"], want)
        gone = ~want
        assert float(inc.data[f"m{i}"][gone].abs().sum()) == 0.0  # erased content
        assert torch.equal(inc.data[f"m{i}"][want], full.data[f"m{i}"][want])
    item = inc[7]
    assert set(item.masks) == set(item.data) and item.masks["m0"]
    short = MMNISTDataset(str(tmp_path), split="train", missing_ratio=r, keep_incomplete=False)
    assert len(short) == math.ceil((1 - r) ** 4 * n) and not hasattr(short[0], "masks")
    with pytest.raises(AttributeError):
        MMNISTDataset(str(tmp_path / "nowhere"), split="train")


def test_mnist_svhn_pairing_and_items(tmp_path):
    from multivae_amd.data.datasets import MnistSvhn
    from multivae_amd.data.datasets.mnist_svhn import load_mnist, load_svhn

    root = str(tmp_path)
    make_mnist_svhn(root)
    np.random.seed(0)
    torch.manual_seed(0)
    ds = MnistSvhn(root, split="train", data_multiplication=3)
    # 20 MNIST and 26 SVHN samples per class -> min = 20 pairs per class and pairing round
    assert len(ds) == 10 * 20 * 3
    assert os.path.exists(os.path.join(root, "mnist_svhn_idx_data_mul_3", "train", "mnist_idx.pt"))
    mx, my = load_mnist(root, True)
    sx, sy = load_svhn(root, "train")
    for i in (0, 17, 311, 599):
        it = ds[i]
        assert it.data["mnist"].shape == (1, 28, 28) and it.data["svhn"].shape == (3, 32, 32)
        im, isv = int(ds.data["mnist"].index[i]), int(ds.data["svhn"].index[i])
        assert int(my[im]) == int(sy[isv]) % 10 == int(it.labels)  # same digit in both modalities
        assert torch.equal(it.data["mnist"], mx[im].float().div(255).unsqueeze(0))
        assert torch.equal(it.data["svhn"], sx[isv].float().div(255))
    # an existing pairing is reused (the reference's files load unchanged)
    i1 = torch.load(os.path.join(root, "mnist_svhn_idx_data_mul_3", "train", "mnist_idx.pt"), weights_only=True)
    ds2 = MnistSvhn(root, split="train", data_multiplication=3)
    assert sorted(ds2.data["mnist"].index.tolist()) == sorted(i1.tolist())
    with pytest.raises(AttributeError):
        MnistSvhn(root, split="valid")


def test_mnist_svhn_pairing_matches_reference_fixture():
    """`rand_match_on_idx` against index vectors produced by the REFERENCE's own function under the same torch seed
    (tests/golden/make_pairing_golden.py; mnist_svhn.py:100-115): same pairs in the same order."""
    import json
    import types

    import golden_cases as G
    from multivae_amd.data.datasets import MnistSvhn

    z = np.load(os.path.join(G.GOLDEN, "pairing_mnist_svhn.npz"))
    for c in json.loads(bytes(z["cfg_json"]).decode()):
        lab = lambda n, seed: torch.from_numpy((G.P.hash_uniform(n, seed) * 10).astype(np.int64) % 10)
        l1, i1 = lab(c["n1"], c["seed"]).sort()
        l2, i2 = lab(c["n2"], c["seed"] + 100).sort()
        torch.manual_seed(c["seed"])
        r1, r2 = MnistSvhn.rand_match_on_idx(types.SimpleNamespace(data_mul=c["data_mul"]), l1, i1, l2, i2, max_d=c["max_d"])
        assert np.array_equal(r1.numpy(), z[c["name"] + "/idx1"]) and np.array_equal(r2.numpy(), z[c["name"] + "/idx2"]), c


@pytest.mark.gpu
def test_trainer_on_the_example_datasets(tmp_path):
    """One epoch of BaseTrainer on both loaders through the device-resident batch iterator: MMVAE+ on PolyMNIST-shaped data
    with 40 % missing modalities (masks reach the model), MoPoE on the paired MnistSvhn set (base + index gathered on the
    GPU); the batches equal the datasets' own items."""
    from multivae_amd.data.datasets import MMNISTDataset, MnistSvhn
    from multivae_amd.models import MoPoE, MoPoEConfig, MVTCAE, MVTCAEConfig
    from multivae_amd.trainers import BaseTrainer, BaseTrainerConfig

    root = str(tmp_path)
    make_mmnist(root, "train", 96)
    make_mnist_svhn(root)
    inc = MMNISTDataset(root, split="train", missing_ratio=0.4)
    model = MVTCAE(MVTCAEConfig(n_modalities=5, latent_dim=8, input_dims={f"m{i}": (3, 28, 28) for i in range(5)}))
    tr = BaseTrainer(model, inc, training_config=BaseTrainerConfig(output_dir=root, per_device_train_batch_size=32,
                                                                   num_epochs=1, learning_rate=1e-3))
    assert tr.train_loader.fast
    batch = next(iter(tr.train_loader))
    assert hasattr(batch, "masks") and batch.masks["m3"].dtype == torch.bool and batch.data["m1"].is_cuda
    hist = tr.train()
    assert np.isfinite(hist[0]["train_epoch_loss"])
    np.random.seed(1)
    torch.manual_seed(1)
    pair = MnistSvhn(root, split="train", data_multiplication=2)
    mp = MoPoE(MoPoEConfig(n_modalities=2, latent_dim=6, input_dims=dict(mnist=(1, 28, 28), svhn=(3, 32, 32))))
    tr2 = BaseTrainer(mp, pair, training_config=BaseTrainerConfig(output_dir=root, per_device_train_batch_size=64,
                                                                  num_epochs=1, learning_rate=1e-3))
    assert tr2.train_loader.fast and set(tr2.train_loader.resample) == {"mnist", "svhn"}
    tr2.train_loader.shuffle = False
    b0 = next(iter(tr2.train_loader))
    for i in (0, 5, 63):
        assert torch.equal(b0.data["svhn"][i].cpu(), pair[i].data["svhn"])
        assert torch.equal(b0.data["mnist"][i].cpu(), pair[i].data["mnist"])
    tr2.train_loader.shuffle = True
    assert np.isfinite(tr2.train()[0]["train_epoch_loss"])
