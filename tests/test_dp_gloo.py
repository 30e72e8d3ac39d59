"""Data-parallel path with world_size 2 over gloo on the CPU: the flat gradient buffer is
exchanged with ONE all-reduce(SUM), the 1/world_size average is applied afterwards (DDP semantics), and the
result equals the mean over shards of the oracle's per-shard gradients (MoPoE's row-range subset assignment is
per local shard, so the DP result is NOT the big-batch gradient).  The kernels themselves need a GPU; here the
per-shard gradients come from the CPU oracle and are written into the flat buffer's views, which is exactly
where the HIP backward accumulates them."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_cases as G
from oracle import elbo, nets


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_grads(sd_np, data, eps, dims, L):
    sd = {k: G.t(v).clone().requires_grad_(True) for k, v in sd_np.items()}
    enc_f, dec_f = nets.build_default_mlp(sd, dims)
    names = list(dims)
    e = {m: enc_f[m](data[m]) for m in names}
    o = elbo.mopoe_forward(e, data, dec_f, eps, names=names, beta=1.0)
    o["loss"].backward()
    return {k: v.grad for k, v in sd.items()}, float(o["loss"])


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", init_method="env://", world_size=world, rank=rank)
    try:
        from multivae_amd.models import MoPoE, MoPoEConfig
        from multivae_amd.trainers import FlatParams
        from multivae_amd.trainers.base import shard_indices

        dims, L, n = G.TINY_DIMS, 5, 24
        torch.manual_seed(1234 + rank)  # ranks start from DIFFERENT weights: the broadcast must fix that
        model = MoPoE(MoPoEConfig(n_modalities=4, latent_dim=L, input_dims=dict(dims)))
        flat = FlatParams(model)
        flat.broadcast(0)
        full = {m: G.t(G.P.uniform((n,) + d, 900 + i)) for i, (m, d) in enumerate(dims.items())}
        idx = shard_indices(n, world, rank)
        shard = {m: v[idx] for m, v in full.items()}
        eps = torch.randn(len(idx), L, generator=torch.Generator().manual_seed(1000 + rank))
        sd_np = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
        grads, loss = _shard_grads(sd_np, shard, eps, dims, L)
        flat.zero_grad()
        for k, p in model.named_parameters():
            p.grad.add_(grads[k])  # what the HIP backward does: accumulate into the flat-buffer views
        flat.all_reduce()
        flat.grad.mul_(1.0 / world)
        ret[rank] = dict(flat=flat.dense(flat.grad).clone(), params=flat.dense(flat.flat).clone(), idx=idx, eps=eps, sd=sd_np)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_allreduce_matches_mean_of_shard_gradients():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["params"], r1["params"]), "parameter broadcast from rank 0"
    assert torch.equal(r0["flat"], r1["flat"]), "every rank holds the same averaged gradient"
    assert sorted(r0["idx"].tolist() + r1["idx"].tolist()) == list(range(24))
    # reference: mean over shards of the oracle's per-shard gradients, parameter order of model.parameters()
    dims, L, n = G.TINY_DIMS, 5, 24
    full = {m: G.t(G.P.uniform((n,) + d, 900 + i)) for i, (m, d) in enumerate(dims.items())}
    acc = None
    for r in (r0, r1):
        shard = {m: v[r["idx"]] for m, v in full.items()}
        g, _ = _shard_grads(r0["sd"], shard, r["eps"], dims, L)
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    ref = torch.cat([(acc[k] / world).reshape(-1) for k in r0["sd"].keys()])
    assert ref.numel() == r0["flat"].numel()
    assert torch.allclose(r0["flat"], ref, rtol=1e-6, atol=1e-8)


# ------------------------------------------------------------------------------------------------------------------------
# BaseTrainer itself under world_size 2 (the torch.optim path: schedulers of other optimizer classes, use_fused_adam=False)
# ------------------------------------------------------------------------------------------------------------------------
class _PlainTorchModel:
    """Built lazily (multivae_amd imports) — a plugin model written in plain PyTorch: loss = sum-of-squares ELBO stand-in."""

    @staticmethod
    def make(seed):
        from multivae_amd._output import ModelOutput
        from multivae_amd.models.base.base_config import BaseMultiVAEConfig
        from multivae_amd.models.base.base_model import BaseModel

        class Plain(BaseModel):
            def __init__(self):
                super().__init__(BaseMultiVAEConfig(n_modalities=2))
                self.model_name = "Plain"
                g = torch.Generator().manual_seed(seed)
                self.w = torch.nn.Parameter(torch.randn(3, 4, generator=g))
                self.b = torch.nn.Parameter(torch.randn(3, generator=g))
                self.frozen = torch.nn.Parameter(torch.ones(2), requires_grad=False)

            def forward(self, inputs, **kwargs):
                x, y = inputs.data["a"], inputs.data["b"]
                r = torch.nn.functional.linear(x, self.w, self.b) - y
                s = (r * r).sum()
                return ModelOutput(loss=s / x.shape[0], loss_sum=s, metrics=dict(rows=torch.tensor(float(x.shape[0]))))

        return Plain()


def _dataset(n):
    from multivae_amd.data.datasets.base import MultimodalBaseDataset

    g = torch.Generator().manual_seed(77)
    return MultimodalBaseDataset(dict(a=torch.randn(n, 4, generator=g), b=torch.randn(n, 3, generator=g)))


def _trainer_worker(rank, world, port, ret, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MVK_TRAINER_ALLOW_CPU="1")
    from multivae_amd.trainers import BaseTrainer, BaseTrainerConfig

    model = _PlainTorchModel.make(seed=100 + rank)  # different weights per rank: the trainer broadcasts rank 0's
    cfg = BaseTrainerConfig(output_dir=outdir, per_device_train_batch_size=5, num_epochs=2, learning_rate=1e-2,
                            optimizer_cls="Adam", use_fused_adam=False, scheduler_cls="StepLR",
                            scheduler_params=dict(step_size=1, gamma=0.5), no_cuda=True, dist_backend="gloo",
                            world_size=world, rank=rank, local_rank=rank)
    tr = BaseTrainer(model, _dataset(23), training_config=cfg)
    hist = tr.train()
    ret[rank] = dict(w=model.w.detach().clone(), b=model.b.detach().clone(),
                     loss=[h["train_epoch_loss"] for h in hist])


@pytest.mark.timeout(300)
def test_base_trainer_distributed_torch_optim_path(tmp_path):
    """The distributed train_step of BaseTrainer with a torch.optim optimizer + lr scheduler (no fused Adam): gradients
    stay views of the flat buffer, ONE all-reduce per step, averaged; ranks end with identical parameters, equal to a
    single-process replay of the same schedule; the epoch loss is the local sum over the FULL dataset length (:748)."""
    from multivae_amd.trainers.base import shard_indices

    world, n, bs = 2, 23, 5
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_trainer_worker, args=(world, port, ret, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["w"], r1["w"]) and torch.equal(r0["b"], r1["b"]), "ranks diverged"
    # single-process replay: rank 0's initial weights, per step the MEAN over ranks of the local mean-loss gradients
    model = _PlainTorchModel.make(seed=100)
    ds = _dataset(n)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    from multivae_amd.data.datasets.base import DatasetOutput

    losses = {0: [], 1: []}
    for epoch in range(2):
        idx = [shard_indices(n, world, r) for r in range(world)]
        nb = (len(idx[0]) + bs - 1) // bs
        ep = {0: 0.0, 1: 0.0}
        for b in range(nb):
            grads = None
            for r in range(world):
                sel = idx[r][b * bs:(b + 1) * bs]
                out = model(DatasetOutput(data={k: v[sel] for k, v in ds.data.items()}))
                g = torch.autograd.grad(out.loss, [model.w, model.b])
                ep[r] += float(out.loss_sum)
                grads = g if grads is None else [a + c for a, c in zip(grads, g)]
            model.w.grad, model.b.grad = grads[0] / world, grads[1] / world
            opt.step()
        sched.step()
        for r in range(world):
            losses[r].append(ep[r] / n)
    assert torch.allclose(r0["w"], model.w.detach(), rtol=1e-5, atol=1e-7)
    assert torch.allclose(r0["b"], model.b.detach(), rtol=1e-5, atol=1e-7)
    for r, got in ((0, r0["loss"]), (1, r1["loss"])):
        assert got == pytest.approx(losses[r], rel=1e-5)
