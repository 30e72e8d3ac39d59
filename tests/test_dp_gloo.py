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
        ret[rank] = dict(flat=flat.grad.clone(), params=flat.flat.clone(), idx=idx, eps=eps, sd=sd_np)
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
