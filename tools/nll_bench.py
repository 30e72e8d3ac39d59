"""Measure the importance-sampled joint-likelihood evaluation (compute_joint_nll) on one MI355X:
MoPoE on MnistSvhn-shaped data, K = 1000 importance samples per data point.  The CPU side is the oracle's restatement
of the reference's per-data-point / per-K-chunk loop (oracle.elbo.mopoe_joint_nll) on a bounded sample of points.

    python tools/nll_bench.py [--points 512] [--K 1000] [--reps 5] [--cpu-points 4]
prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=512)
    ap.add_argument("--K", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-points", type=int, default=4)
    ap.add_argument("--rows-budget", type=int, default=0)
    a = ap.parse_args()
    from multivae_amd import kernels
    from multivae_amd.data.datasets.base import DatasetOutput

    if a.rows_budget:
        kernels.IWAE_ROWS_BUDGET = a.rows_budget
    dev = torch.device("cuda:0")
    L = 20
    model = bench.build_model(1, L, dev)
    data = bench.synthetic_batch(a.points, dev)
    inputs = DatasetOutput(data=data)
    noise = torch.randn(a.K, a.points, L, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    nll = model.compute_joint_nll(inputs, K=a.K, noise=noise)  # warm-up
    torch.cuda.synchronize()
    times = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        nll = model.compute_joint_nll(inputs, K=a.K, noise=noise)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    times.sort()
    t_gpu = times[len(times) // 2]
    out = dict(metric="joint_nll_points_per_s", value=a.points / t_gpu, unit="data points/s", K=a.K, points=a.points,
               s_per_call=t_gpu, decoder_rows_per_s=a.points * a.K / t_gpu, rows_budget=kernels.IWAE_ROWS_BUDGET,
               nll_per_point=float(nll) / a.points)
    if a.cpu_points:
        from oracle import elbo, nets

        torch.set_num_threads(int(os.environ.get("MVK_CPU_THREADS", "16")))
        n = a.cpu_points
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        enc_f, dec_f = nets.build_mnist_svhn(sd, L)
        cdata = {m: v[:n].cpu() for m, v in data.items()}
        with torch.no_grad():
            t0 = time.perf_counter()
            e = {m: enc_f[m](cdata[m]) for m in ("mnist", "svhn")}
            o = elbo.mopoe_joint_nll(e, cdata, dec_f, noise[:, :n].cpu(), names=["mnist", "svhn"], batch_size_K=100)
            t_cpu = time.perf_counter() - t0
        # same estimator on the same points and noise (row-range subset selection depends on the batch size, so compare
        # per-point values of a GPU call on the same n points)
        sub = DatasetOutput(data={m: v[:n] for m, v in data.items()})
        g = model.compute_joint_nll(sub, K=a.K, noise=noise[:, :n].contiguous())
        out["cpu_baseline"] = dict(value=n / t_cpu, unit="data points/s", cores=torch.get_num_threads(), kind="port",
                                   sample=f"{n} data points x K={a.K} (oracle.elbo.mopoe_joint_nll, chunks of 100)",
                                   rel_diff_vs_gpu=abs(float(o[0]) - float(g)) / abs(float(o[0])))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
