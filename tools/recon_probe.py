"""Isolated timing of the fused reconstruction-NLL kernel against plain device copies of the same byte count
(run on the GPU box):  python tools/recon_probe.py
Separates fixed launch / ramp costs from streaming efficiency: the kernel is timed back-to-back (HIP events around 50
launches) at the bench shape (K=10, B=512, MnistSvhn) and at 4x the batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multivae_amd import kernels
from multivae_amd._lib import DIST


def timeit(fn, n=20, reps=5):
    """GPU time per call: n calls captured in one hipGraph (the Python / ctypes launch cost of these small kernels is
    larger than their run time, so eager back-to-back timing would measure the host)."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3  # us


def main():
    d = torch.device("cuda:0")
    K = 10
    for B in (512, 2048):
        recs = [torch.randn(K, B, 784, device=d), torch.randn(K, B, 3072, device=d)]
        xs = [torch.rand(B, 784, device=d), torch.rand(B, 3072, device=d)]
        spec = dict(K=K, B=B, x=xs, masks=[None, None], dist=[DIST["normal"]] * 2, scale=[1.0, 1.0], rescale=[3.9, 1.0],
                    coef=[1.0 / (K * B)] * 2, lossw=[1.0, 1.0], extra_coef=[], extra_lossw=[], loss_sum_scale=float(B))
        rr = [r.requires_grad_() for r in recs]

        def recon():
            kernels.ReconLossFn.apply(spec, 2, *rr)

        def assembly_only():
            kernels.ReconLossFn.apply(dict(spec, x=[], masks=[], dist=[], scale=[], rescale=[], coef=[], lossw=[],
                                           extra_coef=[1.0], extra_lossw=[1.0]), 0, xs[0])

        bytes_alg = 4 * (2 * K * B * 3856 + B * 3856)
        t = timeit(recon) - 8.0  # minus the 1-block scalar assembly kernel (~8 us)
        src = torch.empty(K * B * 3856, device=d)
        dst = torch.empty_like(src)
        tc = timeit(lambda: dst.copy_(src))
        rows_only = [r.detach() for r in recs]

        def recon_rows():
            kernels.recon_nll_rows(rows_only, xs, [0, 0], [1.0, 1.0], K, B)

        tr = timeit(recon_rows)
        print(f"B={B}: recon fwd+grad {t:.1f} us ({bytes_alg / t / 1e6:.2f} TB/s algorithmic); "
              f"rows only (read {4 * (K + 1) * B * 3856 / 1e6:.0f} MB) {tr:.1f} us = {4 * (K + 1) * B * 3856 / tr / 1e6:.2f} TB/s; "
              f"copy of {src.numel() * 4 / 1e6:.0f} MB {tc:.1f} us = {2 * src.numel() * 4 / tc / 1e6:.2f} TB/s")


if __name__ == "__main__":
    main()
