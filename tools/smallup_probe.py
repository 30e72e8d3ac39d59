"""Standalone launches of the image-layer kernels (csrc/smallconv.hip: ConvTranspose2d(32, 3, 4, 2, 1) + Sigmoid and its
backward) at the headline batch n = K * B = 5120, for timing and rocprofv3 --pmc passes.
Usage: python tools/smallup_probe.py [n] [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from multivae_amd import _lib, kernels as K  # noqa: E402
from multivae_amd._lib import call, ptr, stream_ptr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
Cu, Cv, h = 3, 32, 16
V = torch.randn(n, h, h, Cv, generator=g).relu().to(d)
W = (torch.randn(Cv, Cu, 4, 4, generator=g) / 16).to(d)
b = torch.randn(Cu, generator=g).to(d)
U = torch.empty(n, Cu, 2 * h, 2 * h, device=d)
dU = torch.randn(n, Cu, 2 * h, 2 * h, generator=g).to(d)
dV = torch.empty_like(V)
dW, db, dbv = torch.zeros_like(W), torch.zeros_like(b), torch.zeros(Cv, device=d)
ws = K._ws(V)


def fwd():
    call("mvk_conv4s2_small_up_fwd", ptr(V), ptr(W), ptr(b), ptr(U), n, h, h, Cu, Cv, K.SIGMOID, stream_ptr())


def bwd():
    call("mvk_conv4s2_small_up_bwd", ptr(dU), ptr(U), K.SIGMOID, ptr(V), K.RELU, ptr(W), ptr(dV), ptr(dW), ptr(db), ptr(dbv), ptr(ws),
         ws.numel(), n, h, h, Cu, Cv, stream_ptr())


amax = torch.zeros(1, device=d)
K.amax_of(V, amax)
X = torch.rand(n // 10, Cu, 2 * h, 2 * h, generator=g).to(d)
rows = torch.empty(n, device=d)


def fwd_s():  # scaled fp16 pairs (small_up_fwd_h_kernel)
    call("mvk_conv4s2_small_up_fwd_s", ptr(V), ptr(W), ptr(b), ptr(U), n, h, h, Cu, Cv, K.SIGMOID, ptr(amax), stream_ptr())


def nll():  # fused tail, bf16 pieces
    call("mvk_conv4s2_small_up_fwd_nll_w", ptr(V), ptr(W), ptr(b), ptr(X), X.shape[0], 0.75, 1.0, ptr(U), ptr(rows), n, h, h, Cu, Cv,
         K.SIGMOID, stream_ptr())


def nll_s():  # fused tail, scaled fp16 pairs
    call("mvk_conv4s2_small_up_fwd_nll_s", ptr(V), ptr(W), ptr(b), ptr(X), X.shape[0], 0.75, 1.0, ptr(U), ptr(rows), n, h, h, Cu, Cv,
         K.SIGMOID, ptr(amax), stream_ptr())


mb_f = 4e-6 * n * h * h * (Cv + 4 * Cu)
mb_n = mb_f + 4e-6 * X.numel()
for name, f, mb in (("fwd", fwd, mb_f), ("fwd_s", fwd_s, mb_f), ("nll", nll, mb_n), ("nll_s", nll_s, mb_n),
                    ("bwd", bwd, 4e-6 * n * h * h * (2 * Cv + 8 * Cu))):
    f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record()
        f()
        e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
    print(f"{name}: median {t[len(t) // 2]:.1f} us (min {t[0]:.1f}), {mb:.1f} MB -> {mb / t[len(t) // 2]:.2f} TB/s "
          f"= {mb / t[len(t) // 2] / 8e3 * 1e3:.3f} of 8 TB/s")
