"""Per-wave cycle split of small_up_bwd_kernel (SUPROF build, run ON the GPU box via tools/smallup_phase.sh)."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from multivae_amd import _lib, kernels as K  # noqa: E402
from multivae_amd._lib import call, ptr, stream_ptr  # noqa: E402

lib = _lib.load()
lib.mvk_smallup_debug_buffer.argtypes = [ctypes.c_void_p]
d = torch.device("cuda:0")
n, Cu, Cv, h = 5120, 3, 32, 16
g = torch.Generator().manual_seed(1)
V = torch.randn(n, h, h, Cv, generator=g).relu().to(d)
W = (torch.randn(Cv, Cu, 4, 4, generator=g) / 16).to(d)
U = torch.rand(n, Cu, 2 * h, 2 * h, generator=g).to(d)
dU = torch.randn(n, Cu, 2 * h, 2 * h, generator=g).to(d)
dV = torch.empty_like(V)
dW, db, dbv = torch.zeros_like(W), torch.zeros(Cu, device=d), torch.zeros(Cv, device=d)
ws = K._ws(V)
buf = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=d)
lib.mvk_smallup_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
for _ in range(2):
    buf.zero_()
    call("mvk_conv4s2_small_up_bwd", ptr(dU), ptr(U), K.SIGMOID, ptr(V), K.RELU, ptr(W), ptr(dV), ptr(dW), ptr(db), ptr(dbv), ptr(ws),
         ws.numel(), n, h, h, Cu, Cv, stream_ptr())
    torch.cuda.synchronize()
t = buf.view(512, 4, 8).double().cpu()
tot = t[..., 0]
names = ["barrier waits", "staging (registers -> LDS)", "prefetch issue", "backward-data MFMA loop", "epilogue (mask, stores)", "weight-gradient MFMA loop"]
print(f"cycles per wave: mean {tot.mean():.0f}  (10 images per workgroup)")
for i, nm in enumerate(names):
    print(f"  {nm:32s} {100 * (t[..., 1 + i] / tot).mean():5.1f} %")
print(f"  outside the loop                 {100 * ((tot - t[..., 1:7].sum(-1)) / tot).mean():5.1f} %")
