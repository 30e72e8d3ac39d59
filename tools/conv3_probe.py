"""Isolated timings of the 3x3 / stride-1 convolution launches of the ResNet configurations (cfg4 / cfg5 layer shapes):
register-stationary kernels (conv3rs.hip) vs the tiled implicit-GEMM engine (debug flag 0x400), forward form and
weight gradient.  Run on the GPU box:  python tools/conv3_probe.py [cfg5|cfg4]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multivae_amd import _lib
from multivae_amd import kernels as K

d = torch.device("cuda:0")
lib = _lib.load()
lib.mvk_debug_set_flags.argtypes = [ctypes.c_int]
which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
if which == "prof":  # a few launches per kernel NAME (one shape each) for the counter passes of tools/conv3_pmc.sh
    # kernel -> GFLOP per launch: tools/make_profiles.py C3_GFLOP
    for n_, H_, Ci_, Co_, wg_ in [(128, 64, 64, 64, True), (128, 32, 64, 128, False), (128, 16, 128, 128, False),
                                  (128, 16, 128, 256, False), (1600, 14, 128, 64, False)]:
        x = torch.randn(n_, H_, H_, Ci_, device=d)
        w = torch.randn(Co_, Ci_, 3, 3, device=d) / (3 * Ci_ ** 0.5)
        b = torch.randn(Co_, device=d)
        src = torch.randn(n_, H_, H_, Co_, device=d)
        (wf, wb), = K.pack_weights([(w, "c3", True, True)])
        wparam = w.clone().requires_grad_(True)
        wparam.grad = torch.zeros_like(wparam)
        for _ in range(5):
            K.conv3x3(x, wf, b, n_, H_, H_, Ci_, Co_, act=K.LEAKY, y_act_src=src, y_src_act=K.LEAKY)
            if wg_:
                K.conv3x3_wgrad(x, src, wparam, n_, H_, H_, Ci_, Co_)
        torch.cuda.synchronize()
    sys.exit(0)
if which == "one":  # the 64 -> 64 channel layers only (variant builds of tools/conv3_variants.sh)
    n = 128
    shapes = [(64, 64, 64), (32, 64, 64)]
elif which == "cfg5":
    n = 128
    shapes = [(64, 64, 64), (32, 64, 64), (32, 64, 128), (32, 128, 64), (16, 128, 128), (16, 128, 256), (16, 256, 128)]
else:
    n = 1600
    shapes = [(28, 64, 64), (14, 64, 64), (14, 64, 128), (14, 128, 64), (7, 128, 128), (7, 128, 256), (7, 256, 128)]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{which}: n = {n}")
print("| H | Cin | Cout | GFLOP | tiled us | TF/s | register-stationary us | TF/s | wgrad tiled us | wgrad new us |")
for H, Cin, Cout in shapes:
    x = torch.randn(n, H, H, Cin, device=d)
    w = torch.randn(Cout, Cin, 3, 3, device=d) / (3 * Cin ** 0.5)
    b = torch.randn(Cout, device=d)
    src = torch.randn(n, H, H, Cout, device=d)
    dy = torch.randn(n, H, H, Cout, device=d)
    (wf, wb), = K.pack_weights([(w, "c3", True, True)])
    wparam = w.clone().requires_grad_(True)
    wparam.grad = torch.zeros_like(wparam)
    gf = 2.0 * n * H * H * 9 * Cin * Cout / 1e9
    res = []
    for flag in (0x400, 0):
        lib.mvk_debug_set_flags(flag)
        t = timeit(lambda: K.conv3x3(x, wf, b, n, H, H, Cin, Cout, act=K.LEAKY, y_act_src=src, y_src_act=K.LEAKY))
        tw = timeit(lambda: K.conv3x3_wgrad(x, dy, wparam, n, H, H, Cin, Cout))
        res.append((t, tw))
    lib.mvk_debug_set_flags(0)
    print(f"| {H} | {Cin} | {Cout} | {gf:.1f} | {res[0][0]:.0f} | {gf / res[0][0] * 1e3:.0f} | {res[1][0]:.0f} | {gf / res[1][0] * 1e3:.0f} | "
          f"{res[0][1]:.0f} | {res[1][1]:.0f} |")
