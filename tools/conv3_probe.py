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
        pool = K.AmaxPool(x, 16)
        xam, sam = K.amax_of(x, pool.take()), K.amax_of(src, pool.take())
        for _ in range(5):
            K.conv3x3(x, wf, b, n_, H_, H_, Ci_, Co_, act=K.LEAKY, y_act_src=src, y_src_act=K.LEAKY)
            K.conv3x3_s(x, wf, b, n_, H_, H_, Ci_, Co_, xam, wf.mvk_amax, pool.take(), act=K.LEAKY, y_act_src=src, y_src_act=K.LEAKY)
            if wg_:
                K.conv3x3_wgrad(x, src, wparam, n_, H_, H_, Ci_, Co_)
                K.conv3x3_wgrad_s(x, src, wparam, None, n_, H_, H_, Ci_, Co_, xam, sam)
        torch.cuda.synchronize()
    sys.exit(0)
if which in ("f16", "f16cfg4", "f16one"):  # bf16-piece kernels vs the scaled-fp16 form (mvk_conv3x3_s): time, error against float64
    n = 1600 if which == "f16cfg4" else 128
    shapes = ([(64, 64, 64), (32, 64, 64), (32, 64, 128), (32, 128, 64), (16, 128, 128), (16, 128, 256)] if which == "f16"
              else [(64, 64, 64), (63, 64, 64)] if which == "f16one"  # variant builds (tools/conv3_variants.sh): 64 -> 64 only
              else [(28, 64, 64), (14, 64, 64), (14, 64, 128), (14, 128, 64), (7, 128, 128), (7, 128, 256)])

    def t_us(fn, reps=10):
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

    print(f"{which}: n = {n}; masked backward-data form (mask + column sums); err = max |y - f64| / max |f64| on 2 images")
    print("| H | Cin | Cout | GFLOP | bf16x3 us | TF/s | err | fp16x2 us | TF/s | err | amax us |")
    for H, Cin, Cout in shapes:
        x = torch.randn(n, H, H, Cin, device=d)
        w = torch.randn(Cout, Cin, 3, 3, device=d) / (3 * Cin ** 0.5)
        src = torch.randn(n, H, H, Cout, device=d)
        (wf, wb), = K.pack_weights([(w, "c3", True, True)])
        bparam = torch.zeros(Cout, device=d).requires_grad_(True)
        bparam.grad = torch.zeros(Cout, device=d)
        pool = K.AmaxPool(x, 64)
        xam = K.amax_of(x, pool.take())
        gf = 2.0 * n * H * H * 9 * Cin * Cout / 1e9
        ref = torch.nn.functional.conv2d(x[:2].permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
        ref = ref * torch.where(src[:2] > 0, 1.0, 0.2).double()
        err = lambda y: float((y[:2].double() - ref).abs().max() / ref.abs().max())
        f3 = lambda: K.conv3x3_f(x, wf, None, n, H, H, Cin, Cout, y_act_src=src, y_src_act=K.LEAKY, out_bias=bparam)[0]
        f2 = lambda: K.conv3x3_s(x, wf, None, n, H, H, Cin, Cout, xam, wf.mvk_amax, pool.take(), y_act_src=src,
                                 y_src_act=K.LEAKY, out_bias=bparam)[0]
        ok3 = K.conv3x3_fused_ok(n, H, H, Cin, Cout)
        t3, e3 = (t_us(f3), err(f3())) if ok3 else (float("nan"), float("nan"))
        t2, e2 = t_us(f2), err(f2())
        slot_ = torch.zeros(1, device=d)
        ta = t_us(lambda: K.amax_of(x, slot_), reps=50)
        wline = ""
        wparam = w.clone().requires_grad_(True)
        wparam.grad = torch.zeros_like(wparam)
        dyam = K.amax_of(src, pool.take())
        if K.conv3x3_fused_ok(n, H, H, Cin, Cout):
            w3 = t_us(lambda: K.conv3x3_wgrad_f(x, src, wparam, bparam, n, H, H, Cin, Cout, x_act=K.LEAKY, dy_scale=0.1))
            w2 = t_us(lambda: K.conv3x3_wgrad_s(x, src, wparam, bparam, n, H, H, Cin, Cout, xam, dyam, x_act=K.LEAKY, dy_scale=0.1))
            wline = f"    weight gradient of the row above (incl. ordered finish): bf16x3 {w3:.0f} us, fp16x2 {w2:.0f} us"
        if which == "f16one":  # plain forward form (no mask loads) and a cache-resident batch: is the loop waiting for memory?
            p3 = t_us(lambda: K.conv3x3_f(x, wf, None, n, H, H, Cin, Cout, pre_scale=0.5))
            p2 = t_us(lambda: K.conv3x3_s(x, wf, None, n, H, H, Cin, Cout, xam, wf.mvk_amax, pool.take()))
            lib.mvk_debug_set_flags(0x800)
            nn = 32
            xs_, ss_ = x[:nn].contiguous(), src[:nn].contiguous()
            q3 = t_us(lambda: K.conv3x3_f(xs_, wf, None, nn, H, H, Cin, Cout, y_act_src=ss_, y_src_act=K.LEAKY, out_bias=bparam)[0], reps=30)
            q2 = t_us(lambda: K.conv3x3_s(xs_, wf, None, nn, H, H, Cin, Cout, xam, wf.mvk_amax, pool.take(), y_act_src=ss_,
                                          y_src_act=K.LEAKY, out_bias=bparam)[0], reps=30)
            lib.mvk_debug_set_flags(0)
            print(f"    plain forward: bf16x3 {p3:.0f} us, fp16x2 {p2:.0f} us;  n = {nn} masked: bf16x3 {q3:.0f} us, fp16x2 {q2:.0f} us")
        print(f"| {H} | {Cin} | {Cout} | {gf:.1f} | {t3:.0f} | {gf / t3 * 1e3:.0f} | {e3:.1e} | {t2:.0f} | {gf / t2 * 1e3:.0f} | {e2:.1e} | {ta:.0f} |")
        if wline:
            print(wline)
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
