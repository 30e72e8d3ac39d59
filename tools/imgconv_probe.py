"""A/B probe of the register-stationary convolution kernels (csrc/imgconv.hip) against the implicit-GEMM engine:
correctness on random data (vs the engine and vs a float64 torch reference on the CPU for small batches) and
interleaved timings at the headline batch (n = 5120 images).  Usage: python tools/imgconv_probe.py [n] [rounds]"""
import ctypes
import math
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from multivae_amd import _lib, kernels as K  # noqa: E402

lib = _lib.load()
lib.mvk_debug_set_flags.argtypes = [ctypes.c_int]
d = torch.device("cuda:0")


def flags(f):
    lib.mvk_debug_set_flags(f)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


SHAPES = [(8, 32, 64), (4, 64, 128)]  # (h, Cu, Cv)


def check(n):
    for h, Cu, Cv in SHAPES:
        gen = torch.Generator().manual_seed(h * 1000 + n)
        U = torch.randn(n, Cu, 2 * h, 2 * h, generator=gen)
        V = torch.randn(n, Cv, h, h, generator=gen)
        Wc = torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(16 * Cu)
        bu, bv = torch.randn(Cu, generator=gen), torch.randn(Cv, generator=gen)
        Us, Vs = torch.randn(n, 2 * h, 2 * h, Cu, generator=gen).to(d), torch.randn(n, h, h, Cv, generator=gen).to(d)
        wd, wu = K.pack_conv(Wc.to(d))
        Ud, Vd = nhwc(U).to(d), nhwc(V).to(d)
        pb_u, pb_v = torch.nn.Parameter(torch.zeros(Cu, device=d)), torch.nn.Parameter(torch.zeros(Cv, device=d))
        outs = {}
        for name, f in (("new", 0x200), ("old", 0x100)):
            flags(f)
            K.DIRECT_GRAD = False
            up = K.conv_up(Vd, wu, bu.to(d), n, h, h, Cu, Cv, act=1)
            up2, gb_u = K.conv_up(Vd, wu, None, n, h, h, Cu, Cv, u_act_src=Us, u_act=1, out_bias=pb_u)
            dn = K.conv_down(Ud, wd, bv.to(d), n, h, h, Cu, Cv, act=1)
            dn2, gb_v = K.conv_down(Ud, wd, None, n, h, h, Cu, Cv, v_act_src=Vs, v_act=1, out_bias=pb_v)
            wg = K.conv_wgrad(Ud, Vd, Wc.to(d), n, h, h, Cu, Cv)
            torch.cuda.synchronize()
            outs[name] = [t.clone() for t in (up, up2, gb_u, dn, dn2, gb_v, wg)]
        flags(0)
        names = ["up+bias+relu", "up*mask", "up colsum", "down+bias+relu", "down*mask", "down colsum", "wgrad"]
        for nm, a, b in zip(names, outs["new"], outs["old"]):
            print(f"  h={h} n={n} {nm:16s} new vs engine {relerr(a, b):.2e}", flush=True)
        if n <= 64:
            ref_up = torch.relu(F.conv_transpose2d(V.double(), Wc.double(), bu.double(), stride=2, padding=1))
            ref_dn = torch.relu(F.conv2d(U.double(), Wc.double(), bv.double(), stride=2, padding=1))
            print(f"  h={h} n={n} up   vs float64: new {relerr(nchw(outs['new'][0].cpu()), ref_up):.2e} "
                  f"engine {relerr(nchw(outs['old'][0].cpu()), ref_up):.2e}")
            print(f"  h={h} n={n} down vs float64: new {relerr(nchw(outs['new'][3].cpu()), ref_dn):.2e} "
                  f"engine {relerr(nchw(outs['old'][3].cpu()), ref_dn):.2e}")
            Wr = Wc.double().clone().requires_grad_()
            (F.conv2d(U.double(), Wr, None, stride=2, padding=1) * V.double()).sum().backward()
            print(f"  h={h} n={n} wgrad vs float64: new {relerr(outs['new'][6].cpu(), Wr.grad):.2e} "
                  f"engine {relerr(outs['old'][6].cpu(), Wr.grad):.2e}")


def bench(n, rounds):
    for h, Cu, Cv in SHAPES:
        gen = torch.Generator().manual_seed(h)
        Ud = torch.randn(n, 2 * h, 2 * h, Cu, generator=gen).to(d)
        Vd = torch.randn(n, h, h, Cv, generator=gen).to(d)
        Wc = (torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(16 * Cu)).to(d)
        bu, bv = torch.randn(Cu, generator=gen).to(d), torch.randn(Cv, generator=gen).to(d)
        wd, wu = K.pack_conv(Wc)
        pb = torch.nn.Parameter(torch.zeros(Cv, device=d))
        K.DIRECT_GRAD = False
        cases = {
            "wgrd": lambda: K.conv_wgrad(Ud, Vd, Wc, n, h, h, Cu, Cv),
            "up  ": lambda: K.conv_up(Vd, wu, bu, n, h, h, Cu, Cv, act=1),
            "down": lambda: K.conv_down(Ud, wd, None, n, h, h, Cu, Cv, v_act_src=Vd, v_act=1, out_bias=pb),
        }
        gflop = 2.0 * n * h * h * 16 * Cu * Cv / 1e9
        for cname, fn in cases.items():
            times = {"new": [], "old": []}
            for r in range(rounds + 1):
                for name, f in (("new", 0x200), ("old", 0x100)):
                    flags(f)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        times[name].append(e0.elapsed_time(e1) / 5 * 1e3)
            flags(0)
            tn, to = sorted(times["new"]), sorted(times["old"])
            print(f"h={h} {cname} n={n}: new min {tn[0]:.1f} med {tn[len(tn)//2]:.1f} us ({gflop/tn[0]*1e3:.0f} TF/s)"
                  f" | engine min {to[0]:.1f} med {to[len(to)//2]:.1f} us", flush=True)


def prof(which, n=5120, reps=5):
    """reps launches of each case on one engine (for rocprofv3 --pmc runs)."""
    flags(0x200 if which == "new" else 0x100)
    for h, Cu, Cv in SHAPES:
        gen = torch.Generator().manual_seed(h)
        Ud = torch.randn(n, 2 * h, 2 * h, Cu, generator=gen).to(d)
        Vd = torch.randn(n, h, h, Cv, generator=gen).to(d)
        Wc = (torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(16 * Cu)).to(d)
        bu = torch.randn(Cu, generator=gen).to(d)
        wd, wu = K.pack_conv(Wc)
        pb = torch.nn.Parameter(torch.zeros(Cv, device=d))
        K.DIRECT_GRAD = False
        pool = K.AmaxPool(Ud, 8)
        uam, vam, y = K.amax_of(Ud, pool.take()), K.amax_of(Vd, pool.take()), pool.take()
        for _ in range(reps):
            K.conv_up(Vd, wu, bu, n, h, h, Cu, Cv, act=1)
            K.conv_down(Ud, wd, None, n, h, h, Cu, Cv, v_act_src=Vd, v_act=1, out_bias=pb)
            K.conv_wgrad(Ud, Vd, Wc, n, h, h, Cu, Cv)
            if which == "new":  # the same launches on scaled fp16 pairs (NP = 2 kernels)
                K.conv_up(Vd, wu, bu, n, h, h, Cu, Cv, act=1, amax=(vam, wu.mvk_amax, y))
                K.conv_down(Ud, wd, None, n, h, h, Cu, Cv, v_act_src=Vd, v_act=1, out_bias=pb, amax=(uam, wd.mvk_amax, y))
                K.conv_wgrad(Ud, Vd, Wc, n, h, h, Cu, Cv, amax=(uam, vam))
        torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "prof":
        prof(sys.argv[2] if len(sys.argv) > 2 else "new")
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    for nn in (2, 6, 130):
        check(nn)
    check(n)
    bench(n, rounds)
