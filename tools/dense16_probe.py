#!/usr/bin/env python3
"""csrc/dense16.hip against float64 on the CPU + kernel timings at the headline decoder batch (n = 5120).
usage: python tools/dense16_probe.py [M]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multivae_amd import _lib  # noqa: E402
from multivae_amd._lib import call, ptr, stream_ptr  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
L, Hd, D, B = 20, 512, 784, 512
dev = torch.device("cuda:0")
torch.manual_seed(0)
lib = _lib.load()
z = torch.randn(M, L, device=dev)
w0 = (torch.rand(Hd, L, device=dev) - 0.5) * 2 / L ** 0.5
b0 = (torch.rand(Hd, device=dev) - 0.5) * 2 / L ** 0.5
w1 = (torch.rand(D, Hd, device=dev) - 0.5) * 2 / Hd ** 0.5
b1 = (torch.rand(D, device=dev) - 0.5) * 2 / Hd ** 0.5
x = torch.rand(B, D, device=dev)
scale, gw = 0.75, 1.0 / M


def planes(R, Cc):
    t = torch.empty(2, R, Cc, dtype=torch.float16, device=dev)
    return t, t[0], t[1]


def unsplit(hi, lo, bound, R, Cc):
    out = torch.empty(R, Cc, device=dev)
    call("mvk_dense16_unsplit", ptr(hi), ptr(lo), ptr(bound), None, 1.0, 0, R, Cc, ptr(out), stream_ptr())
    return out


def rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())


# ---- weight planes
nk, nk_hi, nk_lo = planes(D, Hd)
kn, kn_hi, kn_lo = planes(Hd, D)
nk_inv, kn_inv = torch.empty(D, device=dev), torch.empty(Hd, device=dev)
call("mvk_dense16_pack", ptr(w1), D, Hd, ptr(nk_hi), ptr(nk_lo), ptr(nk_inv), ptr(kn_hi), ptr(kn_lo), ptr(kn_inv), stream_ptr())
w1r = (nk_hi.float() + nk_lo.float() / 2048) * nk_inv[:, None]
w1t = (kn_hi.float() + kn_lo.float() / 2048) * kn_inv[:, None]
print("pack NK rel", rel(w1r, w1.double().cpu()), " KN rel", rel(w1t, w1.t().double().cpu()))

# ---- first layer
zam = torch.zeros(1, device=dev)
call("mvk_amax", ptr(z), z.numel(), ptr(zam), stream_ptr())
hp, h_hi, h_lo = planes(M, Hd)
hb = torch.zeros(1, device=dev)
call("mvk_dense16_first", ptr(z), ptr(w0), ptr(b0), ptr(zam), ptr(h_hi), ptr(h_lo), ptr(hb), M, Hd, L, 1, stream_ptr())
h64 = torch.relu(z.double().cpu() @ w0.double().cpu().t() + b0.double().cpu())
h = unsplit(h_hi, h_lo, hb, M, Hd)
print("first rel", rel(h, h64), " bound", float(hb), " actual max", float(h64.max()))

# ---- forward + NLL tail
xam = torch.zeros(1, device=dev)
call("mvk_amax", ptr(x), x.numel(), ptr(xam), stream_ptr())
gp, g_hi, g_lo = planes(M, D)
gb = torch.zeros(8, device=dev)
NT, MT = lib.mvk_dense16_fwd_nll_rows(D), lib.mvk_dense16_colsum_rows(M)
rows_part = torch.zeros(NT, M, device=dev)
cs_part = torch.zeros(MT, D, device=dev)


def fwd():
    call("mvk_dense16_fwd_nll", ptr(h_hi), ptr(h_lo), ptr(hb), ptr(nk_hi), ptr(nk_lo), ptr(nk_inv), ptr(b1), ptr(x), B, ptr(xam),
         scale, gw, ptr(g_hi), ptr(g_lo), ptr(gb), ptr(rows_part), ptr(cs_part), M, D, Hd, stream_ptr())


fwd()
torch.cuda.synchronize()
pre = h64 @ w1.double().cpu().t() + b1.double().cpu()
r64 = torch.sigmoid(pre)
xx = x.double().cpu().repeat(M // B, 1)
import math
rows64 = (0.5 * (r64 - xx) ** 2 / scale ** 2).sum(1) + D * (math.log(scale) + 0.918938533204672742)
g64 = gw * (r64 - xx) / scale ** 2 * r64 * (1 - r64)
g = unsplit(g_hi, g_lo, gb, M, D)
print("fwd rows rel", rel(rows_part.sum(0), rows64), " G rel", rel(g, g64), " g_bound", float(gb[0]), " actual", float(g64.abs().max()))
print("colsum rel", rel(cs_part.sum(0), g64.sum(0)))

# ---- backward data
dh = torch.empty(M, Hd, device=dev)
db0 = torch.zeros(Hd, device=dev)
ws = torch.empty(64 << 20, device=dev)


def bwd_data():
    call("mvk_dense16_bwd_data", ptr(g_hi), ptr(g_lo), ptr(gb), ptr(kn_hi), ptr(kn_lo), ptr(kn_inv), ptr(h_hi), ptr(dh), ptr(db0),
         ptr(ws), ws.numel(), M, Hd, D, stream_ptr())


bwd_data()
torch.cuda.synchronize()
dh64 = (g64 @ w1.double().cpu()) * (h64 > 0)
print("bwd_data rel", rel(dh, dh64), " db0 rel", rel(db0, dh64.sum(0)))

# ---- weight gradient
dw1 = torch.zeros(D, Hd, device=dev)
db1 = torch.zeros(D, device=dev)


def wgrad():
    call("mvk_dense16_wgrad", ptr(g_hi), ptr(g_lo), ptr(gb), ptr(h_hi), ptr(h_lo), ptr(hb), ptr(cs_part), MT, ptr(dw1), ptr(db1),
         ptr(ws), ws.numel(), M, D, Hd, stream_ptr())


wgrad()
torch.cuda.synchronize()
dw64 = g64.t() @ h64
print("wgrad rel", rel(dw1, dw64), " db1 rel", rel(db1, g64.sum(0)))


def timeit(fn, name, flop):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    print(f"{name}: {us:.1f} us  ({flop / us / 1e6:.1f} TFLOP/s of fp32 work)")


fl = 2.0 * M * D * Hd
timeit(fwd, "fwd_nll", fl)
timeit(bwd_data, "bwd_data", fl)
timeit(wgrad, "wgrad", fl)
timeit(lambda: call("mvk_dense16_first", ptr(z), ptr(w0), ptr(b0), ptr(zam), ptr(h_hi), ptr(h_lo), ptr(hb), M, Hd, L, 1, stream_ptr()),
       "first", 2.0 * M * Hd * L)
timeit(lambda: call("mvk_dense16_pack", ptr(w1), D, Hd, ptr(nk_hi), ptr(nk_lo), ptr(nk_inv), ptr(kn_hi), ptr(kn_lo), ptr(kn_inv),
                    stream_ptr()), "pack", 1.0)

if os.environ.get("D16_ABLATE"):
    lib.mvk_dense16_debug.argtypes = [__import__("ctypes").c_int]
    lib.mvk_dense16_debug.restype = None
    stamps = torch.zeros(4, device=gb.device)
    lib.mvk_dense16_debug_stamps.argtypes = [__import__("ctypes").c_void_p]
    lib.mvk_dense16_debug_stamps.restype = None
    lib.mvk_dense16_debug_stamps(stamps.data_ptr())
    lib.mvk_dense16_debug(16)
    fwd()
    torch.cuda.synchronize()
    lib.mvk_dense16_debug_stamps(None)
    print("cycle stamps [main, epilogue] of the first / last workgroup:", stamps.tolist())
    for flags in (0, 8):
        lib.mvk_dense16_debug(flags)
        timeit(fwd, f"fwd_nll dbg={flags}", fl)
        timeit(bwd_data, f"bwd_data dbg={flags}", fl)
    lib.mvk_dense16_debug(0)
