"""Kernel-level parity: every C-ABI entry point of libmvk.so against the CPU oracle (oracle/) on seeded
inputs.  Tolerance: 1e-4 relative (BASELINE.json north_star), measured against the largest magnitude of the
reference tensor so that near-zero entries of a large tensor do not dominate; most kernels are far tighter.
All tests need a real MI355X (`-m gpu`)."""
import math
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_cases as G
from oracle import elbo, nets, train

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def dev():
    return torch.device("cuda:0")


def g(seed):
    return torch.Generator().manual_seed(seed)


ELEMENTWISE = re.compile(r"^(z\b|zs|kld|kl\b|joint|cond kl|subset|mu\b|lv\b|lw\b|u\b|w\b)")


def close_elementwise(got, ref, what, rtol=1e-4, atol_frac=1e-6):
    """|got - ref| <= rtol |ref| + atol_frac max|ref| for EVERY entry (latents, posterior parameters, KL rows, importance
    weights): small entries of a tensor are checked on their own scale, not against the tensor's maximum.  Matching
    infinities (log-variances of missing modalities) are equal."""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    same_inf = torch.isinf(ref) & (got == ref)
    fin = torch.isfinite(ref)
    assert bool((fin | same_inf).all()) and bool(torch.isfinite(got[fin]).all()), what
    if not bool(fin.any()):
        return
    tol = rtol * ref[fin].abs() + atol_frac * ref[fin].abs().max()
    bad = (got[fin] - ref[fin]).abs() > tol
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())} of {bad.numel()} entries outside rtol {rtol} + "
                                 f"{atol_frac} max; worst excess {float(((got[fin] - ref[fin]).abs() / tol).max()):.2f}x")


def close(got, ref, rtol=RTOL, what=""):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    if ELEMENTWISE.match(what):
        close_elementwise(got, ref, what, rtol=min(rtol, 1e-4))
    fin = torch.isfinite(ref)
    scale = ref[fin].abs().max().clamp_min(1e-30) if bool(fin.any()) else torch.tensor(1.0)
    err = ((got - ref).abs()[fin].max() if bool(fin.any()) else torch.tensor(0.0)) / scale
    assert torch.isfinite(got[fin]).all(), what
    assert float(err) <= rtol, f"{what}: rel-to-max err {float(err):.3e} > {rtol}"
    return float(err)


def close_per_slice(got, ref, keep, rtol, what):
    """Every slice [i] (an image, or an (image, channel) plane) on ITS OWN scale: max |got_i - ref_i| <= rtol max |ref_i| for the
    slices `keep` selects.  `close` is relative to the tensor's maximum, which says nothing about an image 1e-6 times smaller
    than the largest one; the scaled-fp16 kernels claim full precision for elements down to 2^-28 of the operand's maximum
    (csrc/bf3.hpp), i.e. for whole images / weight rows down to ~2^-20 of it."""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    k = keep.dim()
    g2, r2 = got.reshape(*got.shape[:k], -1), ref.reshape(*ref.shape[:k], -1)
    err = (g2 - r2).abs().amax(-1) / r2.abs().amax(-1).clamp_min(1e-300)
    assert bool(keep.any()), what
    worst = float(err[keep].max())
    assert worst <= rtol, f"{what}: worst per-slice error {worst:.3e} of the slice's own maximum > {rtol} ({int(keep.sum())} slices)"
    return worst


@pytest.fixture(scope="module")
def K():
    from multivae_amd import kernels

    return kernels


# ------------------------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K_,act", [(7, 5, 3, "relu"), (64, 40, 20, "none"), (300, 512, 784, "relu"),
                                        (513, 784, 512, "sigmoid"), (128, 128, 16, "none"), (33, 65, 130, "relu")])
def test_linear_fwd_bwd(K, M, N, K_, act):
    gen = g(M * 1000 + N)
    x = torch.randn(M, K_, generator=gen)
    w = torch.randn(N, K_, generator=gen) / math.sqrt(K_)
    b = torch.randn(N, generator=gen)
    dy = torch.randn(M, N, generator=gen)
    a = {"relu": 1, "sigmoid": 2, "none": 0}[act]
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y_ref = F.linear(xr, wr, br)
    y_ref = {"relu": torch.relu, "sigmoid": torch.sigmoid, "none": lambda t: t}[act](y_ref)
    y_ref.backward(dy)
    d = dev()
    xd, wd, bd, dyd = x.to(d), w.to(d), b.to(d), dy.to(d)
    y = K.linear_fwd(xd, wd, bd, a)
    close(y, y_ref, what="linear fwd")
    y_src = y if a else None
    dx = K.linear_bwd_data(dyd, wd, y_out=y_src, y_act=a)
    close(dx, xr.grad, what="linear bwd data")
    dw, db = K.linear_bwd_weight(dyd, xd, wd, bd, y_out=y_src, y_act=a)
    close(dw, wr.grad, what="linear bwd weight")
    close(db, br.grad, what="linear bwd bias")
    # direct accumulation into a pre-existing .grad (flat-buffer views): p.grad += dL/dp, autograd gets None
    wd.grad, bd.grad = torch.ones_like(wd), torch.ones_like(bd)
    r = K.linear_bwd_weight(dyd, xd, wd, bd, y_out=y_src, y_act=a)
    assert r == (None, None)
    close(wd.grad - 1, wr.grad, rtol=2e-4, what="direct-accumulated weight grad")
    close(bd.grad - 1, br.grad, rtol=2e-4, what="direct-accumulated bias grad")


@pytest.mark.parametrize("M,N,K_,act", [(32, 32, 12544, "none"), (128, 64, 16384, "none"), (20, 24, 4100, "relu"), (256, 32, 2048, "sigmoid")])
def test_linear_fwd_few_rows_long_reduction(K, M, N, K_, act):
    """mvk_linear_fwd where the output has a handful of 16 x 16 tiles and K is long (the heads of the ResNet encoders: 32 rows x
    12544 features at cfg4): the reduction is split over workgroups, raw slices go to the caller's scratch
    and heads_finish_kernel adds them in order with bias + activation.  Against float64 (exact-fp32 MFMA: 3e-6 of the largest
    output), ragged K / M / N, and bit-identical launch to launch."""
    gen = g(M + K_)
    x = torch.randn(M, K_, generator=gen)
    w = torch.randn(N, K_, generator=gen) / math.sqrt(K_)
    b = torch.randn(N, generator=gen)
    a = {"relu": 1, "sigmoid": 2, "none": 0}[act]
    f = {"relu": torch.relu, "sigmoid": torch.sigmoid, "none": lambda t: t}[act]
    y64 = f(F.linear(x.double(), w.double(), b.double()))
    d = dev()
    xd, wd, bd = x.to(d), w.to(d), b.to(d)
    y = K.linear_fwd(xd, wd, bd, a)
    err = float((y.double().cpu() - y64).abs().max() / y64.abs().max())
    assert err <= 3e-6, err
    assert torch.equal(y, K.linear_fwd(xd, wd, bd, a))


def test_linear_bwd_data_fused_prev_act_and_accumulate(K):
    gen = g(3)
    M, N, K_ = 50, 24, 36
    dy = torch.randn(M, N, generator=gen)
    w = torch.randn(N, K_, generator=gen)
    prev = torch.relu(torch.randn(M, K_, generator=gen))
    ref = (dy @ w) * (prev > 0).float()
    d = dev()
    got = K.linear_bwd_data(dy.to(d), w.to(d), prev_out=prev.to(d), prev_act=1)
    close(got, ref, what="prev-act fusion")
    K.linear_bwd_data(dy.to(d), w.to(d), prev_out=prev.to(d), prev_act=1, out=got, accumulate=True)
    close(got, 2 * ref, what="accumulate")


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_modes(K, ta, tb):
    gen = g(10 + 2 * ta + tb)
    M, N, K_ = 45, 52, 28
    A = torch.randn(M, K_, generator=gen)
    B = torch.randn(K_, N, generator=gen)
    bias = torch.randn(13, generator=gen)
    ref = A @ B + bias[torch.arange(N) % 13]
    d = dev()
    a_in = (A.t().contiguous() if ta else A).to(d)
    b_in = (B.t().contiguous() if tb else B).to(d)
    got = K.gemm(a_in, b_in, M, N, K_, ta=bool(ta), tb=bool(tb), bias=bias.to(d), bias_mod=13)
    close(got, ref, what=f"gemm ta={ta} tb={tb}")


def test_gemm_splitk_accumulate(K):
    gen = g(5)
    M, N, K_ = 40, 24, 5000
    A = torch.randn(M, K_, generator=gen)
    B = torch.randn(K_, N, generator=gen)
    C0 = torch.randn(M, N, generator=gen)
    d = dev()
    out = C0.to(d).clone()
    K.gemm(A.to(d), B.to(d), M, N, K_, out=out, accumulate=True)
    close(out, C0 + A @ B, what="split-K accumulate")


# ------------------------------------------------------------------------------------------------------------
# 4x4 / stride 2 / pad 1 convolution pair (NHWC inside)
# ------------------------------------------------------------------------------------------------------------
def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("n,h,w,Cu,Cv", [(3, 2, 3, 8, 12), (5, 4, 4, 64, 128), (4, 8, 8, 32, 64), (2, 16, 16, 4, 32),
                                         (130, 4, 4, 16, 20), (70, 16, 16, 3, 32), (5, 8, 6, 1, 16), (3, 16, 16, 3, 64),
                                         (300, 16, 16, 2, 64)])
def test_conv_down_up_wgrad(K, n, h, w, Cu, Cv):
    gen = g(n * 100 + h)
    U = torch.randn(n, Cu, 2 * h, 2 * w, generator=gen)
    V = torch.randn(n, Cv, h, w, generator=gen)
    Wc = torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(16 * Cu)  # conv weight [out=Cv, in=Cu]
    bv = torch.randn(Cv, generator=gen)
    bu = torch.randn(Cu, generator=gen)
    d = dev()
    wd, wu = K.pack_conv(Wc.to(d))
    # down: V' = relu(conv(U)+b)
    ref_down = torch.relu(F.conv2d(U, Wc, bv, stride=2, padding=1))
    got = K.conv_down(nhwc(U).to(d), wd, bv.to(d), n, h, w, Cu, Cv, act=1)
    close(nchw(got.cpu()), ref_down, what="conv down")
    # down from an NCHW source
    got2 = K.conv_down(U.to(d), wd, bv.to(d), n, h, w, Cu, Cv, act=1, u_nchw=True)
    close(nchw(got2.cpu()), ref_down, what="conv down nchw src")
    # up: U' = convT(V) + b  (ConvTranspose2d weight [in=Cv, out=Cu] is the same tensor)
    ref_up = F.conv_transpose2d(V, Wc, bu, stride=2, padding=1)
    got = K.conv_up(nhwc(V).to(d), wu, bu.to(d), n, h, w, Cu, Cv)
    close(nchw(got.cpu()), ref_up, what="conv up")
    got3 = K.conv_up(nhwc(V).to(d), wu, bu.to(d), n, h, w, Cu, Cv, u_nchw=True)
    close(got3.cpu(), ref_up, what="conv up nchw dst")
    # wgrad: d/dW of sum(conv(U) * V)
    Wr = Wc.clone().requires_grad_()
    (F.conv2d(U, Wr, None, stride=2, padding=1) * V).sum().backward()
    Wd = Wc.to(d)
    got = K.conv_wgrad(nhwc(U).to(d), nhwc(V).to(d), Wd, n, h, w, Cu, Cv)
    close(got, Wr.grad, what="conv wgrad")
    got4 = K.conv_wgrad(U.to(d), nhwc(V).to(d), Wd, n, h, w, Cu, Cv, u_nchw=True)
    close(got4, Wr.grad, what="conv wgrad nchw src")


def test_conv_pins_against_numpy_definition(K):
    """The same kernels against the torch-free numpy definitions of oracle/nets.py (small case)."""
    gen = g(77)
    n, h, w, Cu, Cv = 2, 2, 2, 4, 8
    U = torch.randn(n, Cu, 2 * h, 2 * w, generator=gen)
    V = torch.randn(n, Cv, h, w, generator=gen)
    Wc = torch.randn(Cv, Cu, 4, 4, generator=gen)
    b0 = torch.zeros(Cv)
    d = dev()
    wd, wu = K.pack_conv(Wc.to(d))
    got = K.conv_down(nhwc(U).to(d), wd, None, n, h, w, Cu, Cv)
    close(nchw(got.cpu()), torch.from_numpy(nets.conv2d_np(U, Wc, b0, 2, 1)).float(), what="down vs numpy")
    got = K.conv_up(nhwc(V).to(d), wu, None, n, h, w, Cu, Cv)
    close(nchw(got.cpu()), torch.from_numpy(nets.conv_transpose2d_np(V, Wc, torch.zeros(Cu), 2, 1)).float(),
          what="up vs numpy")


@pytest.mark.parametrize("n,h,Cv,Cu", [(3, 16, 32, 3), (2, 4, 8, 1), (65, 16, 32, 3)])
def test_up_nchw_small(K, n, h, Cv, Cu):
    from multivae_amd._lib import call, ptr, stream_ptr

    gen = g(n + h)
    V = torch.randn(n, Cv, h, h, generator=gen)
    Wt = torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(4 * Cv)
    b = torch.randn(Cu, generator=gen)
    ref = torch.sigmoid(F.conv_transpose2d(V, Wt, b, stride=2, padding=1))
    d = dev()
    out = torch.empty(n, Cu, 2 * h, 2 * h, device=d)
    Vd, Wd, bd = nhwc(V).to(d), Wt.to(d), b.to(d)
    call("mvk_conv4s2_up_nchw_small", ptr(Vd), ptr(Wd), ptr(bd), ptr(out), n, h, h, Cu, Cv, 2, stream_ptr())
    close(out, ref, what="up nchw small")


def _debug_flags(f):
    import ctypes

    from multivae_amd import _lib

    lib = _lib.load()
    lib.mvk_debug_set_flags.argtypes = [ctypes.c_int]
    lib.mvk_debug_set_flags(f)


@pytest.mark.parametrize("n", [2, 6, 130, 700])
@pytest.mark.parametrize("h,Cu,Cv", [(8, 32, 64), (4, 64, 128)])
def test_imgconv_register_stationary_kernels(K, n, h, Cu, Cv):
    """csrc/imgconv.hip (weights resident in registers, images streamed through LDS) against a float64 convolution:
    forward form (bias + ReLU) and backward-data form (x ReLU'(saved activation), bias-gradient column sums), for both
    directions of both SVHN layer pairs; batches smaller and larger than the 256 workgroups, odd and even."""
    gen = g(h * 1000 + n)
    U = torch.randn(n, Cu, 2 * h, 2 * h, generator=gen)
    V = torch.randn(n, Cv, h, h, generator=gen)
    Wc = torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(16 * Cu)
    bu, bv = torch.randn(Cu, generator=gen), torch.randn(Cv, generator=gen)
    Us, Vs = torch.randn(n, Cu, 2 * h, 2 * h, generator=gen), torch.randn(n, Cv, h, h, generator=gen)
    d = dev()
    wd, wu = K.pack_conv(Wc.to(d))
    pb_u, pb_v = torch.nn.Parameter(torch.zeros(Cu, device=d)), torch.nn.Parameter(torch.zeros(Cv, device=d))
    Ud, Vd = nhwc(U).to(d), nhwc(V).to(d)
    direct = K.DIRECT_GRAD
    K.DIRECT_GRAD = False
    _debug_flags(0x200)  # take the kernels for every batch size
    try:
        up = K.conv_up(Vd, wu, bu.to(d), n, h, h, Cu, Cv, act=1)
        up2, gb_u = K.conv_up(Vd, wu, None, n, h, h, Cu, Cv, u_act_src=nhwc(Us).to(d), u_act=1, out_bias=pb_u)
        dn = K.conv_down(Ud, wd, bv.to(d), n, h, h, Cu, Cv, act=1)
        dn2, gb_v = K.conv_down(Ud, wd, None, n, h, h, Cu, Cv, v_act_src=nhwc(Vs).to(d), v_act=1, out_bias=pb_v)
        wg = K.conv_wgrad(Ud, Vd, Wc.to(d), n, h, h, Cu, Cv)  # transposing-read weight-gradient kernel for n >= 4
        torch.cuda.synchronize()
    finally:
        _debug_flags(0)
        K.DIRECT_GRAD = direct
    ref_up = F.conv_transpose2d(V.double(), Wc.double(), None, stride=2, padding=1)
    ref_dn = F.conv2d(U.double(), Wc.double(), None, stride=2, padding=1)
    close(nchw(up.cpu()), torch.relu(ref_up + bu.double().view(1, -1, 1, 1)), rtol=2e-6, what="imgconv up")
    close(nchw(dn.cpu()), torch.relu(ref_dn + bv.double().view(1, -1, 1, 1)), rtol=2e-6, what="imgconv down")
    ref_up2 = ref_up * (Us > 0)
    ref_dn2 = ref_dn * (Vs > 0)
    close(nchw(up2.cpu()), ref_up2, rtol=2e-6, what="imgconv up x mask")
    close(nchw(dn2.cpu()), ref_dn2, rtol=2e-6, what="imgconv down x mask")
    close(gb_u, ref_up2.sum((0, 2, 3)), rtol=1e-5, what="imgconv up column sums")
    close(gb_v, ref_dn2.sum((0, 2, 3)), rtol=1e-5, what="imgconv down column sums")
    Wr = Wc.double().clone().requires_grad_()
    (F.conv2d(U.double(), Wr, None, stride=2, padding=1) * V.double()).sum().backward()
    close(wg, Wr.grad, rtol=2e-6, what="imgconv wgrad")


@pytest.mark.parametrize("n", [2, 130, 700])
@pytest.mark.parametrize("h,Cu,Cv", [(8, 32, 64), (4, 64, 128)])
@pytest.mark.parametrize("spread", [0.0, 3.0])
def test_imgconv_scaled_fp16(K, n, h, Cu, Cv, spread):
    """mvk_conv4s2_down_s / _up_s (csrc/imgconv.hip NP = 2): 3 fp16 MFMAs per product on scaled (hi, lo) pairs, weights
    converted in the kernel from the fp32 pack with the maximum the pack launch published.  Same float64 reference and the
    tolerance of test_imgconv_register_stationary_kernels, on unit-scale data and on data whose images spread over 8 orders of
    magnitude; max |result| as published by the launch is exact; with y_amax alone the bf16-piece kernel publishes too."""
    gen = g(h * 1000 + n + 7)
    U = torch.randn(n, Cu, 2 * h, 2 * h, generator=gen) * torch.exp(spread * torch.randn(n, 1, 1, 1, generator=gen)) * 2.3
    V = torch.randn(n, Cv, h, h, generator=gen) * torch.exp(spread * torch.randn(n, 1, 1, 1, generator=gen)) * 0.04
    Wc = torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(16 * Cu)
    bu, bv = torch.randn(Cu, generator=gen), torch.randn(Cv, generator=gen)
    Us, Vs = torch.randn(n, Cu, 2 * h, 2 * h, generator=gen), torch.randn(n, Cv, h, h, generator=gen)
    d = dev()
    wd, wu = K.pack_conv(Wc.to(d))
    assert float(wd.mvk_amax) == float(Wc.abs().max()) and float(wu.mvk_amax) == float(Wc.abs().max())
    pb_u, pb_v = torch.nn.Parameter(torch.zeros(Cu, device=d)), torch.nn.Parameter(torch.zeros(Cv, device=d))
    Ud, Vd = nhwc(U).to(d), nhwc(V).to(d)
    pool = K.AmaxPool(Ud, 16)
    uam, vam = K.amax_of(Ud, pool.take()), K.amax_of(Vd, pool.take())
    direct = K.DIRECT_GRAD
    K.DIRECT_GRAD = False
    _debug_flags(0x200)  # take the kernels for every batch size
    try:
        if h == 4 and n % 2:
            assert not K.conv4s2_scaled_ok(n, h, h, Cu, Cv)
            return
        assert K.conv4s2_scaled_ok(n, h, h, Cu, Cv)
        y1, y2, y3, y4, y5 = (pool.take() for _ in range(5))
        up = K.conv_up(Vd, wu, bu.to(d), n, h, h, Cu, Cv, act=1, amax=(vam, wu.mvk_amax, y1))
        up2, gb_u = K.conv_up(Vd, wu, None, n, h, h, Cu, Cv, u_act_src=nhwc(Us).to(d), u_act=1, out_bias=pb_u,
                              amax=(vam, wu.mvk_amax, y2))
        dn = K.conv_down(Ud, wd, bv.to(d), n, h, h, Cu, Cv, act=1, amax=(uam, wd.mvk_amax, y3))
        dn2, gb_v = K.conv_down(Ud, wd, None, n, h, h, Cu, Cv, v_act_src=nhwc(Vs).to(d), v_act=1, out_bias=pb_v,
                                amax=(uam, wd.mvk_amax, y4))
        dn3 = K.conv_down(Ud, wd, bv.to(d), n, h, h, Cu, Cv, act=1, amax=(None, None, y5))  # bf16 pieces, publishing
        wg = None
        if n >= 4:  # imgwgrad_kernel NP = 2 (one accumulator per tile, V in three pieces); loose bounds on purpose
            wg = K.conv_wgrad(Ud, Vd, Wc.to(d), n, h, h, Cu, Cv, amax=((uam * 3.0).contiguous(), (vam * 20.0).contiguous()))
        torch.cuda.synchronize()
    finally:
        _debug_flags(0)
        K.DIRECT_GRAD = direct
    ref_up = F.conv_transpose2d(V.double(), Wc.double(), None, stride=2, padding=1)
    ref_dn = F.conv2d(U.double(), Wc.double(), None, stride=2, padding=1)
    close(nchw(up.cpu()), torch.relu(ref_up + bu.double().view(1, -1, 1, 1)), rtol=2e-6, what="imgconv up")
    close(nchw(dn.cpu()), torch.relu(ref_dn + bv.double().view(1, -1, 1, 1)), rtol=2e-6, what="imgconv down")
    close(nchw(dn3.cpu()), torch.relu(ref_dn + bv.double().view(1, -1, 1, 1)), rtol=2e-6, what="imgconv down, bf16 pieces")
    ref_up2 = ref_up * (Us > 0)
    ref_dn2 = ref_dn * (Vs > 0)
    close(nchw(up2.cpu()), ref_up2, rtol=2e-6, what="imgconv up x mask")
    close(nchw(dn2.cpu()), ref_dn2, rtol=2e-6, what="imgconv down x mask")
    close(gb_u, ref_up2.sum((0, 2, 3)), rtol=1e-5, what="imgconv up column sums")
    close(gb_v, ref_dn2.sum((0, 2, 3)), rtol=1e-5, what="imgconv down column sums")
    for slot, t in ((y1, up), (y2, up2), (y3, dn), (y4, dn2), (y5, dn3)):
        assert float(slot) == float(t.abs().max()), "published max |result|"
    # image by image on the image's own scale (the bias-free results): every image down to 2^-20 of the largest one
    su, sv = U.abs().amax((1, 2, 3)), V.abs().amax((1, 2, 3))
    close_per_slice(nchw(up2.cpu()), ref_up2, sv >= sv.max() * 2.0 ** -20, 6e-6, "imgconv up x mask, per image")
    close_per_slice(nchw(dn2.cpu()), ref_dn2, su >= su.max() * 2.0 ** -20, 6e-6, "imgconv down x mask, per image")
    if wg is not None:
        Wr = Wc.double().clone().requires_grad_()
        (F.conv2d(U.double(), Wr, None, stride=2, padding=1) * V.double()).sum().backward()
        close(wg, Wr.grad, rtol=2e-6, what="imgconv wgrad, scaled fp16")


# ------------------------------------------------------------------------------------------------------------
# whole networks (single autograd nodes) against the oracle's functional networks
# ------------------------------------------------------------------------------------------------------------
def _sd(shapes, seed):
    import golden_cases as G

    return {k: torch.from_numpy(v) for k, v in G.P.make_state_dict(shapes, seed).items()}


def _grads_close(mod, prefix, sd_ref, rtol=RTOL):
    for name, p in mod.named_parameters():
        ref = sd_ref[prefix + name].grad
        close(p.grad, ref, rtol=rtol, what=f"grad {prefix}{name}")


@pytest.fixture(params=["default dispatch", "register-stationary kernels"])
def svhn_engine(request, monkeypatch):
    """The size-based dispatch (tiled engine at small batches), then csrc/imgconv.hip for every batch size (debug flag 0x200)
    with the scaled-fp16 product form of the decoder from the first row on."""
    from multivae_amd import kernels as K_

    if request.param != "default dispatch":
        monkeypatch.setattr(K_, "IMG_F16_MIN_ROWS", 1)
        _debug_flags(0x200)
    yield request.param
    _debug_flags(0)


@pytest.mark.parametrize("B", [1, 5, 64])
def test_svhn_encoder_decoder_nodes(B, svhn_engine):
    import golden_cases as G
    from multivae_amd.models.base.base_config import BaseAEConfig
    from multivae_amd.models.nn.svhn import Decoder_VAE_SVHN, Encoder_VAE_SVHN

    L = 20
    shapes = {}
    shapes.update(G.P.svhn_encoder_shapes("enc.", L))
    shapes.update(G.P.svhn_decoder_shapes("dec.", L))
    sd = _sd(shapes, 42 + B)
    gen = g(B)
    x = torch.rand(B, 3, 32, 32, generator=gen)
    z = torch.randn(2, B, L, generator=gen)
    d = dev()
    enc = Encoder_VAE_SVHN(BaseAEConfig(input_dim=(3, 32, 32), latent_dim=L)).to(d)
    dec = Decoder_VAE_SVHN(BaseAEConfig(input_dim=(3, 32, 32), latent_dim=L)).to(d)
    enc.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("enc.")})
    dec.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("dec.")})
    ref = {k: v.clone().requires_grad_() for k, v in sd.items()}
    mu_r, lv_r = nets.svhn_encoder(ref, "enc.", x)
    out = enc(x.to(d))
    close(out.embedding, mu_r, what="svhn enc mu")
    close(out.log_covariance, lv_r, what="svhn enc lv")
    gm, gl = torch.randn(mu_r.shape, generator=gen), torch.randn(lv_r.shape, generator=gen)
    (mu_r * gm + lv_r * gl).sum().backward()
    (out.embedding * gm.to(d) + out.log_covariance * gl.to(d)).sum().backward()
    _grads_close(enc, "enc.", ref)
    zr = z.clone().requires_grad_()
    rec_r = nets.svhn_decoder(ref, "dec.", zr)
    zd = z.to(d).requires_grad_()
    rec = dec(zd).reconstruction
    assert rec.shape == (2, B, 3, 32, 32)
    close(rec, rec_r, what="svhn dec out")
    gr = torch.randn(rec_r.shape, generator=gen)
    (rec_r * gr).sum().backward()
    (rec * gr.to(d)).sum().backward()
    _grads_close(dec, "dec.", ref)
    close(zd.grad, zr.grad, what="svhn dec dz")


def test_copy_batch(K):
    """mvk_copy_batch: up to 8 device-to-device copies in one launch (the modalities of a batch into a captured step's input
    buffers): every byte arrives, neighbours are untouched, sizes that are not a multiple of a tile, an empty entry, and the
    argument checks (misaligned pointer, odd byte count, too many entries)."""
    import ctypes as C

    from multivae_amd import _lib

    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    shapes = [(512, 784), (512, 3, 32, 32), (7, 4), (0, 8), (4099,)]
    srcs = [torch.randn(*sh, generator=g).to(d) for sh in shapes[:-1]] + [torch.randn(4 * 4099, generator=g).to(d)]
    guard = 64
    dsts = [torch.full((s.numel() + 2 * guard,), -7.0, device=d) for s in srcs]
    descs = (_lib.CopyDesc * len(srcs))()
    for e, dst, s in zip(descs, dsts, srcs):
        e.dst, e.src, e.bytes = dst[guard:].data_ptr(), s.data_ptr(), 4 * s.numel()
    _lib.call("mvk_copy_batch", descs, len(srcs), _lib.stream_ptr())
    torch.cuda.synchronize()
    for dst, s in zip(dsts, srcs):
        assert torch.equal(dst[guard:guard + s.numel()], s.reshape(-1))
        assert bool((dst[:guard] == -7.0).all()) and bool((dst[guard + s.numel():] == -7.0).all())
    lib = _lib.load()
    bad = (_lib.CopyDesc * 1)()
    bad[0].dst, bad[0].src, bad[0].bytes = dsts[0][guard + 1:].data_ptr(), srcs[0].data_ptr(), 64  # 4 bytes off a 16-byte boundary
    assert lib.mvk_copy_batch(bad, 1, None) == -1
    bad[0].dst, bad[0].bytes = dsts[0][guard:].data_ptr(), 60
    assert lib.mvk_copy_batch(bad, 1, None) == -1
    assert lib.mvk_copy_batch((_lib.CopyDesc * 9)(), 9, None) == -1 and lib.mvk_copy_batch(None, 0, None) == 0


@pytest.mark.parametrize("fused_heads_bwd", [False, True])
@pytest.mark.parametrize("B,D", [(6, (2,)), (33, (1, 28, 28)), (512, (3, 4))])
def test_mlp_encoder_decoder_nodes(B, D, fused_heads_bwd, monkeypatch):
    """Encoder_VAE_MLP / Decoder_AE_MLP nodes vs the oracle networks; the encoder also with the heads' backward in one
    launch (mvk_heads_bwd; kernels.heads_bwd_mlp / MVK_HEADS_BWD_MLP decide for the MLP encoders)."""
    import golden_cases as G
    from multivae_amd import kernels as K_

    monkeypatch.setattr(K_, "_HEADS_BWD_MLP_MODE", "1" if fused_heads_bwd else "0")
    from multivae_amd.models.base.base_config import BaseAEConfig
    from multivae_amd.models.nn.default_architectures import Decoder_AE_MLP, Encoder_VAE_MLP

    L = 7
    nin = int(np.prod(D))
    shapes = {}
    shapes.update(G.P.mlp_encoder_shapes("enc.", nin, L))
    shapes.update(G.P.mlp_decoder_shapes("dec.", L, nin))
    sd = _sd(shapes, 7 + B)
    gen = g(B + 1)
    x = torch.rand(B, *D, generator=gen)
    z = torch.randn(3, B, L, generator=gen)
    d = dev()
    enc = Encoder_VAE_MLP(BaseAEConfig(input_dim=D, latent_dim=L)).to(d)
    dec = Decoder_AE_MLP(BaseAEConfig(input_dim=D, latent_dim=L)).to(d)
    enc.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("enc.")})
    dec.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("dec.")})
    ref = {k: v.clone().requires_grad_() for k, v in sd.items()}
    mu_r, lv_r = nets.mlp_encoder(ref, "enc.", x)
    out = enc(x.to(d))
    close(out.embedding, mu_r, what="mlp enc mu")
    close(out.log_covariance, lv_r, what="mlp enc lv")
    gm, gl = torch.randn(mu_r.shape, generator=gen), torch.randn(lv_r.shape, generator=gen)
    (mu_r * gm + lv_r * gl).sum().backward()
    (out.embedding * gm.to(d) + out.log_covariance * gl.to(d)).sum().backward()
    _grads_close(enc, "enc.", ref)
    zr = z.clone().requires_grad_()
    rec_r = nets.mlp_decoder(ref, "dec.", zr, D)
    zd = z.to(d).requires_grad_()
    rec = dec(zd).reconstruction
    assert rec.shape == (3, B, *D)
    close(rec, rec_r, what="mlp dec out")
    gr = torch.randn(rec_r.shape, generator=gen)
    (rec_r * gr).sum().backward()
    (rec * gr.to(d)).sum().backward()
    _grads_close(dec, "dec.", ref)
    close(zd.grad, zr.grad, what="mlp dec dz")


# ------------------------------------------------------------------------------------------------------------
# fused ELBO kernels
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dist,scale", [("normal", 1.0), ("normal", 0.75), ("laplace", 0.5), ("bernoulli", 1.0)])
@pytest.mark.parametrize("Kk,B,D", [(1, 6, 3), (10, 17, 784), (9, 5, 3072), (17, 3, 4100)])
def test_recon_nll(dist, scale, Kk, B, D):
    from multivae_amd._lib import DIST, ReconDesc, call, stream_ptr

    gen = g(Kk * 7 + B)
    recon = torch.randn(Kk, B, D, generator=gen)
    x = torch.rand(B, D, generator=gen)
    if dist == "bernoulli":
        x = (x > 0.5).float()
    mask = torch.rand(B, generator=gen) > 0.3
    rowcoef = torch.rand(Kk, B, generator=gen)
    rescale, coef = 1.7, 0.3
    rr = recon.clone().requires_grad_()
    rows_ref = elbo._row_nll(dist, rr, x, rescale, scale)  # [K,B]
    (rows_ref * rowcoef * mask.float() * coef).sum().backward()
    d = dev()
    rd, xd, md, rcd = recon.to(d), x.to(d), mask.to(d), rowcoef.to(d)
    rows = torch.empty(Kk, B, device=d)
    drecon = torch.empty_like(rd)
    desc = (ReconDesc * 1)()
    e = desc[0]
    e.recon, e.x, e.mask, e.rows, e.drecon, e.rowcoef = (rd.data_ptr(), xd.data_ptr(), md.data_ptr(),
                                                         rows.data_ptr(), drecon.data_ptr(), rcd.data_ptr())
    e.D, e.dist, e.scale, e.rescale, e.coef = D, DIST[dist], scale, rescale, coef
    call("mvk_recon_nll_fwd", desc, 1, Kk, B, stream_ptr())
    close(rows, rows_ref, what="nll rows")
    close(drecon, rr.grad, what="nll drecon (fused)")
    drecon2 = torch.zeros_like(rd)
    e.drecon, e.rows = drecon2.data_ptr(), None
    call("mvk_recon_nll_bwd", desc, 1, Kk, B, stream_ptr())
    close(drecon2, rr.grad, what="nll drecon (second pass)")


def test_recon_nll_multi_modality_one_launch():
    from multivae_amd._lib import ReconDesc, call, stream_ptr

    gen = g(4)
    Kk, B = 3, 9
    d = dev()
    Ds, keep = [784, 3072, 5], []
    desc = (ReconDesc * 3)()
    refs = []
    for i, D in enumerate(Ds):
        recon = torch.randn(Kk, B, D, generator=gen)
        x = torch.rand(B, D, generator=gen)
        refs.append(elbo._row_nll("normal", recon, x, 1.0 + i, 1.0))
        rd, xd = recon.to(d), x.to(d)
        rows = torch.empty(Kk, B, device=d)
        keep += [rd, xd, rows]
        e = desc[i]
        e.recon, e.x, e.mask, e.rows, e.drecon, e.rowcoef = rd.data_ptr(), xd.data_ptr(), None, rows.data_ptr(), None, None
        e.D, e.dist, e.scale, e.rescale, e.coef = D, 0, 1.0, 1.0 + i, 1.0
    call("mvk_recon_nll_fwd", desc, 3, Kk, B, stream_ptr())
    for i in range(3):
        close(keep[3 * i + 2], refs[i], what=f"rows modality {i}")


def _subsets_bits(names):
    order = sorted(names)
    pos = {m: i for i, m in enumerate(order)}
    subs = elbo.mopoe_subsets(names)
    return order, subs, [sum(1 << pos[m] for m in mods) for _, mods in subs]


@pytest.mark.parametrize("M,B,L,Kk,masked", [(2, 16, 20, 1, False), (2, 512, 20, 10, False), (4, 9, 5, 3, True),
                                             (5, 33, 70, 2, False)])
def test_mopoe_posterior(K, M, B, L, Kk, masked):
    gen = g(M * 31 + B)
    names = [f"m{(i * 3) % M}x{i}" for i in range(M)]  # deliberately not sorted
    enc = {m: (torch.randn(B, L, generator=gen), torch.randn(B, L, generator=gen) * 0.5) for m in names}
    eps = torch.randn(Kk, B, L, generator=gen)
    order, subs, bits = _subsets_bits(names)
    S = len(subs)
    masks = choice = None
    if masked:
        masks = {m: torch.rand(B, generator=gen) > 0.3 for m in names}
        masks[names[0]][:] = True
    encr = {m: (a.clone().requires_grad_(), b.clone().requires_grad_()) for m, (a, b) in enc.items()}
    if masked:
        inf0 = elbo.mopoe_inference(enc, names, masks=masks, choice=torch.eye(S)[torch.zeros(B, dtype=torch.long)])
        idx = torch.multinomial(inf0["weights"].t(), 1, generator=gen).squeeze(1)
        choice = torch.eye(S)[idx]
    inf = elbo.mopoe_inference(encr, names, masks=masks, choice=choice)
    z_ref = elbo.rsample(inf["joint_mu"], inf["joint_logvar"], eps)
    kld_ref, klds = elbo.mopoe_joint_divergence(inf["mus"], inf["logvars"], inf["weights"])
    dz = torch.randn(Kk, B, L, generator=gen)
    gk = torch.rand(B, generator=gen)
    kld_rows_ref = (inf["weights"] * klds).sum(0)
    ((z_ref * dz).sum() + (kld_rows_ref * gk).sum()).backward()
    d = dev()
    if masked:
        sel = choice.argmax(1).to(torch.int32).to(d)
        weights = inf["weights"].detach().contiguous().to(d)
    else:
        bnd = elbo.mopoe_row_bounds(B, S)
        sel = torch.zeros(B, dtype=torch.int32)
        for k in range(S):
            sel[bnd[k]:bnd[k + 1]] = k
        sel, weights = sel.to(d), None
    mus = [enc[m][0].to(d).requires_grad_() for m in order]
    lvs = [enc[m][1].to(d).requires_grad_() for m in order]
    outs = K.MoPoEPosteriorFn.apply(eps.to(d), torch.tensor(bits, dtype=torch.int32, device=d), sel, weights, True,
                                    *mus, *lvs)
    z, kld_rows, mus_o, lvs_o, jmu, jlv = outs
    close(z, z_ref, what="z")
    close(kld_rows, kld_rows_ref, what="kld rows")
    close(mus_o, inf["mus"], what="subset mus")
    close(lvs_o, inf["logvars"], what="subset logvars")
    close(jmu, inf["joint_mu"], what="joint mu")
    ((z * dz.to(d)).sum() + (kld_rows * gk.to(d)).sum()).backward()
    for i, m in enumerate(order):
        close(mus[i].grad, encr[m][0].grad, what=f"dmu {m}")
        close(lvs[i].grad, encr[m][1].grad, what=f"dlv {m}")


@pytest.mark.parametrize("M,B,L,Kk,masked", [(2, 64, 20, 1, False), (4, 9, 5, 2, True), (3, 40, 33, 1, True)])
def test_mvtcae_posterior(K, M, B, L, Kk, masked):
    gen = g(M + B)
    names = [f"m{i}" for i in range(M)]
    mus_c = [torch.randn(B, L, generator=gen) for _ in range(M)]
    lvs_c = [torch.randn(B, L, generator=gen) * 0.5 for _ in range(M)]
    eps = torch.randn(Kk, B, L, generator=gen)
    masks = None
    if masked:
        masks = [torch.rand(B, generator=gen) > 0.35 for _ in range(M)]
        masks[0][:] = True
    mur = [t.clone().requires_grad_() for t in mus_c]
    lvr = [t.clone().requires_grad_() for t in lvs_c]
    lv_eff = [torch.where(masks[i].unsqueeze(-1), lvr[i], torch.full_like(lvr[i], float("inf"))) if masked else lvr[i]
              for i in range(M)]
    jmu, jlv = elbo.poe(torch.stack(mur), torch.stack(lv_eff))
    z_ref = elbo.rsample(jmu, jlv, eps)
    jkl_ref = -0.5 * (1 - jlv.exp() - jmu.pow(2) + jlv).sum(-1)
    ckl_ref = []
    for i in range(M):
        k = -0.5 * (1 - jlv.exp() / lv_eff[i].exp() - (jmu - mur[i]).pow(2) / lv_eff[i].exp() + jlv - lv_eff[i]).sum(-1)
        if masked:
            k = torch.where(masks[i], k, torch.zeros_like(k))
        ckl_ref.append(k)
    ckl_ref = torch.stack(ckl_ref)
    dz = torch.randn(Kk, B, L, generator=gen)
    gj, gc = torch.rand(B, generator=gen), torch.rand(M, B, generator=gen)
    ((z_ref * dz).sum() + (jkl_ref * gj).sum() + (ckl_ref * gc).sum()).backward()
    d = dev()
    mud = [t.to(d).requires_grad_() for t in mus_c]
    lvd = [t.to(d).requires_grad_() for t in lvs_c]
    mk = [m.to(d) for m in masks] if masked else None
    z, jkl, ckl, jmu_o, jlv_o = K.MVTCAEPosteriorFn.apply(eps.to(d), mk, *mud, *lvd)
    close(z, z_ref, what="z")
    close(jkl, jkl_ref, what="joint kl rows")
    close(ckl, ckl_ref, what="cond kl rows")
    close(jmu_o, jmu, what="joint mu")
    ((z * dz.to(d)).sum() + (jkl * gj.to(d)).sum() + (ckl * gc.to(d)).sum()).backward()
    for i in range(M):
        close(mud[i].grad, mur[i].grad, what=f"dmu {i}")
        g_ref = lvr[i].grad
        close(lvd[i].grad, g_ref, what=f"dlv {i}")
        if masked:  # exactly zero gradient on missing rows (tests/test_mvtcae.py:160-175)
            assert float(lvd[i].grad[~mk[i]].abs().max() if (~mk[i]).any() else 0.0) == 0.0
            assert float(mud[i].grad[~mk[i]].abs().max() if (~mk[i]).any() else 0.0) == 0.0


@pytest.mark.parametrize("family", ["normal", "laplace_with_softmax"])
def test_mmvae_std(K, family):
    from multivae_amd._lib import FAMILY

    gen = g(2)
    lv = torch.randn(9, 20, generator=gen)
    lr = lv.clone().requires_grad_()
    ref = elbo.mmvae_std(lr, family)
    gs = torch.randn(9, 20, generator=gen)
    (ref * gs).sum().backward()
    d = dev()
    ld = lv.to(d).requires_grad_()
    sd = K.MMVAEStdFn.apply(ld, FAMILY[family])
    close(sd, ref, what="std")
    (sd * gs.to(d)).sum().backward()
    close(ld.grad, lr.grad, what="dlv")


def test_adam_matches_oracle(K):
    gen = g(9)
    n = 100003
    p = torch.randn(n, generator=gen)
    m = torch.zeros(n)
    v = torch.zeros(n)
    d = dev()
    pd, md, vd = p.to(d), m.to(d), v.to(d)
    for step in range(1, 4):
        gr = torch.randn(n, generator=gen)
        train.adam_update(p, gr * 0.5, m, v, step, lr=1e-3)
        K.adam_step(pd, gr.to(d), md, vd, step, 1e-3, grad_scale=0.5)
    close(pd, p, rtol=1e-6, what="adam params")
    close(vd, v, rtol=1e-6, what="adam v")


@pytest.mark.parametrize("amsgrad", [False, True])
def test_adam_vector_path_and_fused_zero_grad(K, amsgrad):
    """mvk_adam_step_fused: the 16-byte path (n % 4 == 0, aligned buffers) against the scalar path (an unaligned view of the
    same data), and zero_grad clears the gradient as it is consumed."""
    gen = g(12)
    n = 4 * 25001
    d = dev()
    base = [torch.randn(n + 1, generator=gen).to(d) for _ in range(2)] + [torch.rand(n + 1, generator=gen).to(d) for _ in range(2)]
    gr = torch.randn(n + 1, generator=gen).to(d)
    vec = [t[:n].clone() for t in base]         # aligned, n % 4 == 0: float4 kernel
    sca = [t.clone()[1:] for t in base]         # 4 bytes off a 16-byte boundary: scalar kernel
    for t, s_ in zip(vec, sca):
        s_.copy_(t)
    gv, gs_ = gr[:n].clone(), gr.clone()[1:]
    gs_.copy_(gv)
    K.adam_step(vec[0], gv, vec[1], vec[2], 3, 1e-3, 0.9, 0.99, 1e-8, 0.01, grad_scale=0.5, vmax=vec[3] if amsgrad else None,
                zero_grad=True)
    K.adam_step(sca[0], gs_, sca[1], sca[2], 3, 1e-3, 0.9, 0.99, 1e-8, 0.01, grad_scale=0.5, vmax=sca[3] if amsgrad else None,
                zero_grad=False)
    for a_, b_, nm in zip(vec, sca, ("p", "m", "v", "vmax")):
        close(a_, b_, rtol=1e-6, what=nm)  # (not bit-equal: the two kernels contract / divide in a different order)
    assert float(gv.abs().max()) == 0.0 and torch.equal(gs_, gr[:n])


def test_adam_amsgrad_matches_torch(K):
    """mvk_adam_step_amsgrad against torch.optim.Adam(amsgrad=True) on the CPU (the reference's MMVAE+ setting,
    examples/mmvae_plus/mmnist.py:61-62) over 4 steps with weight decay and a changing learning rate."""
    gen = g(10)
    n = 50021
    p0 = torch.randn(n, generator=gen)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=2e-3, betas=(0.85, 0.97), eps=1e-7, weight_decay=0.01, amsgrad=True)
    d = dev()
    pd, md, vd, xd = p0.to(d), torch.zeros(n, device=d), torch.zeros(n, device=d), torch.zeros(n, device=d)
    for step in range(1, 5):
        gr = torch.randn(n, generator=gen) * (3.0 if step == 1 else 0.3)  # a large first step: the maximum matters
        lr = 2e-3 * 0.5 ** (step - 1)
        opt.param_groups[0]["lr"] = lr
        ref.grad = gr.clone()
        opt.step()
        K.adam_step(pd, gr.to(d), md, vd, step, lr, 0.85, 0.97, 1e-7, 0.01, vmax=xd)
    close(pd, ref.detach(), rtol=2e-6, what="amsgrad params")
    close(xd, opt.state[ref]["max_exp_avg_sq"], rtol=1e-6, what="max_exp_avg_sq")
    close(vd, opt.state[ref]["exp_avg_sq"], rtol=1e-6, what="exp_avg_sq")


@pytest.mark.parametrize("bwd_bf", ["0", "1"])
@pytest.mark.parametrize("n,h,w,Cu,Cv", [(3, 16, 16, 3, 32), (700, 16, 16, 3, 32), (5, 8, 8, 1, 16), (2, 16, 8, 4, 64)])
def test_small_up_fwd_bwd(K, n, h, w, Cu, Cv, bwd_bf, monkeypatch):
    """The per-image MFMA kernels of the image-producing layer (smallconv.hip) against torch CPU.  bwd_bf: the backward on the
    exact-fp32 matrix instructions / the split-bf16 kernel (default at the SVHN decoder's shape 16x16x3x32)."""
    from multivae_amd import _lib
    from multivae_amd._lib import call, ptr, stream_ptr

    if bwd_bf != "0" and (h, w, Cu, Cv) != (16, 16, 3, 32):
        pytest.skip("the split-bf16 backward covers the SVHN decoder's shape only")
    monkeypatch.setenv("MVK_SMALL_BWD_BF", bwd_bf)

    assert _lib.load().mvk_conv4s2_small_up_supported(h, w, Cu, Cv)
    gen = g(n + Cu)
    V = torch.relu(torch.randn(n, Cv, h, w, generator=gen))
    Wt = torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(4 * Cv)
    b = torch.randn(Cu, generator=gen)
    dout = torch.randn(n, Cu, 2 * h, 2 * w, generator=gen)
    Vr, Wr, br = V.clone().requires_grad_(), Wt.clone().requires_grad_(), b.clone().requires_grad_()
    ref = torch.sigmoid(F.conv_transpose2d(Vr, Wr, br, stride=2, padding=1))
    ref.backward(dout)
    d = dev()
    Vd, Wd, bd, dod = nhwc(V).to(d), Wt.to(d), b.to(d), dout.to(d)
    out = torch.empty(n, Cu, 2 * h, 2 * w, device=d)
    call("mvk_conv4s2_small_up_fwd", ptr(Vd), ptr(Wd), ptr(bd), ptr(out), n, h, w, Cu, Cv, 2, stream_ptr())
    close(out, ref, what="small up fwd")
    dV = torch.empty(n, h, w, Cv, device=d)
    dW = torch.zeros_like(Wd)
    db = torch.zeros_like(bd)
    ws = K._ws(Vd)
    dbv = torch.zeros(Cv, device=d)
    call("mvk_conv4s2_small_up_bwd", ptr(dod), ptr(out), 2, ptr(Vd), 1, ptr(Wd), ptr(dV), ptr(dW), ptr(db), ptr(dbv),
         ptr(ws), ws.numel(), n, h, w, Cu, Cv, stream_ptr())
    close(dbv, (Vr.grad * (V > 0).float()).sum(dim=(0, 2, 3)), what="small up channel sums of dV")
    close(nchw(dV.cpu()), Vr.grad * (V > 0).float(), what="small up dV (relu mask fused)")
    close(dW, Wr.grad, what="small up dW")
    close(db, br.grad, what="small up db")


@pytest.mark.parametrize("n,rowscale,spread", [(3, False, 0.0), (700, True, 0.0), (1030, True, 1.0), (1030, False, 2.0)])
def test_small_up_bwd_scaled_fp16(K, n, rowscale, spread):
    """The image layer's backward on scaled fp16 pairs (small_up_bwd_h_kernel behind mvk_conv4s2_small_up_bwd_pre_s: 3 MFMAs per
    product, the gradient image under the bound the fused tail publishes x max |rowscale|, V under its producer's bound) against
    float64 autograd of ConvTranspose2d(32, 3, 4, 2, 1) (reference: models/nn/svhn.py:58-60): dV with the ReLU mask of V fused, its
    channel sums, dW, db.  spread: images (and row weights) whose magnitudes differ by 10^+-spread — dV is checked per image on the
    image's own scale.  The same launch on bf16 pieces (mvk_conv4s2_small_up_bwd_pre_y) must agree."""
    from multivae_amd._lib import call, ptr, stream_ptr

    h = w = 16
    Cu, Cv = 3, 32
    gen = g(n + 29)
    V = torch.relu(torch.randn(n, Cv, h, w, generator=gen))
    dpre = torch.randn(n, Cu, 2 * h, 2 * w, generator=gen) * 0.05
    rs = None
    if spread:
        V = V * (10.0 ** ((torch.rand(n, 1, 1, 1, generator=gen) * 2 - 1) * spread))
        dpre = dpre * (10.0 ** ((torch.rand(n, 1, 1, 1, generator=gen) * 2 - 1) * spread))
    V[0, 0, 0, 0] = 1e-12 * float(V.max())  # positive, and its leading fp16 piece is zero: the mask must still pass it
    if rowscale:
        rs = torch.randn(n, generator=gen) * (10.0 ** ((torch.rand(n, generator=gen) * 2 - 1) * spread))
        rs[n // 2] = 0.0  # a row that does not enter the loss
    Wt = torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(4 * Cv)
    gimg = dpre.double() * (rs.double().view(n, 1, 1, 1) if rs is not None else 1.0)
    V64, W64 = V.double().requires_grad_(), Wt.double().requires_grad_()
    F.conv_transpose2d(V64, W64, None, stride=2, padding=1).backward(gimg)
    mask = (V > 0).double()
    dV_ref = V64.grad * mask
    d = dev()
    Vd, Wd, dpd = nhwc(V).to(d), Wt.to(d), dpre.to(d)
    rsd = rs.to(d) if rs is not None else None
    v_amax, du_amax = torch.zeros(1, device=d), torch.zeros(1, device=d)
    K.amax_of(Vd, v_amax)
    K.amax_of(dpd, du_amax)
    ws = K._ws(Vd)

    def run(entry, *bounds):
        dV = torch.empty(n, h, w, Cv, device=d)
        dW, db, dbv = torch.zeros_like(Wd), torch.zeros(Cu, device=d), torch.zeros(Cv, device=d)
        dv_amax = torch.zeros(1, device=d)
        call(entry, ptr(dpd), ptr(rsd), ptr(Vd), 1, ptr(Wd), ptr(dV), ptr(dW), ptr(db), ptr(dbv), ptr(ws), ws.numel(), n, h, w, Cu,
             Cv, *[ptr(b_) for b_ in bounds], ptr(dv_amax), stream_ptr())
        return dV, dW, db, dbv, dv_amax

    dV, dW, db, dbv, dv_amax = run("mvk_conv4s2_small_up_bwd_pre_s", du_amax, v_amax)
    close(nchw(dV.cpu()), dV_ref, rtol=3e-6, what="scaled small up dV (relu mask fused)")
    # per image: full precision for images down to 1e-4 of the largest one (the bound max |dpre| x max |rowscale| may sit 10^2
    # above the largest product; bf3.hpp: elements keep full precision down to 2^-28 of the bound)
    imax = dV_ref.abs().flatten(1).max(dim=1).values
    close_per_slice(nchw(dV.cpu()), dV_ref, imax >= 1e-4 * imax.max(), 3e-6, "scaled small up dV, per image")
    assert bool((nchw(dV.cpu())[mask == 0] == 0).all()), "the ReLU mask of V"
    assert float(dV[0, 0, 0, 0]) != 0.0 or float(dV_ref[0, 0, 0, 0]) == 0.0, "a tiny positive V lost its gradient"
    close(dW, W64.grad, rtol=3e-6, what="scaled small up dW")
    close(db, gimg.sum(dim=(0, 2, 3)), rtol=3e-6, what="scaled small up db")
    close(dbv, dV_ref.sum(dim=(0, 2, 3)), rtol=3e-6, what="scaled small up channel sums of dV")
    assert float(dv_amax) == float(dV.abs().max()), "published max |dV|"
    dV2, dW2, db2, dbv2, _ = run("mvk_conv4s2_small_up_bwd_pre_y")
    close(dV, dV2, rtol=3e-6, what="scaled vs bf16-piece dV")
    close(dW, dW2, rtol=3e-6, what="scaled vs bf16-piece dW")
    if n <= 8:  # bounds far above the data cost range, not precision; a zero gradient under a zero bound stays finite
        dV3, dW3, _, _, _ = run("mvk_conv4s2_small_up_bwd_pre_s", du_amax * 1000.0, v_amax * 1000.0)
        ref3 = dV_ref.clone()
        ref3[0, :, 0, 0] = nchw(dV3.cpu())[0, :, 0, 0].double()  # (the 1e-12 entry of V is 2^-50 of THIS bound: below what the mask resolves)
        close(nchw(dV3.cpu()), ref3, rtol=3e-6, what="scaled small up dV, loose bounds")
        close(dW3, W64.grad, rtol=3e-6, what="scaled small up dW, loose bounds")
        dpd.zero_()
        dV4, dW4, db4, _, a4 = run("mvk_conv4s2_small_up_bwd_pre_s", torch.zeros(1, device=d), v_amax)
        assert float(dV4.abs().max()) == 0.0 and float(dW4.abs().max()) == 0.0 and float(db4.abs().max()) == 0.0 and float(a4) == 0.0


@pytest.mark.parametrize("n,xrows,spread", [(3, 3, 0.0), (700, 70, 0.0), (1030, 103, 3.0)])
def test_small_up_fwd_scaled_fp16(K, n, xrows, spread):
    """The image-producing layer on scaled fp16 pairs (small_up_fwd_h_kernel: 3 MFMAs per product, weights as the A operand, the
    bound of V from an amax slot) against float64: the image, and the fused tail's Normal NLL row sums and d rows / d pre-activation
    (reference: models/nn/svhn.py:59-60, models/base/base_utils.py:62-87).  spread: images whose magnitudes differ by 10^+-spread —
    checked per image (a per-tensor scale must not cost the small images their precision)."""
    from multivae_amd._lib import call, ptr, stream_ptr

    h = w = 16
    Cu, Cv = 3, 32
    gen = g(n + 17)
    V = torch.relu(torch.randn(n, Cv, h, w, generator=gen))
    if spread:
        V = V * (10.0 ** ((torch.rand(n, 1, 1, 1, generator=gen) * 2 - 1) * spread))
    Wt = torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(4 * Cv)
    b = torch.randn(Cu, generator=gen)
    X = torch.rand(xrows, Cu, 2 * h, 2 * w, generator=gen)
    scale, gw = 0.75, 0.3
    pre64 = F.conv_transpose2d(V.double(), Wt.double(), b.double(), stride=2, padding=1)
    ref = torch.sigmoid(pre64)
    d = dev()
    Vd, Wd, bd, Xd = nhwc(V).to(d), Wt.to(d), b.to(d), X.to(d)
    amax = torch.zeros(1, device=d)
    K.amax_of(Vd, amax)
    out = torch.empty(n, Cu, 2 * h, 2 * w, device=d)
    call("mvk_conv4s2_small_up_fwd_s", ptr(Vd), ptr(Wd), ptr(bd), ptr(out), n, h, w, Cu, Cv, 2, ptr(amax), stream_ptr())
    close(out, ref, what="scaled small up fwd")
    # per image, on the pre-activations without the bias (act = none): every image within 3e-6 of ITS OWN largest entry
    raw = torch.empty_like(out)
    call("mvk_conv4s2_small_up_fwd_s", ptr(Vd), ptr(Wd), ptr(None), ptr(raw), n, h, w, Cu, Cv, 0, ptr(amax), stream_ptr())
    raw64 = pre64 - b.double().view(1, Cu, 1, 1)
    err = (raw.double().cpu() - raw64).abs().flatten(1).max(dim=1).values
    den = raw64.abs().flatten(1).max(dim=1).values.clamp_min(1e-30)
    assert float((err / den).max()) <= 3e-6, f"per-image rel err {float((err / den).max()):.3e}"
    # fused tail: rows = sum (r - x)^2 / (2 s^2) + D (log s + 1/2 log 2 pi), dpre = gw (r - x) / s^2 r (1 - r)
    xs = X.double()[torch.arange(n) % xrows]
    rows_ref = ((ref - xs) ** 2).flatten(1).sum(1) / (2 * scale * scale) + Cu * 4 * h * w * (math.log(scale) + 0.5 * math.log(2 * math.pi))
    dpre_ref = gw * (ref - xs) / (scale * scale) * ref * (1 - ref)
    dpre, rows = torch.empty_like(out), torch.empty(n, device=d)
    call("mvk_conv4s2_small_up_fwd_nll_s", ptr(Vd), ptr(Wd), ptr(bd), ptr(Xd), xrows, scale, gw, ptr(dpre), ptr(rows), n, h, w, Cu, Cv, 2,
         ptr(amax), stream_ptr())
    close(rows, rows_ref, rtol=2e-6, what="scaled fused tail rows")
    # (with spread, an entry of a LARGE image whose sigmoid is not saturated carries that image's absolute pre-activation error,
    # 4e-7 of the image's maximum, through a slope of order one: the per-image check above is the precision statement there)
    close(dpre, dpre_ref, rtol=3e-6 if not spread else 2e-4, what="scaled fused tail dpre")
    # and the unscaled fused tail agrees (same sums, other product form)
    dpre2, rows2 = torch.empty_like(out), torch.empty(n, device=d)
    call("mvk_conv4s2_small_up_fwd_nll_w", ptr(Vd), ptr(Wd), ptr(bd), ptr(Xd), xrows, scale, gw, ptr(dpre2), ptr(rows2), n, h, w, Cu, Cv, 2,
         stream_ptr())
    close(rows, rows2, rtol=2e-6, what="scaled vs bf16-piece tail rows")
    # ... and publishes a bound of the gradient it stored (what the layer's backward scales by), same numbers
    dpre3, rows3, pub = torch.empty_like(out), torch.empty(n, device=d), torch.zeros(1, device=d)
    call("mvk_conv4s2_small_up_fwd_nll_sy", ptr(Vd), ptr(Wd), ptr(bd), ptr(Xd), xrows, scale, gw, ptr(dpre3), ptr(rows3), n, h, w, Cu, Cv,
         2, ptr(amax), ptr(pub), stream_ptr())
    assert torch.equal(dpre3, dpre) and torch.equal(rows3, rows)
    assert float(dpre.abs().max()) <= float(pub) <= 64.0 * float(dpre.abs().max())  # a bound from the largest row sum (<= sqrt(3072) x)
    if n <= 8:  # edge values: an all-zero map under a zero bound (scale clamps, no NaN), then a bound far above the data
        Z, zero = torch.zeros_like(Vd), torch.zeros(1, device=d)
        call("mvk_conv4s2_small_up_fwd_s", ptr(Z), ptr(Wd), ptr(bd), ptr(out), n, h, w, Cu, Cv, 2, ptr(zero), stream_ptr())
        close(out, torch.sigmoid(b.double()).view(1, Cu, 1, 1).expand(n, Cu, 2 * h, 2 * w), what="scaled small up fwd, zero map")
        loose = amax * 1000.0  # a 1000x loose bound costs range, not precision
        call("mvk_conv4s2_small_up_fwd_s", ptr(Vd), ptr(Wd), ptr(bd), ptr(out), n, h, w, Cu, Cv, 2, ptr(loose), stream_ptr())
        close(out, ref, what="scaled small up fwd, loose bound")


@pytest.mark.parametrize("M,N,Kd", [(40960, 128, 1024), (4096, 64, 2048), (65536, 32, 512), (8192, 64, 96)])
def test_split_bf16_engine_has_fp32_accuracy(K, M, N, Kd):
    """The default GEMM engine (igemm_bf.hpp: every fp32 operand split into 3 bf16 pieces, 6 bf16 MFMAs per product)
    must be as accurate as an fp32 GEMM: error against an fp64 product no worse than 1.5x torch's fp32 matmul
    (rocBLAS), in max and rms norm."""
    torch.manual_seed(5)
    x = torch.randn(M, Kd, device=dev()) * torch.rand(M, 1, device=dev()) * 4
    w = torch.randn(N, Kd, device=dev()) / Kd ** 0.5
    ref = x.double() @ w.double().t()
    y = K.linear_fwd(x, w, None, 0)
    yt = x @ w.t()
    den = ref.abs().max().item()
    e_mvk = (y.double() - ref).abs().max().item() / den
    e_t = (yt.double() - ref).abs().max().item() / den
    r_mvk = (y.double() - ref).pow(2).mean().sqrt().item()
    r_t = (yt.double() - ref).pow(2).mean().sqrt().item()
    assert e_mvk <= 1.5 * e_t + 1e-7, (e_mvk, e_t)
    assert r_mvk <= 1.5 * r_t + 1e-9, (r_mvk, r_t)


@pytest.mark.parametrize("n,h,w,Cu,Cv", [(96, 4, 4, 64, 128), (40, 8, 8, 32, 64)])
def test_conv_reads_pre_split_input(K, n, h, w, Cu, Cv):
    """MVK_FMT_IN_BF3: a convolution whose gathered input is stored pre-split (three bf16 planes, mvk_f32_to_bf3)
    computes what the launch computes from the fp32 tensor (same arithmetic when both run on the split engine; small
    problems take the exact-fp32 engine for fp32 inputs, hence a tolerance), and the planes reconstruct the values."""
    torch.manual_seed(11)
    wt = torch.randn(Cv, Cu, 4, 4, device=dev()) * 0.05
    wd, wu = K.pack_conv(wt)
    V = torch.relu(torch.randn(n, h, w, Cv, device=dev()))
    U = torch.randn(n, 2 * h, 2 * w, Cu, device=dev())
    V3, U3 = K.to_bf3(V), K.to_bf3(U)
    assert V3.shape == (3,) + tuple(V.shape) and V3.dtype == torch.bfloat16
    back = V3.float().sum(0)
    assert float((back - V).abs().max()) <= 2e-7 * float(V.abs().max())
    up_ref = K.conv_up(V, wu, None, n, h, w, Cu, Cv, 1)
    up_pre = K.conv_up(V3, wu, None, n, h, w, Cu, Cv, 1, in_bf3=True)
    close(up_pre, up_ref, 2e-6, "up from pre-split input")
    dn_ref = K.conv_down(U, wd, None, n, h, w, Cu, Cv, 0, v_act_src=V, v_act=1)
    dn_pre = K.conv_down(U3, wd, None, n, h, w, Cu, Cv, 0, v_act_src=V, v_act=1, in_bf3=True)
    close(dn_pre, dn_ref, 2e-6, "down from pre-split input")


@pytest.mark.parametrize("M,B,L,Kk", [(2, 9, 20, 1), (3, 5, 70, 4)])
def test_jmvae_posterior(K, M, B, L, Kk):
    """mvk_jmvae_posterior_fwd/bwd vs the formulas of jmvae_model.py:133-174 in torch (L > 64 and K > 1 included)."""
    gen = g(21)
    jmu = torch.randn(B, L, generator=gen).requires_grad_(True)
    jlv = (0.3 * torch.randn(B, L, generator=gen)).requires_grad_(True)
    mus = [torch.randn(B, L, generator=gen).requires_grad_(True) for _ in range(M)]
    lvs = [(0.3 * torch.randn(B, L, generator=gen)).requires_grad_(True) for _ in range(M)]
    eps = torch.randn(Kk, B, L, generator=gen)
    z_ref = jmu + torch.exp(0.5 * jlv) * eps
    kld_ref = (-0.5 * (1 + jlv - jmu.pow(2) - jlv.exp())).sum(-1)
    ljm_ref = sum(0.5 * (lvs[m] - jlv + (jlv.exp() + (jmu - mus[m]) ** 2) / lvs[m].exp() - 1) for m in range(M)).sum(-1)
    wz, wk, wj = torch.randn(Kk, B, L, generator=gen), torch.randn(B, generator=gen), torch.randn(B, generator=gen)
    ((z_ref * wz).sum() + (kld_ref * wk).sum() + (ljm_ref * wj).sum()).backward()
    d = dev()
    leaf = lambda t: t.detach().to(d).requires_grad_(True)
    djmu, djlv = leaf(jmu), leaf(jlv)
    dmus, dlvs = [leaf(t) for t in mus], [leaf(t) for t in lvs]
    z, kld, ljm = K.JMVAEPosteriorFn.apply(eps.to(d), djmu, djlv, *dmus, *dlvs)
    close(z, z_ref, what="z")
    close(kld, kld_ref, what="kld rows")
    close(ljm, ljm_ref, what="ljm rows")
    ((z * wz.to(d)).sum() + (kld * wk.to(d)).sum() + (ljm * wj.to(d)).sum()).backward()
    close(djmu.grad, jmu.grad, what="d joint mu")
    close(djlv.grad, jlv.grad, what="d joint lv")
    for m in range(M):
        close(dmus[m].grad, mus[m].grad, what=f"d mu[{m}]")
        close(dlvs[m].grad, lvs[m].grad, what=f"d lv[{m}]")


@pytest.mark.parametrize("n,H,W,Cin,Cout", [(3, 7, 7, 64, 128), (2, 14, 14, 128, 64), (5, 28, 28, 3, 64), (2, 28, 28, 64, 3),
                                          (40, 16, 16, 64, 64), (3, 64, 64, 3, 64), (3, 64, 64, 64, 3), (2, 13, 11, 3, 32),
                                          (2, 13, 19, 32, 3), (1, 9, 40, 1, 16), (1, 9, 40, 16, 1),
                                          # few positions, long reductions: the split-K route of the tiled engine (round 5)
                                          (32, 7, 7, 128, 128), (8, 7, 7, 256, 128), (32, 7, 7, 128, 256)])
def test_conv3x3_fwd_bwd(K, n, H, W, Cin, Cout):
    """mvk_conv3x3 / mvk_conv3x3_wgrad (+ kind-2 weight pack) vs F.conv2d(3, 1, 1) + LeakyReLU(0.2): forward, backward
    data with the fused activation derivative and bias-gradient column sums, backward weight in the reference layout."""
    gen = g(31)
    x = torch.randn(n, Cin, H, W, generator=gen)
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) / (3 * Cin ** 0.5)).requires_grad_(True)
    b = (0.1 * torch.randn(Cout, generator=gen)).requires_grad_(True)
    xin = F.leaky_relu(x, 0.2).requires_grad_(True)  # x itself is a LeakyReLU output: its derivative multiplies dx
    y = F.leaky_relu(F.conv2d(xin, w, b, 1, 1), 0.2)
    dy = torch.randn(n, Cout, H, W, generator=gen)
    y.backward(dy)
    dxin_pre = xin.grad * torch.where(xin > 0, 1.0, 0.2)  # gradient w.r.t. the pre-activation of xin
    d = dev()
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(d)
    wd = w.detach().to(d)
    (wf, wb), = K.pack_weights([(wd, "c3", True, True)])
    X = nhwc(xin)
    Y = K.conv3x3(X, wf, b.detach().to(d), n, H, W, Cin, Cout, act=K.LEAKY)
    close(Y, nhwc(y), what="conv3x3 forward")
    dpre = nhwc(dy) * torch.where(Y > 0, 1.0, 0.2)
    bparam = torch.zeros(Cin, device=d).requires_grad_(True)  # stands for the bias of the layer that produced xin
    bparam.grad = torch.zeros(Cin, device=d)
    dX, _ = K.conv3x3(dpre, wb, None, n, H, W, Cout, Cin, y_act_src=X, y_src_act=K.LEAKY, out_bias=bparam)
    close(dX, nhwc(dxin_pre), what="conv3x3 backward data")
    close(bparam.grad, dxin_pre.sum((0, 2, 3)), what="fused bias-gradient column sums")
    wparam = wd.clone().requires_grad_(True)
    wparam.grad = torch.zeros_like(wparam)
    K.conv3x3_wgrad(X, dpre, wparam, n, H, W, Cin, Cout)
    close(wparam.grad, w.grad, what="conv3x3 backward weight")


@pytest.mark.parametrize("n,H,W,Cin,Cout", [(3, 64, 64, 3, 64), (7, 28, 28, 3, 64), (2, 9, 13, 1, 32), (5, 28, 28, 4, 128)])
def test_conv3x3_image_side_publishes_its_maximum(K, n, H, W, Cin, Cout):
    """mvk_conv3x3_y: the direct kernel for an image (Cin <= 4) on the input side (conv_img of the ResNet encoders, the backward
    data of the decoders' conv_img) writes what mvk_conv3x3 writes — bit for bit, with and without the mask of a landing
    activation — and publishes max |Y| into a zeroed slot (amax protocol): exactly the maximum of its result, ragged bands and
    inactive threads included; a slot that already holds a larger bound keeps it."""
    gen = g(57)
    d = dev()
    x = torch.randn(n, H, W, Cin, generator=gen).to(d)
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) / (3 * Cin ** 0.5)).to(d)
    b = (0.1 * torch.randn(Cout, generator=gen)).to(d)
    src = torch.randn(n, H, W, Cout, generator=gen).to(d)
    (wf, _), = K.pack_weights([(w, "c3", True, True)])
    for kw in (dict(act=K.LEAKY), dict(y_act_src=src, y_src_act=K.LEAKY)):
        ref = K.conv3x3(x, wf, b, n, H, W, Cin, Cout, **kw)
        slot = torch.zeros(1, device=d)
        got = K.conv3x3(x, wf, b, n, H, W, Cin, Cout, y_amax=slot, **kw)
        assert torch.equal(got, ref)
        assert float(slot) == float(ref.abs().max())
        big = torch.full((1,), 1e6, device=d)
        K.conv3x3(x, wf, b, n, H, W, Cin, Cout, y_amax=big, **kw)
        assert float(big) == 1e6
    y64 = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), b.double().cpu(), 1, 1), 0.2)
    close(K.conv3x3(x, wf, b, n, H, W, Cin, Cout, act=K.LEAKY, y_amax=torch.zeros(1, device=d)), y64.permute(0, 2, 3, 1), what="conv3x3_y vs float64")


@pytest.mark.parametrize("n,H,W,Cin,Cout", [(5, 64, 64, 64, 64), (9, 32, 32, 64, 128), (40, 16, 16, 128, 128), (33, 16, 16, 128, 256),
                                          (70, 28, 28, 64, 64), (130, 14, 14, 128, 64), (300, 7, 7, 128, 128), (2, 7, 7, 128, 256),
                                          (1, 9, 13, 64, 64), (3, 32, 32, 128, 64), (20, 16, 16, 256, 128), (2, 14, 14, 256, 256)])
def test_conv3x3_register_stationary(K, n, H, W, Cin, Cout):
    """csrc/conv3rs.hip (weights resident in registers, the images one zero-padded stream of positions through an LDS ring)
    against a float64 F.conv2d(3, 1, 1): plain forward with bias + LeakyReLU; backward-data form (activation derivative of
    the landing layer + bias-gradient column sums); the residual epilogue.  Taken for EVERY size here (debug flag 0x800),
    so workers with zero, one and two tiles and ragged last tiles are covered; (3, 32, 32, 128, 64) exceeds the LDS ring
    and must fall back to the tiled engine with the same result."""
    gen = g(53)
    d = dev()
    x = torch.randn(n, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (3 * Cin ** 0.5)
    b = 0.1 * torch.randn(Cout, generator=gen)
    res = torch.randn(n, Cout, H, W, generator=gen)
    src = torch.randn(n, Cout, H, W, generator=gen)
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(d)
    (wf, _), = K.pack_weights([(w.to(d), "c3", True, True)])
    conv = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    lrelu = lambda t: F.leaky_relu(t, 0.2)
    mask = torch.where(src > 0, 1.0, 0.2).double()
    _debug_flags(0x800)
    try:
        Y = K.conv3x3(nhwc(x), wf, b.to(d), n, H, W, Cin, Cout, act=K.LEAKY)
        close(Y, nhwc(lrelu(conv).float()), what="forward + bias + LeakyReLU", rtol=3e-6)
        bparam = torch.zeros(Cout, device=d).requires_grad_(True)
        bparam.grad = torch.zeros(Cout, device=d)
        Y, _ = K.conv3x3(nhwc(x), wf, None, n, H, W, Cin, Cout, y_act_src=nhwc(src), y_src_act=K.LEAKY, out_bias=bparam)
        ref = (conv - b.double().view(1, -1, 1, 1)) * mask
        close(Y, nhwc(ref.float()), what="conv * act'(mask source)", rtol=3e-6)
        close(bparam.grad, ref.sum((0, 2, 3)).float(), what="column sums", rtol=1e-5)
        Y = K.conv3x3(nhwc(x), wf, b.to(d), n, H, W, Cin, Cout, act=K.NONE, res=nhwc(res), res_alpha=0.1)
        close(Y, nhwc((res.double() + 0.1 * conv).float()), what="res + 0.1 * conv", rtol=3e-6)
        Y = K.conv3x3(nhwc(x), wf, b.to(d), n, H, W, Cin, Cout, act=K.RELU, y_act_src=nhwc(src), y_src_act=K.RELU, res=nhwc(res))
        close(Y, nhwc((res.double() + torch.relu(conv) * (src > 0)).float()), what="res + relu(conv) * relu'(src)", rtol=2e-6)
        # weight gradient (output-stationary c3wg_kernel): dW += sum_p dY[p] X[p + tap], twice into the same buffer
        w64 = w.double().requires_grad_(True)
        F.conv2d(x.double(), w64, None, 1, 1).backward(src.double())
        wparam = w.to(d).clone().requires_grad_(True)
        wparam.grad = torch.zeros_like(wparam)
        K.conv3x3_wgrad(nhwc(x), nhwc(src), wparam, n, H, W, Cin, Cout)
        close(wparam.grad, w64.grad.float(), what="weight gradient", rtol=3e-6)
        K.conv3x3_wgrad(nhwc(x), nhwc(src), wparam, n, H, W, Cin, Cout)
        close(wparam.grad, 2 * w64.grad.float(), what="weight gradient accumulates", rtol=3e-6)
    finally:
        _debug_flags(0)
    Yt = K.conv3x3(nhwc(x), wf, b.to(d), n, H, W, Cin, Cout, act=K.LEAKY)  # the default dispatch (tiled engine at these sizes)
    close(Yt, nhwc(lrelu(conv).float()), what="default dispatch", rtol=3e-6)


@pytest.mark.parametrize("n,H,W,Cout", [(40, 7, 7, 128), (9, 16, 16, 128), (3, 16, 16, 256), (5, 28, 20, 64)])
@pytest.mark.parametrize("spread", [0.0, 2.0])
def test_conv3x3_256_input_channels_as_two_slices(K, n, H, W, Cout, spread):
    """A 3x3 layer with 256 input channels on the register-stationary kernels: two launches over 128-channel slices
    (mvk_conv3x3_s_part; kernels.conv3x3_s_split), the second adding the first one's partial sum in front of its bias, activation and
    mask.  Against float64 at the tolerance of the one-launch layers, in the forward form (input activation, bias, LeakyReLU) and
    the backward-data form (0.1 conv x act'(mask source), column sums); max |Y| as published is exact; the tiled engine agrees."""
    Cin = 256
    gen = g(77)
    d = dev()
    x = torch.randn(n, Cin, H, W, generator=gen) * torch.exp(spread * torch.randn(n, 1, 1, 1, generator=gen)) * 2.1
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (3 * Cin ** 0.5) * torch.exp(spread * torch.randn(Cout, 1, 1, 1, generator=gen))
    b = 0.1 * torch.randn(Cout, generator=gen)
    src = torch.randn(n, Cout, H, W, generator=gen)
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(d)
    (wf, _), = K.pack_weights([(w.to(d), "c3", True, False)])
    pool = K.AmaxPool(wf, 8)
    X = nhwc(x)
    xam = K.amax_of(X, pool.take())
    ax = F.leaky_relu(x.double(), 0.2)
    conv = F.conv2d(ax, w.double(), None, 1, 1)
    bb = b.double().view(1, -1, 1, 1)
    _debug_flags(0x800)
    try:
        assert K.conv3x3_scaled_ok(n, H, W, 128, Cout) and not K.conv3x3_scaled_ok(n, H, W, Cin, Cout)
        yam = pool.take()
        Y = K.conv3x3_s_split(X, wf, b.to(d), n, H, W, Cin, Cout, xam, wf.mvk_amax, yam, act=K.LEAKY, x_act=K.LEAKY)
        close(Y, nhwc(F.leaky_relu(conv + bb, 0.2).float()), what="lrelu(conv(lrelu(x)) + b), 256 channels in two slices", rtol=3e-6)
        assert float(yam) == float(Y.abs().max()), "published max |Y|"
        bparam = torch.zeros(Cout, device=d).requires_grad_(True)
        bparam.grad = torch.zeros(Cout, device=d)
        mask = torch.where(src > 0, 1.0, 0.2).double()
        yam = pool.take()
        Y, _ = K.conv3x3_s_split(X, wf, None, n, H, W, Cin, Cout, xam, wf.mvk_amax, yam, y_act_src=nhwc(src), y_src_act=K.LEAKY,
                                 out_bias=bparam, x_act=K.LEAKY, pre_scale=0.1)
        ref = 0.1 * conv * mask
        close(Y, nhwc(ref.float()), what="0.1 conv * act'(mask source), two slices", rtol=3e-6)
        close(bparam.grad, ref.sum((0, 2, 3)).float(), what="column sums", rtol=1e-5)
        assert float(yam) == float(Y.abs().max())
        # the stack's dispatcher takes the same route and returns the published bound; the tiled engine agrees
        Y1, am1 = K._rs_conv(pool, X, xam, wf, b.to(d), n, H, W, Cin, Cout, act=K.LEAKY)
        plain = F.conv2d(x.double(), w.double(), None, 1, 1)
        close(Y1, nhwc(F.leaky_relu(plain + bb, 0.2).float()), what="_rs_conv, 256 input channels", rtol=3e-6)
        assert am1 is not None and float(am1) == float(Y1.abs().max())
        Y2 = K.conv3x3(X, wf, b.to(d), n, H, W, Cin, Cout, act=K.LEAKY)
        close(Y1, Y2, what="two slices vs tiled engine", rtol=3e-6)
    finally:
        _debug_flags(0)


@pytest.mark.parametrize("n,H,W,Cin,Cout", [(5, 64, 64, 64, 64), (9, 32, 32, 64, 128), (40, 16, 16, 128, 128), (33, 16, 16, 128, 256),
                                          (70, 28, 28, 64, 64), (130, 14, 14, 128, 64), (2, 7, 7, 128, 256), (1, 9, 13, 64, 64),
                                          (3, 32, 32, 128, 64), (2, 62, 50, 128, 128)])
@pytest.mark.parametrize("spread", [0.0, 3.0])
def test_conv3x3_scaled_fp16(K, n, H, W, Cin, Cout, spread):
    """mvk_conv3x3_s (csrc/conv3rs.hip, NP = 2): every product is 3 fp16 MFMAs on scaled (hi, lo) pairs instead of 6 bf16
    ones.  Same float64 reference and the SAME tolerance as the bf16-piece kernels (test_conv3x3_register_stationary), on
    unit-scale data and on data whose images / weight rows spread over e^(+-3 sigma) ~ 8 orders of magnitude; the operand
    scales come from mvk_amax / the pack launch, and max |Y| as published by the launch is exact."""
    gen = g(61)
    d = dev()
    x = torch.randn(n, Cin, H, W, generator=gen) * torch.exp(spread * torch.randn(n, 1, 1, 1, generator=gen)) * 3.7
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (3 * Cin ** 0.5) * torch.exp(spread * torch.randn(Cout, 1, 1, 1, generator=gen))
    b = 0.1 * torch.randn(Cout, generator=gen)
    res = torch.randn(n, Cout, H, W, generator=gen)
    src = torch.randn(n, Cout, H, W, generator=gen)
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(d)
    (wf, wb), = K.pack_weights([(w.to(d), "c3", True, True)])
    assert float(wf.mvk_amax) == float(w.abs().max()) and wb.mvk_amax.data_ptr() == wf.mvk_amax.data_ptr()
    pool = K.AmaxPool(wf, 8)
    X = nhwc(x)
    xam = K.amax_of(X, pool.take())
    assert float(xam) == float(x.abs().max())
    ax = F.leaky_relu(x.double(), 0.2)
    conv = F.conv2d(ax, w.double(), None, 1, 1)
    bb = b.double().view(1, -1, 1, 1)
    _debug_flags(0x800)
    try:
        assert K.conv3x3_scaled_ok(n, H, W, Cin, Cout)
        yam = pool.take()
        Y = K.conv3x3_s(X, wf, b.to(d), n, H, W, Cin, Cout, xam, wf.mvk_amax, yam, act=K.LEAKY, x_act=K.LEAKY)
        close(Y, nhwc(F.leaky_relu(conv + bb, 0.2).float()), what="lrelu(conv(lrelu(x)) + b)", rtol=3e-6)
        assert float(yam) == float(Y.abs().max()), "published max |Y|"
        bparam = torch.zeros(Cout, device=d).requires_grad_(True)
        bparam.grad = torch.zeros(Cout, device=d)
        mask = torch.where(src > 0, 1.0, 0.2).double()
        yam = pool.take()
        Y, _ = K.conv3x3_s(X, wf, None, n, H, W, Cin, Cout, xam, wf.mvk_amax, yam, y_act_src=nhwc(src), y_src_act=K.LEAKY,
                           out_bias=bparam, x_act=K.LEAKY, pre_scale=0.1)
        ref = 0.1 * conv * mask
        close(Y, nhwc(ref.float()), what="0.1 conv * act'(mask source)", rtol=3e-6)
        close(bparam.grad, ref.sum((0, 2, 3)).float(), what="column sums", rtol=1e-5)
        assert float(yam) == float(Y.abs().max())
        plain = F.conv2d(x.double(), w.double(), None, 1, 1)
        Y = K.conv3x3_s(X, wf, b.to(d), n, H, W, Cin, Cout, xam, wf.mvk_amax, act=K.NONE, res=nhwc(res), res_alpha=0.1)
        close(Y, nhwc((res.double() + 0.1 * (plain + bb)).float()), what="res + 0.1 * conv", rtol=3e-6)
        # the residual form with a second store (mvk_conv3x3_s2): Y = res + 0.1 a and a = lrelu(conv + b) from one launch — `a` is
        # bit for bit what the plain launch stores, max |Y| is published
        yam = pool.take()
        Y, A = K.conv3x3_s2(X, wf, b.to(d), n, H, W, Cin, Cout, xam, wf.mvk_amax, yam, K.LEAKY, nhwc(res), 0.1)
        A1 = K.conv3x3_s(X, wf, b.to(d), n, H, W, Cin, Cout, xam, wf.mvk_amax, act=K.LEAKY)
        assert torch.equal(A, A1), "the second store of mvk_conv3x3_s2"
        close(Y, nhwc((res.double() + 0.1 * F.leaky_relu(plain + bb, 0.2)).float()), what="res + 0.1 * lrelu(conv + b)", rtol=3e-6)
        assert float(yam) == float(Y.abs().max())
        yam = pool.take()
        Y = K.conv3x3_s(X, wf, b.to(d), n, H, W, Cin, Cout, xam, wf.mvk_amax, yam, act=K.RELU, y_act_src=nhwc(src),
                        y_src_act=K.RELU, res=nhwc(res))
        close(Y, nhwc((res.double() + torch.relu(plain + bb) * (src > 0)).float()), what="res + relu(conv) * relu'(src)", rtol=3e-6)
        assert float(yam) == float(Y.abs().max())
        # a bound that is merely an upper bound (here 1000 x too large) costs range, not precision
        loose = (xam * 1000.0).contiguous()
        Y = K.conv3x3_s(X, wf, None, n, H, W, Cin, Cout, loose, wf.mvk_amax)
        close(Y, nhwc(plain.float()), what="loose bound", rtol=3e-6)
        # (image, output channel) planes on their own scale: images and weight rows down to 2^-20 of the largest
        Y = K.conv3x3_s(X, wf, None, n, H, W, Cin, Cout, xam, wf.mvk_amax)
        sx, sw = x.abs().amax((1, 2, 3)), w.abs().amax((1, 2, 3))
        keep = (sx >= sx.max() * 2.0 ** -20)[:, None] & (sw >= sw.max() * 2.0 ** -20)[None, :]
        close_per_slice(Y.permute(0, 3, 1, 2), plain, keep, 1e-5, "conv3x3_s per (image, channel)")
        # weight gradient (c3wg_kernel<2>: one accumulator per tap tile, dY carries the 2^11 in a third piece) with the
        # activation of X applied while staging, the 0.1 of the residual branch and the bias gradient in the launch
        if K.conv3x3_wgrad_scaled_ok(n, H, W, Cin, Cout):
            dy = src * torch.exp(spread * torch.randn(n, 1, 1, 1, generator=g(62)))
            w64 = w.double().requires_grad_(True)
            F.conv2d(ax, w64, None, 1, 1).backward(0.1 * dy.double())
            wparam = w.to(d).clone().requires_grad_(True)
            wparam.grad = torch.zeros_like(wparam)
            bp = torch.zeros(Cout, device=d).requires_grad_(True)
            bp.grad = torch.zeros(Cout, device=d)
            dY = nhwc(dy)
            dyam = K.amax_of(dY, pool.take())
            K.conv3x3_wgrad_s(X, dY, wparam, bp, n, H, W, Cin, Cout, xam, dyam, x_act=K.LEAKY, dy_scale=0.1)
            close(wparam.grad, w64.grad.float(), what="weight gradient", rtol=3e-6)
            close(bp.grad, 0.1 * dy.double().sum((0, 2, 3)).float(), what="bias gradient", rtol=1e-5)
            K.conv3x3_wgrad_s(X, dY, wparam, None, n, H, W, Cin, Cout, (xam * 64.0).contiguous(), (dyam * 3.0).contiguous(),
                              x_act=K.LEAKY, dy_scale=0.1)
            close(wparam.grad, 2 * w64.grad.float(), what="weight gradient accumulates (loose bounds)", rtol=3e-6)
    finally:
        _debug_flags(0)


def test_scaled_fp16_edge_values(K):
    """The scaled-fp16 launches on degenerate operands: an all-zero tensor (bound 0: the scale clamps, the result is the bias),
    a NaN (it does not enter the bound — fmax drops it — but reaches every output whose window holds it, and no other)."""
    gen = g(71)
    d = dev()
    n, H, W, Cin, Cout = 3, 16, 16, 64, 64
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (3 * Cin ** 0.5)
    b = torch.randn(Cout, generator=gen)
    (wf, _), = K.pack_weights([(w.to(d), "c3", True, True)])
    pool = K.AmaxPool(wf, 8)
    _debug_flags(0x800)
    try:
        X = torch.zeros(n, H, W, Cin, device=d)
        xam = K.amax_of(X, pool.take())
        assert float(xam) == 0.0
        Y = K.conv3x3_s(X, wf, b.to(d), n, H, W, Cin, Cout, xam, wf.mvk_amax, pool.take())
        assert torch.equal(Y.cpu(), b.view(1, 1, 1, -1).expand(n, H, W, Cout))
        x = torch.randn(n, Cin, H, W, generator=gen)
        X = x.permute(0, 2, 3, 1).contiguous().to(d)
        X[1, 5, 7, 3] = float("nan")
        xam = K.amax_of(X, pool.take())
        assert float(xam) == float(x.abs().max()) or float(xam) < float(x.abs().max())  # the NaN is not the maximum
        yam = pool.take()
        Y = K.conv3x3_s(X, wf, None, n, H, W, Cin, Cout, xam, wf.mvk_amax, yam).cpu()
        bad = torch.isnan(Y).any(-1)
        want = torch.zeros(n, H, W, dtype=torch.bool)
        want[1, 4:7, 6:9] = True
        assert torch.equal(bad, want)
        ref = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
        ok = ~want
        assert float((Y[ok].double() - ref[ok]).abs().max() / ref[ok].abs().max()) < 3e-6
    finally:
        _debug_flags(0)


def test_scaled_entries_refuse_what_they_do_not_cover(K):
    """The scaled-fp16 entry points never fall back silently: a shape / batch outside mvk_*_scaled_ok, or one operand bound
    without the other, is MVK_EINVAL (the Python stub raises)."""
    from multivae_amd import _lib

    d = dev()
    pool = K.AmaxPool(torch.zeros(1, device=d), 8)
    one = torch.ones(1, device=d)
    # 3x3: too small for the default dispatch (no debug flag), and a 256-channel input
    n, H, Cin, Cout = 2, 8, 64, 64
    x = torch.randn(n, H, H, Cin, device=d)
    w = torch.randn(Cout, Cin, 3, 3, device=d) * 0.05
    (wf, _), = K.pack_weights([(w, "c3", True, True)])
    assert not K.conv3x3_scaled_ok(n, H, H, Cin, Cout)
    with pytest.raises(_lib.MvkError):
        K.conv3x3_s(x, wf, None, n, H, H, Cin, Cout, one, wf.mvk_amax, pool.take())
    _debug_flags(0x800)
    try:
        assert not K.conv3x3_scaled_ok(n, H, H, 256, 64)
        with pytest.raises(_lib.MvkError):  # one bound without the other
            K.conv3x3_s(x, wf, None, n, H, H, Cin, Cout, one, None, pool.take())
        wpar = w.clone().requires_grad_(True)
        wpar.grad = torch.zeros_like(wpar)
        with pytest.raises(_lib.MvkError):
            K.conv3x3_wgrad_s(x, x, wpar, None, n, H, H, Cin, Cout, one, None)
    finally:
        _debug_flags(0)
    # 4x4 / stride 2: a layer pair without a register-stationary kernel, and a batch below the threshold
    assert not K.conv4s2_scaled_ok(5120, 8, 8, 16, 32)
    assert not K.conv4s2_scaled_ok(8, 8, 8, 32, 64)
    Wc = torch.randn(64, 32, 4, 4, device=d) * 0.05
    wd, wu = K.pack_conv(Wc)
    U = torch.randn(8, 16, 16, 32, device=d)
    with pytest.raises(_lib.MvkError):
        K.conv_down(U, wd, None, 8, 8, 8, 32, 64, amax=(one, wd.mvk_amax, pool.take()))


def test_amax_kernel(K):
    """mvk_amax: max |x| by atomic max into a slot that keeps what it held; odd lengths, zeros, infinities."""
    d = dev()
    gen = g(67)
    for nel in (1, 3, 4, 1023, 4096 * 257 + 2):
        x = torch.randn(nel, generator=gen).to(d)
        pool = K.AmaxPool(x, 2)
        assert float(K.amax_of(x, pool.take())) == float(x.abs().max())
    slot = torch.full((1,), 7.5, device=d)
    assert float(K.amax_of(torch.zeros(100, device=d), slot)) == 7.5
    x = torch.randn(1000, generator=gen).to(d)
    x[77] = float("-inf")
    assert float(K.amax_of(x, torch.zeros(1, device=d))) == float("inf")


@pytest.mark.parametrize("n,H,W,Cin,Cout", [(3, 32, 32, 64, 64), (15, 16, 16, 128, 128), (20, 14, 14, 64, 128), (3, 7, 7, 128, 64)])
def test_conv3x3_fused_forms(K, n, H, W, Cin, Cout):
    """mvk_conv3x3_f / mvk_conv3x3_wgrad_f (csrc/conv3rs.hip): the activation of the producing layer applied while X is staged,
    the 0.1 of the residual branch on the convolution sum / on both gradients, the bias gradient out of the weight-gradient
    launch — against float64 torch.  Outside the covered problems the entry points refuse (MVK_EINVAL) instead of guessing."""
    from multivae_amd import _lib

    gen = g(59)
    d = dev()
    x = torch.randn(n, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (3 * Cin ** 0.5)
    b = 0.1 * torch.randn(Cout, generator=gen)
    res = torch.randn(n, Cout, H, W, generator=gen)
    src = torch.randn(n, Cout, H, W, generator=gen)
    dy = torch.randn(n, Cout, H, W, generator=gen)
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(d)
    (wf, _), = K.pack_weights([(w.to(d), "c3", True, True)])
    assert not K.conv3x3_fused_ok(n, H, W, Cin, Cout)  # too small for the default dispatch
    with pytest.raises(_lib.MvkError):
        K.conv3x3_f(nhwc(x), wf, None, n, H, W, Cin, Cout, x_act=K.LEAKY)
    _debug_flags(0x800)
    try:
        assert K.conv3x3_fused_ok(n, H, W, Cin, Cout)
        ax = F.leaky_relu(x.double(), 0.2)
        conv = F.conv2d(ax, w.double(), None, 1, 1)
        Y = K.conv3x3_f(nhwc(x), wf, b.to(d), n, H, W, Cin, Cout, act=K.LEAKY, x_act=K.LEAKY)
        close(Y, nhwc(F.leaky_relu(conv + b.double().view(1, -1, 1, 1), 0.2).float()), what="lrelu(conv(lrelu(x)) + b)", rtol=3e-6)
        bparam = torch.zeros(Cout, device=d).requires_grad_(True)
        bparam.grad = torch.zeros(Cout, device=d)
        Y, _ = K.conv3x3_f(nhwc(x), wf, None, n, H, W, Cin, Cout, y_act_src=nhwc(src), y_src_act=K.LEAKY, out_bias=bparam,
                           pre_scale=0.1)
        ref = 0.1 * F.conv2d(x.double(), w.double(), None, 1, 1) * torch.where(src > 0, 1.0, 0.2).double()
        close(Y, nhwc(ref.float()), what="0.1 * conv * act'(src)", rtol=3e-6)
        close(bparam.grad, ref.sum((0, 2, 3)).float(), what="column sums", rtol=1e-5)
        Y = K.conv3x3_f(nhwc(x), wf, b.to(d), n, H, W, Cin, Cout, res=nhwc(res), res_alpha=0.1, x_act=K.LEAKY)
        close(Y, nhwc((res.double() + 0.1 * (conv + b.double().view(1, -1, 1, 1))).float()), what="res + 0.1 conv(lrelu(x))", rtol=3e-6)
        w64 = w.double().requires_grad_(True)
        F.conv2d(ax, w64, None, 1, 1).backward(0.1 * dy.double())
        wparam = w.to(d).clone().requires_grad_(True)
        wparam.grad = torch.zeros_like(wparam)
        bparam.grad.zero_()
        K.conv3x3_wgrad_f(nhwc(x), nhwc(dy), wparam, bparam, n, H, W, Cin, Cout, x_act=K.LEAKY, dy_scale=0.1)
        close(wparam.grad, w64.grad.float(), what="0.1 * weight gradient with lrelu(x)", rtol=3e-6)
        close(bparam.grad, (0.1 * dy.double().sum((0, 2, 3))).float(), what="bias gradient", rtol=1e-5)
        K.conv3x3_wgrad_f(nhwc(x), nhwc(dy), wparam, None, n, H, W, Cin, Cout, x_act=K.LEAKY, dy_scale=0.1)
        close(wparam.grad, 2 * w64.grad.float(), what="accumulates", rtol=3e-6)
    finally:
        _debug_flags(0)


@pytest.mark.parametrize("n,H,W,Cin,Cout", [(3, 7, 7, 64, 128), (2, 16, 16, 128, 64), (1, 5, 9, 6, 10), (2, 8, 8, 3, 64)])
def test_conv3x3_residual_epilogue(K, n, H, W, Cin, Cout):
    """mvk_conv3x3_res: res + alpha * (conv (+ bias, activation) * act'(mask source)) in the convolution's epilogue — the
    block sum `x_s + 0.1 * dx` of models/nn/cub.py:274-280 and the gradient sum of its backward pass."""
    gen = g(37)
    d = dev()
    x = torch.randn(n, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (3 * Cin ** 0.5)
    b = 0.1 * torch.randn(Cout, generator=gen)
    res = torch.randn(n, Cout, H, W, generator=gen)
    src = torch.randn(n, Cout, H, W, generator=gen)
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(d)
    (wf, _), = K.pack_weights([(w.to(d), "c3", True, True)])
    conv = F.conv2d(x, w, b, 1, 1)
    Y = K.conv3x3(nhwc(x), wf, b.to(d), n, H, W, Cin, Cout, act=K.NONE, res=nhwc(res), res_alpha=0.1)
    close(Y, nhwc(res + 0.1 * conv), what="res + 0.1 * conv")
    Y = K.conv3x3(nhwc(x), wf, b.to(d), n, H, W, Cin, Cout, act=K.LEAKY, y_act_src=nhwc(src), y_src_act=K.LEAKY, res=nhwc(res))
    close(Y, nhwc(res + F.leaky_relu(conv, 0.2) * torch.where(src > 0, 1.0, 0.2)), what="res + act(conv) * mask")
    with pytest.raises(Exception):
        K.conv3x3(nhwc(x), wf, None, n, H, W, Cin, Cout, res=nhwc(res), out_bias=torch.zeros(Cout, device=d))


def test_device_rng(K):
    """mvk_device_rng: N(0, 1) / U[lo, hi) noise whose generator state lives in device memory.  Moments and tails, the tie to
    torch's seed (re-seeding restarts the sequence, also with the same seed), and a captured launch that draws fresh noise at
    every replay and continues the eager sequence."""
    d = dev()
    torch.manual_seed(1234)
    x = K.device_randn((1 << 20,), d)
    assert abs(float(x.mean())) < 5e-3 and abs(float(x.var()) - 1.0) < 1e-2
    assert abs(float((x ** 3).mean())) < 2e-2 and abs(float((x ** 4).mean()) - 3.0) < 5e-2
    assert 0.002 < float((x.abs() > 3).float().mean()) < 0.0034 and float(x.abs().max()) < 6.5
    y = K.device_randn((1 << 20,), d)
    assert abs(float((x * y).mean())) < 5e-3 and not torch.equal(x, y)  # the next draw is a new, uncorrelated block
    u = K.device_randn((3, 1001), d, uniform=True, lo=-1.0 + 1.2e-7, hi=1.0)  # ragged size: scalar tail stores
    assert float(u.min()) >= -1.0 + 1.2e-7 and float(u.max()) < 1.0 and abs(float(u.mean())) < 0.05
    torch.manual_seed(1234)
    x2, y2 = K.device_randn((1 << 20,), d), K.device_randn((1 << 20,), d)
    assert torch.equal(x, x2) and torch.equal(y, y2)
    torch.manual_seed(99)
    assert not torch.equal(K.device_randn((1 << 20,), d), x)
    # a captured draw: every replay continues the sequence an eager run would produce
    torch.manual_seed(7)
    ref = [K.device_randn((5, 512, 20), d) for _ in range(4)]
    torch.manual_seed(7)
    first = K.device_randn((5, 512, 20), d)  # the state exists before the capture starts
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_, stream=st):
            z = K.device_randn((5, 512, 20), d)
    torch.cuda.current_stream().wait_stream(st)
    assert torch.equal(first, ref[0])
    for i in (1, 2, 3):
        g_.replay()
        torch.cuda.synchronize()
        assert torch.equal(z, ref[i]), f"replay {i}"
    # a re-seed after the capture re-seeds the SAME state tensor in place: the existing graph keeps a valid
    # pointer and picks the new seed up at its next replay
    state_ptr = K._rng_state(d).data_ptr()
    torch.manual_seed(7)
    again = K.device_randn((5, 512, 20), d)  # eager draw: notices the re-seed
    assert K._rng_state(d).data_ptr() == state_ptr
    assert torch.equal(again, ref[0])
    g_.replay()
    torch.cuda.synchronize()
    assert torch.equal(z, ref[1]), "replay after an in-place re-seed"


@pytest.mark.parametrize("n,H,W,C", [(5, 16, 16, 256), (3, 7, 7, 256), (2, 5, 9, 3), (1, 1, 1, 1), (65536 + 3, 2, 3, 5)])
def test_transpose_act(K, n, H, W, C):
    """mvk_transpose_act: the NCHW flatten of an NHWC map with LeakyReLU in the same pass, its inverse, and both backward
    passes (the ResNet encoders' `fc(actvn(out).view(batch, -1))`, the decoders' `fc(z).view(-1, nf0, s0, s0)`)."""
    gen = g(61)
    d = dev()
    h = torch.randn(n, H, W, C, generator=gen)
    hd = h.to(d).requires_grad_(True)
    flat = K.nhwc_to_flat_nchw(hd, K.LEAKY)
    ref = F.leaky_relu(h, 0.2).permute(0, 3, 1, 2).reshape(n, C * H * W)
    assert torch.equal(flat.cpu(), ref)
    gflat = torch.randn(n, C * H * W, generator=gen)
    flat.backward(gflat.to(d))
    href = h.clone().requires_grad_(True)
    F.leaky_relu(href, 0.2).permute(0, 3, 1, 2).reshape(n, C * H * W).backward(gflat)
    close(hd.grad, href.grad, what="flatten backward", rtol=1e-6)
    fd = ref.to(d).requires_grad_(True)
    back = K.flat_nchw_to_nhwc(fd, C, H, W)
    assert torch.equal(back.cpu(), ref.view(n, C, H, W).permute(0, 2, 3, 1))
    gb = torch.randn(n, H, W, C, generator=gen)
    back.backward(gb.to(d))
    assert torch.equal(fd.grad.cpu(), gb.permute(0, 3, 1, 2).reshape(n, C * H * W))


@pytest.mark.parametrize("n,H,W,C", [(3, 28, 28, 64), (2, 7, 7, 20), (4, 14, 14, 128), (2, 5, 9, 3)])
def test_avgpool_upsample_axpby(K, n, H, W, C):
    """nn.AvgPool2d(3, 2, 1), nn.Upsample(scale_factor=2) forward / backward and the residual combination on NHWC."""
    gen = g(41)
    x = torch.randn(n, C, H, W, generator=gen, requires_grad=True)
    d = dev()
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(d)
    y = F.avg_pool2d(x, 3, 2, 1)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    close(K.avgpool(nhwc(x), n, H, W, C), nhwc(y), what="avgpool fwd")
    close(K.avgpool_bwd(nhwc(gy), n, H, W, C), nhwc(x.grad), what="avgpool bwd")
    x.grad = None
    u = F.interpolate(x, scale_factor=2)
    gu = torch.randn(u.shape, generator=gen)
    u.backward(gu)
    close(K.upsample2(nhwc(x), n, H, W, C), nhwc(u), what="upsample fwd")
    close(K.upsample2_bwd(nhwc(gu), n, H, W, C), nhwc(x.grad), what="upsample bwd")
    a, b = nhwc(x), torch.randn(n, H, W, C, generator=gen).to(d)
    close(K.axpby(a, 1.0, b, 0.1), a + 0.1 * b, what="axpby")
    close(K.axpby(a, 1.0, None, 0.0, act=K.LEAKY), F.leaky_relu(a, 0.2), what="leaky relu")


# ------------------------------------------------------------------------------------------------------------
# importance-sampled joint likelihood kernels (compute_joint_nll)
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("family,Kk,B,L,E,R,prior", [("normal", 7, 5, 5, 1, 2, False), ("normal", 33, 9, 20, 3, 2, False),
                                                     ("laplace_with_softmax", 12, 4, 20, 2, 2, True),
                                                     ("normal", 1000, 3, 64, 31, 5, True), ("normal", 2, 1, 3, 32, 0, False)])
def test_iwae_sample_logw_reduce(K, family, Kk, B, L, E, R, prior):
    """mvk_iwae_sample / mvk_iwae_logw / mvk_iwae_reduce against the torch.distributions formulas the reference's
    compute_joint_nll bodies use (oracle.elbo.latent_log_prob / latent_rsample), up to 32 experts."""
    from multivae_amd._lib import FAMILY

    gen = g(31)
    fam = "normal" if family == "normal" else "laplace_with_softmax"
    locs = [torch.randn(B, L, generator=gen) for _ in range(E)]
    sds = [torch.rand(B, L, generator=gen) * 0.8 + 0.3 for _ in range(E)]
    noise = torch.randn(Kk, B, L, generator=gen) if fam == "normal" else torch.rand(Kk, B, L, generator=gen) * 1.98 - 0.99
    rows = [torch.rand(Kk, B, generator=gen) * 50 for _ in range(R)]
    ploc = torch.randn(1, L, generator=gen) * 0.1 if prior else None
    psd = torch.rand(1, L, generator=gen) + 0.5 if prior else None
    z_ref = elbo.latent_rsample(fam, locs[0], sds[0], noise)
    lpz = elbo.latent_log_prob(fam, z_ref, ploc if prior else torch.zeros(()), psd if prior else torch.ones(())).sum(-1)
    lq = torch.stack([elbo.latent_log_prob(fam, z_ref, locs[e], sds[e]).sum(-1) for e in range(E)])
    lw_ref = -sum(rows) + lpz - (torch.logsumexp(lq, 0) - math.log(E)) if R else lpz - (torch.logsumexp(lq, 0) - math.log(E))
    ll_ref = torch.logsumexp(lw_ref, 0) - math.log(Kk)
    d = dev()
    z = K.iwae_sample(locs[0].to(d), sds[0].to(d), noise.to(d), FAMILY[family])
    close(z, z_ref, rtol=1e-6, what="z")
    lw = K.iwae_logw(z, [r.to(d) for r in rows], [t.to(d) for t in locs], [t.to(d) for t in sds], FAMILY[family],
                     None if ploc is None else ploc.to(d), None if psd is None else psd.to(d))
    close(lw, lw_ref, rtol=1e-5, what="lw")
    ll = K.iwae_reduce([lw])
    close(ll, ll_ref, rtol=1e-5, what="ll")
    # several weight arrays are pooled (MMVAE+ concatenates its M conditioning modalities)
    lw2 = lw_ref + torch.randn(Kk, B, generator=gen)
    ll2 = K.iwae_reduce([lw, lw2.to(d)])
    close(ll2, torch.logsumexp(torch.cat([lw_ref, lw2]), 0) - math.log(2 * Kk), rtol=1e-5, what="ll pooled")


def test_iwae_entry_points_reject_bad_arguments(K):
    from multivae_amd._lib import MvkError

    d = dev()
    z = torch.zeros(2, 3, 4, device=d)
    loc = [torch.zeros(3, 4, device=d)] * 33
    with pytest.raises(MvkError):
        K.iwae_logw(z, [], loc, loc)  # more than MVK_IWAE_MAX_EXPERTS experts
    with pytest.raises(MvkError):
        K.iwae_reduce([torch.zeros(2, 3, device=d)] * 9)  # more than MVK_MAX_MODALITIES arrays


@pytest.mark.parametrize("M,B,L,bits,masked", [(2, 7, 5, [3, 1, 2], False), (4, 9, 20, [15, 1, 2, 4, 8, 6, 13], True),
                                               (3, 5, 70, [7], False), (8, 3, 4, [255, 128, 129], True)])
def test_mvae_posterior(K, M, B, L, bits, masked):
    """mvk_mvae_posterior_fwd/bwd against stable_poe (+ prior expert, +inf log-variance for missing rows), rsample
    and the closed-form KL of the oracle, with autograd gradients; z lands in every member modality's slab."""
    gen = g(41)
    S = len(bits)
    mus = [torch.randn(B, L, generator=gen) for _ in range(M)]
    lvs = [torch.randn(B, L, generator=gen) * 0.7 for _ in range(M)]
    eps = torch.randn(S, B, L, generator=gen)
    masks = None
    if masked:
        masks = [torch.rand(B, generator=gen) > 0.4 for _ in range(M)]
        masks[0][:] = True
        masks[-1][0] = False
    gz = [torch.randn(sum((b >> m) & 1 for b in bits), B, L, generator=gen) for m in range(M)]
    gk = torch.randn(S, B, generator=gen)
    # --- reference formulas
    rm = [t.clone().requires_grad_() for t in mus]
    rl = [t.clone().requires_grad_() for t in lvs]
    total = 0
    zs_ref, klds, stats = [], [], []
    slot = [0] * M
    for si, bt in enumerate(bits):
        ms, ls = [], []
        for m in range(M):
            if (bt >> m) & 1:
                lv = rl[m]
                if masks is not None:
                    lv = torch.where(masks[m].unsqueeze(1), lv, torch.full_like(lv, float("inf")))
                ms.append(rm[m])
                ls.append(lv)
        ms.append(torch.zeros(B, L))
        ls.append(torch.zeros(B, L))
        mu_s, lv_s = elbo.stable_poe(torch.stack(ms), torch.stack(ls))
        z = elbo.rsample(mu_s, lv_s, eps[si])
        kld = -0.5 * (1 + lv_s - mu_s.pow(2) - lv_s.exp()).sum(-1)
        zs_ref.append(z)
        klds.append(kld)
        stats.append((mu_s, lv_s))
        total = total + (kld * gk[si]).sum()
        for m in range(M):
            if (bt >> m) & 1:
                total = total + (z * gz[m][slot[m]]).sum()
                slot[m] += 1
    total.backward()
    # --- kernels
    d = dev()
    dm = [t.to(d).requires_grad_() for t in mus]
    dl = [t.to(d).requires_grad_() for t in lvs]
    dmasks = None if masks is None else [t.to(d) for t in masks]
    outs = K.MVAEPosteriorFn.apply(eps.to(d), dmasks, bits, True, *dm, *dl)
    nz = sum(1 for m in range(M) if any((b >> m) & 1 for b in bits))
    zm, kld, smu, slv = outs[:nz], outs[nz], outs[nz + 1], outs[nz + 2]
    slot = [0] * M
    present = [m for m in range(M) if any((b >> m) & 1 for b in bits)]
    for si, bt in enumerate(bits):
        close(kld[si], klds[si], what=f"kld {si}")
        close(smu[si], stats[si][0], what=f"mu {si}")
        close(slv[si], stats[si][1], what=f"lv {si}")
        for m in range(M):
            if (bt >> m) & 1:
                close(zm[present.index(m)][slot[m]], zs_ref[si], what=f"z {si} in slab of modality {m}")
                slot[m] += 1
    loss = (kld * gk.to(d)).sum()
    for i, m in enumerate(present):
        loss = loss + (zm[i] * gz[m].to(d)).sum()
    loss.backward()
    for m in range(M):
        if m in present:
            close(dm[m].grad, rm[m].grad, what=f"dmu {m}")
            close(dl[m].grad, rl[m].grad, what=f"dlv {m}")
        else:
            assert float(dm[m].grad.abs().max()) == 0.0


def test_reduce_terms_handles_more_than_sixteen_terms(K):
    """mvk_reduce_terms with MVK_MAX_TERMS = 64 terms (MVAE: one per (modality, subset) pair plus the KL rows)."""
    from multivae_amd._lib import TermDesc, call, ptr, stream_ptr

    d = dev()
    n = 40
    vals = [torch.arange(i + 1, dtype=torch.float32, device=d) for i in range(n)]
    terms = (TermDesc * n)()
    for i, t in enumerate(terms):
        t.v, t.n, t.mask, t.period, t.coef, t.lossw = vals[i].data_ptr(), i + 1, None, 1, 0.5, 2.0
    out = torch.empty(n + 2, device=d)
    loss = torch.empty((), device=d)
    call("mvk_reduce_terms", terms, n, 3.0, ptr(out), ptr(loss), stream_ptr())
    exp = torch.tensor([0.5 * i * (i + 1) / 2 for i in range(n)])
    close(out[:n], exp, rtol=1e-6, what="terms")
    close(loss, 2.0 * exp.sum(), rtol=1e-6, what="loss")
    close(out[n + 1], 6.0 * exp.sum(), rtol=1e-6, what="loss_sum")
    # gfill: the constant row gradients coef * lossw of the terms that ask for them, written by the same launch
    gbuf = [torch.full((i + 1,), -7.0, device=d) for i in range(n)]
    for i, t in enumerate(terms):
        t.gfill = gbuf[i].data_ptr() if i % 3 == 0 else None
    call("mvk_reduce_terms", terms, n, 3.0, ptr(out), ptr(loss), stream_ptr())
    close(out[:n], exp, rtol=1e-6, what="terms (second launch)")
    for i in range(n):
        assert float(gbuf[i].min()) == float(gbuf[i].max()) == (1.0 if i % 3 == 0 else -7.0), i


def test_reduce_terms_on_several_workgroups_is_deterministic_and_exact():
    """mvk_reduce_terms_ws: the headline's terms (KL rows 1536, SVHN rows 5120, MNIST partial rows 7 x 5120,
    masked with a period) on up to 16 workgroups against a float64 sum; the same bits launch after launch (fixed slices, partials
    added in workgroup order by whichever workgroup arrives last), the arrival counter back at 0, gfill complete; ragged lengths
    around the 4096 / 1024-entry slice boundaries; and a workspace that is too small falls back to the one-workgroup launch."""
    from multivae_amd._lib import TermDesc, call, ptr, stream_ptr

    d = dev()
    gen = g(5)
    lens = [1536, 5120, 35840, 4097, 16 * 4096 + 1, 1023, 70001]
    vals = [torch.randn(n, generator=gen).to(d) * (i + 1) for i, n in enumerate(lens)]
    B = 512
    mask = (torch.rand(B, generator=gen) > 0.3).to(d)
    n = len(lens)
    terms = (TermDesc * n)()
    gbuf = [torch.full((m,), -7.0, device=d) for m in lens]
    for i, t in enumerate(terms):
        t.v, t.n, t.coef, t.lossw = vals[i].data_ptr(), lens[i], 0.25 + i, 1.5
        t.mask, t.period = (mask.data_ptr(), B) if i in (1, 2) else (None, 1)
        t.gfill = gbuf[i].data_ptr() if i % 2 == 0 else None
    exp = []
    for i, v in enumerate(vals):
        v64 = v.double().cpu()
        if i in (1, 2):
            v64 = v64 * mask.cpu().double().repeat(lens[i] // B)
        exp.append((0.25 + i) * float(v64.sum()))
    exp = torch.tensor(exp, dtype=torch.float64)
    ws = torch.zeros(1 + 32 * 64, device=d)
    outs = []
    for rep in range(5):
        out = torch.empty(n + 2, device=d)
        loss = torch.empty((), device=d)
        call("mvk_reduce_terms_ws", terms, n, 3.0, ptr(out), ptr(loss), ptr(ws), ws.numel(), stream_ptr())
        torch.cuda.synchronize()
        assert ws[:1].view(torch.int32).item() == 0  # the arrival counter is left at zero
        outs.append(out.clone())
        scale = torch.tensor([(0.25 + i) * float(vals[i].double().abs().sum()) for i in range(n)], dtype=torch.float64)
        assert bool(((out[:n].double().cpu() - exp).abs() <= 2e-6 * scale).all()), (out[:n], exp)
        close(loss, 1.5 * exp.sum(), rtol=1e-5, what="loss")
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    for i in range(n):
        assert float(gbuf[i].min()) == float(gbuf[i].max()) == ((0.25 + i) * 1.5 if i % 2 == 0 else -7.0), i
    out1 = torch.empty(n + 2, device=d)
    call("mvk_reduce_terms_ws", terms, n, 3.0, ptr(out1), None, ptr(ws), 16, stream_ptr())  # too small: one workgroup
    out2 = torch.empty(n + 2, device=d)
    call("mvk_reduce_terms", terms, n, 3.0, ptr(out2), None, stream_ptr())
    assert torch.equal(out1, out2)
    close(out1[:n], exp.float(), rtol=1e-5, what="one-workgroup terms")


@pytest.mark.parametrize("Kk,B,L", [(1, 7, 3), (4, 9, 70)])
def test_gauss_sample_kl(K, Kk, B, L):
    gen = g(51)
    mu, lv = torch.randn(B, L, generator=gen), torch.randn(B, L, generator=gen) * 0.6
    eps = torch.randn(Kk, B, L, generator=gen)
    gw, gk = torch.randn(Kk, B, L, generator=gen), torch.randn(B, generator=gen)
    rm, rl = mu.clone().requires_grad_(), lv.clone().requires_grad_()
    w_ref = elbo.rsample(rm, rl, eps)
    kl_ref = -0.5 * (1 - rl.exp() - rm.pow(2) + rl).sum(-1)
    ((w_ref * gw).sum() + (kl_ref * gk).sum()).backward()
    d = dev()
    dm, dl = mu.to(d).requires_grad_(), lv.to(d).requires_grad_()
    w, kl = K.GaussSampleKLFn.apply(eps.to(d), dm, dl)
    close(w, w_ref, what="w")
    close(kl, kl_ref, what="kl")
    ((w * gw.to(d)).sum() + (kl * gk.to(d)).sum()).backward()
    close(dm.grad, rm.grad, what="dmu")
    close(dl.grad, rl.grad, what="dlv")


@pytest.mark.parametrize("Kk,B,shape", [(1, 5, (4,)), (3, 7, (6, 10)), (2, 3, (32, 1590)), (4, 2, (3, 5, 130))])
def test_recon_nll_categorical(Kk, B, shape):
    """MVK_DIST_CATEGORICAL: x * log_softmax(recon + 1e-6) over the last dimension (base_utils.py:28-57), rows, fused
    gradient and second-pass gradient; one-hot and soft targets; CUB-sentence sized rows (32 positions x 1590 words)."""
    from multivae_amd._lib import DIST, ReconDesc, call, stream_ptr

    gen = g(61 + Kk)
    C = shape[-1]
    D = int(np.prod(shape))
    recon = torch.randn(Kk, B, *shape, generator=gen) * 2
    idx = torch.randint(0, C, (B, *shape[:-1]), generator=gen)
    x = F.one_hot(idx, C).float()
    x[0] = torch.softmax(torch.randn(*shape, generator=gen), -1)  # a soft target row
    mask = torch.rand(B, generator=gen) > 0.3
    rowcoef = torch.rand(Kk, B, generator=gen)
    rescale, coef = 1.3, 0.4
    rr = recon.clone().requires_grad_()
    rows_ref = elbo._row_nll("categorical", rr, x, rescale)
    (rows_ref * rowcoef * mask.float() * coef).sum().backward()
    d = dev()
    rd, xd, md, rcd = recon.to(d), x.to(d), mask.to(d), rowcoef.to(d)
    rows = torch.empty(Kk, B, device=d)
    drecon = torch.empty_like(rd)
    desc = (ReconDesc * 1)()
    e = desc[0]
    e.recon, e.x, e.mask, e.rows, e.drecon, e.rowcoef = (rd.data_ptr(), xd.data_ptr(), md.data_ptr(),
                                                         rows.data_ptr(), drecon.data_ptr(), rcd.data_ptr())
    e.D, e.dist, e.scale, e.rescale, e.coef, e.n_classes = D, DIST["categorical"], 1.0, rescale, coef, C
    call("mvk_recon_nll_fwd", desc, 1, Kk, B, stream_ptr())
    close(rows, rows_ref, what="categorical rows")
    close(drecon, rr.grad, what="categorical drecon (fused)")
    drecon2 = torch.zeros_like(rd)
    e.drecon, e.rows = drecon2.data_ptr(), None
    call("mvk_recon_nll_bwd", desc, 1, Kk, B, stream_ptr())
    close(drecon2, rr.grad, what="categorical drecon (second pass)")
    from multivae_amd._lib import MvkError

    e.n_classes = C + 1 if D % (C + 1) else 0
    with pytest.raises(MvkError):
        call("mvk_recon_nll_bwd", desc, 1, Kk, B, stream_ptr())


@pytest.mark.parametrize("n,h,w,Cu,Cv,act", [(1030, 16, 16, 3, 32, 1), (5, 8, 8, 3, 16, 0), (3, 16, 16, 1, 64, 1),
                                             (7, 8, 8, 2, 32, 3), (2, 16, 16, 4, 64, 1)])
def test_small_down_fwd(K, n, h, w, Cu, Cv, act):
    """mvk_conv4s2_small_down_fwd (LDS-staged NCHW image, fp32 MFMA) = Conv2d(Cu, Cv, 4, 2, 1) + bias + activation, also
    through mvk_conv4s2_down's routing for the network-input layer; more images than persistent workgroups."""
    from multivae_amd._lib import call, ptr, stream_ptr

    gen = g(71 + n)
    U = torch.randn(n, Cu, 2 * h, 2 * w, generator=gen)
    Wc = torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(16 * Cu)
    b = torch.randn(Cv, generator=gen)
    ref = F.conv2d(U, Wc, b, stride=2, padding=1)
    ref = {0: ref, 1: torch.relu(ref), 3: F.leaky_relu(ref, 0.2)}[act]
    d = dev()
    wd, _ = K.pack_conv(Wc.to(d))
    Ud, bd = U.to(d), b.to(d)
    V = torch.empty(n, h, w, Cv, device=d)
    call("mvk_conv4s2_small_down_fwd", ptr(Ud), ptr(wd), ptr(bd), ptr(V), n, h, w, Cu, Cv, act, stream_ptr())
    close(nchw(V.cpu()), ref, rtol=2e-6, what="small down fwd")  # exact fp32 MFMA
    got = K.conv_down(Ud, wd, bd, n, h, w, Cu, Cv, act=act, u_nchw=True)
    assert torch.equal(got, V)
    V2 = torch.empty_like(V)
    call("mvk_conv4s2_small_down_fwd", ptr(Ud), ptr(wd), None, ptr(V2), n, h, w, Cu, Cv, 0, stream_ptr())
    close(nchw(V2.cpu()), F.conv2d(U, Wc, None, stride=2, padding=1), rtol=2e-6, what="no bias")


def test_zero_row_launches_are_no_ops(K):
    """Empty inputs: every row-parallel entry point accepts 0 rows, launches nothing and leaves its outputs alone."""
    from multivae_amd._lib import DIST, ReconDesc, call, ptr, stream_ptr

    d = dev()
    w = torch.randn(8, 6, device=d)
    y = K.linear_fwd(torch.empty(0, 6, device=d), w, torch.zeros(8, device=d), 0)
    assert y.shape == (0, 8)
    z = torch.empty(3, 0, 5, device=d)
    assert K.iwae_sample(torch.empty(0, 5, device=d), torch.empty(0, 5, device=d), z).shape == (3, 0, 5)
    sentinel = torch.full((4,), 7.0, device=d)
    desc = (ReconDesc * 1)()
    e = desc[0]
    e.recon, e.x, e.rows, e.D, e.dist, e.scale, e.rescale, e.coef = (sentinel.data_ptr(), sentinel.data_ptr(),
                                                                    sentinel.data_ptr(), 4, DIST["normal"], 1.0, 1.0, 1.0)
    call("mvk_recon_nll_fwd", desc, 1, 2, 0, stream_ptr())  # B = 0
    V = torch.empty(0, 8, 8, 32, device=d)
    call("mvk_conv4s2_small_down_fwd", ptr(sentinel), ptr(torch.zeros(48 * 32, device=d)), None, ptr(V), 0, 8, 8, 3, 32, 1,
         stream_ptr())
    call("mvk_gauss_sample_kl_fwd", ptr(sentinel), ptr(sentinel), ptr(sentinel), 1, 0, 4, ptr(sentinel), ptr(sentinel),
         stream_ptr())
    call("mvk_adam_step", ptr(sentinel), ptr(sentinel), ptr(sentinel), ptr(sentinel), 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0,
         stream_ptr())
    torch.cuda.synchronize()
    assert bool((sentinel == 7.0).all())


def test_public_base_utils_helpers_match_reference_goldens():
    """multivae_amd.models.base.base_utils.{poe, stable_poe, kl_divergence, rsample_from_gaussian, set_decoder_dist,
    cross_entropy} (the reference's import path, base_utils.py:28-172) on their HIP kernels: values against the unit
    goldens generated from the reference (incl. logvar = +inf experts), gradients against the oracle's autograd."""
    from multivae_amd.models.base import base_utils as BU

    _, a = G.load_case("unit_base_utils")
    d = dev()
    mus, lvs, fin = G.t(a["mus"]), G.t(a["lvs"]), G.t(a["fin_lvs"])
    pm, pl = BU.poe(mus.to(d), lvs.to(d))
    close(pm, G.t(a["poe_mu"]), rtol=1e-5, what="poe mu")
    close(pl, G.t(a["poe_lv"]), rtol=1e-5, what="poe lv")
    sm, sl = BU.stable_poe(list(mus.to(d)), list(fin.to(d)))
    close(sm, G.t(a["spoe_mu"]), rtol=1e-5, what="stable_poe mu")
    close(sl, G.t(a["spoe_lv"]), rtol=1e-5, what="stable_poe lv")
    one_mu, one_lv = BU.stable_poe(mus[:1].to(d), fin[:1].to(d))  # a single expert is returned as is
    assert torch.equal(one_mu.cpu(), mus[0]) and torch.equal(one_lv.cpu(), fin[0])
    close(BU.kl_divergence(mus[0].to(d), fin[0].to(d), mus[2].to(d), fin[2].to(d)), G.t(a["kl"]), rtol=1e-5, what="kl")
    # rsample_from_gaussian replays the reference's draw: same generator state -> same noise on the same device type is
    # not comparable across CPU / GPU generators, so check the deterministic parts and the sampling identity
    torch.manual_seed(5)
    z = BU.rsample_from_gaussian(mus[0].to(d), fin[0].to(d), N=4)
    torch.manual_seed(5)
    eps = torch.randn(4, 7, 5, device=d)
    close(z, (mus[0].to(d) + torch.exp(0.5 * fin[0].to(d)) * eps), rtol=1e-6, what="rsample")
    assert BU.rsample_from_gaussian(mus[0].to(d), fin[0].to(d), N=3, flatten=True).shape == (21, 5)
    assert BU.rsample_from_gaussian(mus[0].to(d), fin[0].to(d)).shape == (7, 5)
    assert torch.equal(BU.rsample_from_gaussian(mus[0].to(d), fin[0].to(d), N=2, return_mean=True)[1].cpu(), mus[0])
    recon, target = G.t(a["recon"]).to(d), G.t(a["target"]).to(d)
    close(BU.set_decoder_dist("normal", {})(recon, target), G.t(a["lp_normal"]), rtol=1e-5, what="normal")
    close(BU.set_decoder_dist("normal", dict(scale=0.75))(recon, target), G.t(a["lp_normal_s"]), rtol=1e-5, what="normal s")
    close(BU.set_decoder_dist("laplace", dict(scale=0.75))(recon, target), G.t(a["lp_laplace_s"]), rtol=1e-5, what="laplace")
    close(BU.set_decoder_dist("bernoulli", {})(recon, (target > 0.5).float()), G.t(a["lp_bernoulli"]), rtol=1e-5,
          what="bernoulli")
    close(BU.set_decoder_dist("categorical", {})(recon, G.t(a["onehot"]).to(d)), G.t(a["lp_categorical"]), rtol=1e-5,
          what="categorical")
    close(BU.cross_entropy(recon, dict(tokens=G.t(a["onehot"]).argmax(-1).to(d))), G.t(a["lp_categorical"]), rtol=1e-5,
          what="cross_entropy tokens")
    with pytest.raises(ValueError):
        BU.set_decoder_dist("poisson", {})
    # gradients against the oracle's autograd (finite log-variances; a +inf expert gets exactly zero gradient)
    gen = g(3)
    w1, w2 = torch.randn(7, 5, generator=gen), torch.randn(7, 5, generator=gen)
    for fn_gpu, fn_ref, lv_in in ((BU.poe, elbo.poe, lvs), (BU.stable_poe, elbo.stable_poe, fin)):
        mr, lr = mus.clone().requires_grad_(True), lv_in.clone()
        finite = torch.isfinite(lr)
        lr = torch.where(finite, lr, torch.full_like(lr, 30.0)).requires_grad_(True)
        om, ol = fn_ref(mr, lr)
        ((om * w1).sum() + (ol * w2).sum()).backward()
        mg, lg = mus.clone().to(d).requires_grad_(True), lr.detach().clone().to(d).requires_grad_(True)
        gm, gl = fn_gpu(mg, lg)
        ((gm * w1.to(d)).sum() + (gl * w2.to(d)).sum()).backward()
        close(mg.grad, mr.grad, rtol=1e-4, what=fn_ref.__name__ + " dmus")
        close(lg.grad, lr.grad, rtol=1e-4, what=fn_ref.__name__ + " dlvs")
    with_inf = fin.clone()
    with_inf[1, :3] = float("inf")  # a missing modality for three samples (its encoder output is ignored)
    inf_lv = with_inf.to(d).requires_grad_(True)
    mg = mus.clone().to(d).requires_grad_(True)
    gm, gl = BU.stable_poe(mg, inf_lv)
    ref_m, ref_l = elbo.stable_poe(mus, with_inf)
    close(gm, ref_m, rtol=1e-5, what="stable_poe mu with a +inf expert")
    close(gl, ref_l, rtol=1e-5, what="stable_poe lv with a +inf expert")
    (gm.sum() + gl.sum()).backward()
    assert torch.isfinite(mg.grad).all() and torch.isfinite(inf_lv.grad).all()
    assert float(inf_lv.grad[1, :3].abs().max()) == 0.0 and float(mg.grad[1, :3].abs().max()) == 0.0
    # kl_divergence with a broadcast [1, L] prior: gradients of all four operands
    ops_ref = [t.clone().requires_grad_(True) for t in (mus[0], lvs[0], mus[2][:1], lvs[2][:1])]
    (elbo.kl_divergence(*ops_ref) * w1[:, 0]).sum().backward()
    ops_gpu = [t.detach().clone().to(d).requires_grad_(True) for t in ops_ref]
    (BU.kl_divergence(*ops_gpu) * w1[:, 0].to(d)).sum().backward()
    for og, orf, nm in zip(ops_gpu, ops_ref, ("mean", "log_var", "prior_mean", "prior_log_var")):
        close(og.grad, orf.grad, rtol=1e-4, what="kl d" + nm)
    # decoder log-probabilities: d / d recon
    for name, params, tgt in (("normal", dict(scale=0.75), target), ("laplace", dict(scale=0.75), target),
                              ("bernoulli", {}, (target > 0.5).float()), ("categorical", {}, G.t(a["onehot"]).to(d))):
        rg = recon.detach().clone().requires_grad_(True)
        wgt = torch.randn(recon.shape, generator=gen).to(d)
        (BU.set_decoder_dist(name, dict(params))(rg, tgt) * wgt).sum().backward()
        rr = recon.detach().cpu().clone().requires_grad_(True)
        (elbo.recon_log_prob(name, rr, tgt.cpu(), params.get("scale", 1.0)) * wgt.cpu()).sum().backward()
        close(rg.grad, rr.grad, rtol=1e-4, what="d log_prob " + name)


@pytest.mark.parametrize("M,N,K,layout", [(512, 20, 2048, "kn"), (512, 20, 400, "nk"), (37, 5, 12, "nk"), (1, 32, 64, "kn"),
                                          (130, 17, 100, "kn")])
def test_heads_fwd_matches_float64(M, N, K, layout):
    """mvk_heads_fwd: both encoder heads in one launch (exact fp32 MFMA, K split over 4 waves) vs float64 on the CPU;
    `nk` = torch Linear weights [N][K], `kn` = the packed convolution heads [K][N]; ragged M / N / K tails."""
    from multivae_amd import kernels as K_

    DEV = dev()
    gen = g(M * 7 + N)
    x = torch.randn(M, K, generator=gen)
    w = [torch.randn(N, K, generator=gen) / K ** 0.5 for _ in range(2)]
    b = [torch.randn(N, generator=gen) for _ in range(2)]
    wd = [(wi if layout == "nk" else wi.t().contiguous()).to(DEV) for wi in w]
    sk, sn = (1, K) if layout == "nk" else (N, 1)
    mu, lv = K_.heads_fwd(x.to(DEV), wd[0], b[0].to(DEV), wd[1], b[1].to(DEV), N, sk, sn)
    for got, wi, bi in ((mu, w[0], b[0]), (lv, w[1], b[1])):
        ref = (x.double() @ wi.double().t() + bi.double()).float()
        close_elementwise(got.cpu(), ref, "heads_fwd", rtol=2e-6, atol_frac=2e-6)


@pytest.mark.parametrize("M,N,K,layout,nh,flat_c,deferred", [(512, 20, 2048, "kn", 2, 128, True), (512, 20, 512, "nk", 2, 0, True),
                                                            (130, 17, 48, "nk", 2, 0, False), (1, 32, 64, "kn", 1, 4, False),
                                                            (300, 5, 16, "nk", 1, 0, True), (512, 20, 2048, "kn", 2, 128, False)])
def test_heads_bwd_matches_float64(M, N, K, layout, nh, flat_c, deferred):
    """mvk_heads_bwd: the backward of the encoder heads in one launch (backward data with the activation mask, both weight
    gradients — in the Conv2d [L][C][4][4] layout with flat_c —, both bias gradients and the bias gradient of the layer
    below) vs float64; targets inside the flat gradient buffer (deferred finish) and plain tensors (workspace finish); the
    gradients ACCUMULATE into their targets."""
    from multivae_amd import kernels as K_

    DEV = dev()
    gen = g(M * 5 + N + K)
    x = torch.relu(torch.randn(M, K, generator=gen))
    w = [torch.randn(N, K, generator=gen) / K ** 0.5 for _ in range(nh)]
    dy = [torch.randn(M, N, generator=gen) for _ in range(nh)]
    wd = [(wi if layout == "nk" else wi.t().contiguous()).to(DEV) for wi in w]
    sk, sn = (1, K) if layout == "nk" else (N, 1)
    # parameters with gradients: views of one flat buffer (so that the deferred path takes them) or plain tensors
    sizes = [N * K] * nh + [N] * nh + [flat_c if flat_c else K]  # the layer below: Linear (K units) or Conv2d (flat_c channels)
    flat = torch.zeros(sum((n + 63) // 64 * 64 for n in sizes), device=DEV)
    flat.grad = torch.full_like(flat, 0.5)  # the launch must ADD to what is there
    params, off = [], 0
    for i, n in enumerate(sizes):
        p_ = flat[off:off + n].view((N, K) if i < nh else (n,)).detach().requires_grad_(True)
        p_.grad = flat.grad[off:off + n].view(p_.shape)
        params.append(p_)
        off += (n + 63) // 64 * 64
    wpar, bpar, prev = params[:nh], params[nh:2 * nh], params[2 * nh]
    xd = x.to(DEV)
    dyd = [t.to(DEV) for t in dy]

    def run():
        return K_.heads_bwd(xd, K_.RELU, dyd, wd, bpar, sk, sn, flat_c=flat_c, prev_bias=prev, dw_params=wpar)

    if deferred:
        with K_.deferred_reductions(flat):
            out = run()
    else:
        out = run()
    assert out is not None
    dx, gw, gb, gp = out
    assert all(t is None for t in gw + gb + [gp])  # accumulated straight into .grad
    dx_ref = sum(d.double() @ wi.double() for d, wi in zip(dy, w)) * (x > 0).double()
    close_elementwise(dx.cpu(), dx_ref.float(), "heads_bwd dx", rtol=2e-6, atol_frac=2e-6)
    for h in range(nh):
        ref = dy[h].double().t() @ x.double()  # [N][K], k = (tap, c) when flat_c
        if flat_c:
            ref = ref.view(N, K // flat_c, flat_c).permute(0, 2, 1).reshape(N, K)
        close_elementwise(wpar[h].grad.cpu() - 0.5, ref.float(), "heads_bwd dW", rtol=4e-6, atol_frac=4e-6)
        close_elementwise(bpar[h].grad.cpu() - 0.5, dy[h].double().sum(0).float(), "heads_bwd db", rtol=4e-6, atol_frac=4e-6)
    prev_ref = dx_ref.view(M, K // flat_c, flat_c).sum((0, 1)) if flat_c else dx_ref.sum(0)
    close_elementwise(prev.grad.cpu() - 0.5, prev_ref.float(), "heads_bwd d(previous bias)", rtol=4e-6, atol_frac=4e-6)


@pytest.mark.parametrize("M,N,K,tb,act,bias_mod", [(5120, 2048, 20, False, "relu", 128), (5120, 400, 20, True, "relu", 400),
                                                   (37, 8, 5, True, "none", 8), (9, 1028, 32, False, "sigmoid", 4),
                                                   (1, 4, 1, False, "none", 0), (600, 64, 13, True, "leaky", 64)])
def test_short_reduction_linear_matches_float64(M, N, K, tb, act, bias_mod):
    """K <= 32: mvk_gemm / mvk_linear_fwd take the register-resident-weight kernel (skinny.hip, exact fp32 FMA chains)
    instead of the tiled MFMA engine; both weight layouts, every activation, bias period, ragged tails; vs float64."""
    from multivae_amd import kernels as K_

    d = dev()
    gen = g(M + 3 * N + K)
    x = torch.randn(M, K, generator=gen)
    w = torch.randn(K, N, generator=gen) / K ** 0.5          # B[k][n]
    b = torch.randn(bias_mod, generator=gen) if bias_mod else None
    code = dict(none=K_.NONE, relu=K_.RELU, sigmoid=K_.SIGMOID, leaky=K_.ACT["leaky_relu_0.2"])[act]
    ref = x.double() @ w.double()
    if b is not None:
        ref = ref + b.double().repeat(N // bias_mod)
    ref = dict(none=lambda t: t, relu=torch.relu, sigmoid=torch.sigmoid,
               leaky=lambda t: torch.nn.functional.leaky_relu(t, 0.2))[act](ref)
    wd = (w.t().contiguous() if tb else w).to(d)
    got = K_.gemm(x.to(d), wd, M, N, K, tb=tb, bias=None if b is None else b.to(d), bias_mod=bias_mod, act=code)
    close_elementwise(got, ref.float(), "short-reduction linear", rtol=2e-6, atol_frac=2e-6)
    if tb and bias_mod == N:  # the same through mvk_linear_fwd (torch Linear layout)
        got2 = K_.linear_fwd(x.to(d), wd, b.to(d), code)
        assert torch.equal(got2, got)


@pytest.mark.parametrize("M,N,K,act", [(512, 400, 784, "relu"), (512, 400, 400, "relu"), (37, 50, 64, "none"), (1000, 17, 128, "sigmoid")])
def test_few_row_linear_matches_float64(M, N, K, act):
    """mvk_linear_fwd with <= 1024 rows: one 16 x 16 output tile per workgroup, K split over its waves (skinny.hip), instead
    of a split-K launch of the tiled engine plus its reduce; vs float64."""
    from multivae_amd import kernels as K_

    d = dev()
    gen = g(M + N + K)
    x = torch.randn(M, K, generator=gen)
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen)
    code = dict(none=K_.NONE, relu=K_.RELU, sigmoid=K_.SIGMOID)[act]
    ref = x.double() @ w.double().t() + b.double()
    ref = dict(none=lambda t: t, relu=torch.relu, sigmoid=torch.sigmoid)[act](ref)
    got = K_.linear_fwd(x.to(d), w.to(d), b.to(d), code)
    close_elementwise(got, ref.float(), "few-row linear", rtol=3e-6, atol_frac=3e-6)


# ---------------------------------------------------------------------------------------------------------------------
# csrc/dense16.hip: the MLP decoder on pre-split fp16 pair planes (reference: models/nn/default_architectures.py:225-258
# Decoder_AE_MLP, likelihood models/base/base_utils.py:62-87) against float64 on the CPU
# ---------------------------------------------------------------------------------------------------------------------
from multivae_amd._lib import call, ptr, stream_ptr  # noqa: E402  (ctypes prototypes only: nothing is loaded at import)


def _d16_planes(R, C_):
    t = torch.empty(2, R, C_, dtype=torch.float16, device=dev())
    return t[0], t[1]


def _d16_unsplit(hi, lo, bound, R, C_):
    out = torch.empty(R, C_, device=dev())
    call("mvk_dense16_unsplit", ptr(hi), ptr(lo), ptr(bound), None, 1.0, 0, R, C_, ptr(out), stream_ptr())
    return out


@pytest.mark.parametrize("M,L,H,D,B,spread", [(1280, 20, 512, 784, 128, 0.0), (1048, 20, 512, 784, 131, 2.0), (384, 32, 256, 200, 384, 0.0),
                                               (136, 8, 64, 72, 17, 1.0)])
def test_dense16_chain_vs_float64(M, L, H, D, B, spread):
    """mvk_dense16_pack / _first / _fwd_nll / _bwd_data / _wgrad / _unsplit: every stage against float64 at 3e-6 of the
    tensor's maximum (the tolerance of the scaled-fp16 convolution tests), row tiles and column tiles that do not divide the
    problem, a reduction length with a partial k-tile (K = 784, 200, 72), rows of very different magnitude (spread: a factor
    10^(+-spread) per row of z and per row of W1), weights per plane row on their own scale."""
    from multivae_amd import _lib as L_
    lib = L_.load()
    gen = g(11)
    z = torch.randn(M, L, generator=gen) * torch.pow(10.0, (torch.rand(M, 1, generator=gen) - 0.5) * 2 * spread)
    w0 = (torch.rand(H, L, generator=gen) - 0.5) * 2 / L ** 0.5
    b0 = (torch.rand(H, generator=gen) - 0.5) * 2 / L ** 0.5
    w1 = (torch.rand(D, H, generator=gen) - 0.5) * 2 / H ** 0.5 * torch.pow(10.0, (torch.rand(D, 1, generator=gen) - 0.5) * 2 * spread)
    b1 = (torch.rand(D, generator=gen) - 0.5) * 2 / H ** 0.5
    x = torch.rand(B, D, generator=gen) * 1.5 - 0.25
    scale, gw = 0.75, 1.0 / M
    zd, w0d, b0d, w1d, b1d, xd = (t.to(dev()) for t in (z, w0, b0, w1, b1, x))
    # weight planes, both orientations, per-row scales
    nk_hi, nk_lo = _d16_planes(D, H)
    kn_hi, kn_lo = _d16_planes(H, D)
    nk_inv, kn_inv = torch.empty(D, device=dev()), torch.empty(H, device=dev())
    call("mvk_dense16_pack", ptr(w1d), D, H, ptr(nk_hi), ptr(nk_lo), ptr(nk_inv), ptr(kn_hi), ptr(kn_lo), ptr(kn_inv), stream_ptr())
    w1r = (nk_hi.double() + nk_lo.double() / 2048) * nk_inv.double()[:, None]
    w1t = (kn_hi.double() + kn_lo.double() / 2048) * kn_inv.double()[:, None]
    keep = torch.ones(D, dtype=torch.bool)
    close_per_slice(w1r, w1.double(), keep, 3e-7, "dense16 weight planes [N][K], every row on its own scale")
    close_per_slice(w1t, w1.double().t().contiguous(), torch.ones(H, dtype=torch.bool), 3e-7, "dense16 weight planes [K][N]")
    # first layer under the a-priori bound
    zam, xam = torch.zeros(1, device=dev()), torch.zeros(1, device=dev())
    call("mvk_amax", ptr(zd), zd.numel(), ptr(zam), stream_ptr())
    call("mvk_amax", ptr(xd), xd.numel(), ptr(xam), stream_ptr())
    h_hi, h_lo = _d16_planes(M, H)
    bounds = torch.zeros(2, device=dev())
    call("mvk_dense16_first", ptr(zd), ptr(w0d), ptr(b0d), ptr(zam), ptr(h_hi), ptr(h_lo), ptr(bounds[0:1]), M, H, L, 1, stream_ptr())
    h64 = torch.relu(z.double() @ w0.double().t() + b0.double())
    assert float(bounds[0]) >= float(h64.max())
    close(_d16_unsplit(h_hi, h_lo, bounds[0:1], M, H), h64, 3e-6, "dense16 first layer")
    # output layer + Normal NLL tail
    g_hi, g_lo = _d16_planes(M, D)
    P, CR = lib.mvk_dense16_fwd_nll_rows(D), lib.mvk_dense16_colsum_rows(M)
    rows_part, cs = torch.zeros(P, M, device=dev()), torch.zeros(CR, D, device=dev())
    call("mvk_dense16_fwd_nll", ptr(h_hi), ptr(h_lo), ptr(bounds[0:1]), ptr(nk_hi), ptr(nk_lo), ptr(nk_inv), ptr(b1d), ptr(xd), B,
         ptr(xam), scale, gw, ptr(g_hi), ptr(g_lo), ptr(bounds[1:2]), ptr(rows_part), ptr(cs), M, D, H, stream_ptr())
    r64 = torch.sigmoid(h64 @ w1.double().t() + b1.double())
    xx = x.double()[torch.arange(M) % B]
    rows64 = (0.5 * (r64 - xx) ** 2 / scale ** 2).sum(1) + D * (math.log(scale) + 0.918938533204672742)
    g64 = gw * (r64 - xx) / scale ** 2 * r64 * (1 - r64)
    assert float(bounds[1]) >= float(g64.abs().max())
    G_ = _d16_unsplit(g_hi, g_lo, bounds[1:2], M, D)
    # The tail is NOT linear in the GEMM: an error d_pre of the pre-activation moves G by |dG / d pre| d_pre <= gw / s^2 (1 + |x|) / 4
    # d_pre, and d_pre is relative to sum_k |h_k w_k| (cancellation), not to |pre|.  Entry by entry: 1e-6 of that sum through the
    # derivative bound + 3e-6 of the tensor's maximum (spread = 0: the second term alone covers it, as in the linear tests).
    mag = h64 @ w1.double().abs().t() + b1.double().abs()
    slack = 1e-6 * mag * gw / scale ** 2 * 0.25 * (1 + xx.abs())
    exc = (G_.double().cpu() - g64).abs() - slack
    assert float(exc.max()) <= 3e-6 * float(g64.abs().max()), ("dense16 d NLL / d pre-activation", float(exc.max()), float(g64.abs().max()))
    rslack = (1e-6 * mag * (r64 - xx).abs() / scale ** 2 * 0.25).sum(1)
    rexc = (rows_part.sum(0).double().cpu() - rows64).abs() - rslack
    assert float(rexc.max()) <= 2e-6 * float(rows64.abs().max()), ("dense16 NLL rows", float(rexc.max()))
    if spread == 0.0:
        close(rows_part.sum(0), rows64, 2e-6, "dense16 NLL rows")
        close(G_, g64, 3e-6, "dense16 d NLL / d pre-activation")
    close(cs.sum(0), G_.double().cpu().sum(0), 3e-6, "dense16 column sums")
    g64 = G_.double().cpu()  # the stages below are checked against what THIS gradient implies (each kernel on its own)
    # backward data (ReLU mask from the hi plane of h) + bias gradient of the first layer
    ws = torch.empty(max(CR * 2 * H, 64 * D * H) + 1024, device=dev())
    dh, db0 = torch.empty(M, H, device=dev()), torch.zeros(H, device=dev())
    call("mvk_dense16_bwd_data", ptr(g_hi), ptr(g_lo), ptr(bounds[1:2]), ptr(kn_hi), ptr(kn_lo), ptr(kn_inv), ptr(h_hi), ptr(dh), ptr(db0),
         ptr(ws), ws.numel(), M, H, D, stream_ptr())
    dh64 = (g64 @ w1.double()) * (h64 > 0)
    close(dh, dh64, 3e-6, "dense16 backward data")
    close(db0, dh64.sum(0), 3e-6, "dense16 db0")
    # weight gradient (transposing LDS reads) + bias gradient from the forward's partials
    dw1, db1 = torch.zeros(D, H, device=dev()), torch.zeros(D, device=dev())
    call("mvk_dense16_wgrad", ptr(g_hi), ptr(g_lo), ptr(bounds[1:2]), ptr(h_hi), ptr(h_lo), ptr(bounds[0:1]), ptr(cs), CR, ptr(dw1),
         ptr(db1), ptr(ws), ws.numel(), M, D, H, stream_ptr())
    close(dw1, g64.t() @ h64, 3e-6, "dense16 weight gradient")
    close(db1, g64.sum(0), 3e-6, "dense16 db1")
    # planes -> fp32 with one factor per (column tile, row): the general backward path
    rf = torch.rand(P, M, generator=gen) + 0.5
    out = torch.empty(M, D, device=dev())
    call("mvk_dense16_unsplit", ptr(g_hi), ptr(g_lo), ptr(bounds[1:2]), ptr(rf.to(dev())), 2.0, 128, M, D, ptr(out), stream_ptr())
    fac = rf.double()[torch.arange(D) // 128].t() * 2.0  # [M, D]
    close(out, g64 * fac, 3e-6, "dense16 unsplit with row factors")


def test_mlp_decoder_fused_tail_matches_generic_path(monkeypatch):
    """Decoder_AE_MLP.reconstruction_nll (planes + fused tail) against forward + the generic likelihood on the same inputs:
    NLL rows, and every gradient for (a) the row weight the planes were built with (constant upstream gradient: the planes
    path) and (b) an arbitrary upstream gradient per row (the general path through mvk_dense16_unsplit)."""
    from multivae_amd import kernels
    from multivae_amd.models.base.base_config import BaseAEConfig
    from multivae_amd.models.nn.default_architectures import Decoder_AE_MLP

    monkeypatch.setattr(kernels, "DENSE16_MIN_ROWS", 1)
    torch.manual_seed(3)
    K_, B, L = 3, 96, 20
    dec = Decoder_AE_MLP(BaseAEConfig(latent_dim=L, input_dim=(1, 28, 28))).to(dev())
    z = torch.randn(K_, B, L, device=dev(), requires_grad=True)
    x = torch.rand(B, 1, 28, 28, device=dev())
    scale, w = 0.75, 1.0 / (K_ * B)

    def generic(up):
        for p in dec.parameters():
            p.grad = None
        z.grad = None
        rec = dec(z).reconstruction.reshape(K_, B, -1)
        rows = (0.5 * (rec - x.reshape(1, B, -1)) ** 2 / scale ** 2).sum(-1) + 784 * (math.log(scale) + 0.918938533204672742)
        (rows * up).sum().backward()
        return rows.detach(), [p.grad.clone() for p in dec.parameters()], z.grad.clone()

    def fused(up, const):
        for p in dec.parameters():
            p.grad = None
        z.grad = None
        part = dec.reconstruction_nll(z, x, "normal", scale, row_weight=w)
        assert part is not None and part.shape[1:] == (K_, B)
        if const:  # what ReconLossFn does on the unit-seed path: a registered constant gradient buffer
            gbuf = torch.full_like(part, w)
            kernels.register_const_grad(gbuf, w)
            part.backward(gbuf)
        else:
            (part * up.unsqueeze(0)).sum().backward()
        return part.detach().sum(0), [p.grad.clone() for p in dec.parameters()], z.grad.clone()

    for const in (True, False):
        up = torch.full((K_, B), w, device=dev()) if const else torch.rand(K_, B, device=dev()) * 2 * w
        r0, g0, dz0 = generic(up)
        r1, g1, dz1 = fused(up, const)
        close(r1, r0, 2e-6, "fused MLP tail: NLL rows")
        for a, b_, (nm, _) in zip(g1, g0, dec.named_parameters()):
            close(a, b_, 1e-5, f"fused MLP tail ({'const' if const else 'general'}): d {nm}")
        close(dz1, dz0, 1e-5, "fused MLP tail: dz")


def test_integration_md_binding_runs_with_two_modalities():
    """the reference-side binding INTEGRATION.md documents is EXECUTED — its code blocks, verbatim, against
    libmvk.so — with two modalities (the descriptors travel as an array: a wrong struct stride shows up in the second one) and
    compared with the oracle's restatement of recon_log_probs + row sums (base_utils.py:62-87, mopoe_model.py:186-208)."""
    from multivae_amd import _lib
    from test_host_logic import integration_md_python

    code = integration_md_python().replace('C.CDLL("libmvk.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    gen = g(77)
    Kk, B = 3, 13
    d = dev()
    shapes, dists, rescales = [(1, 28, 28), (3, 32, 32)], ["normal", "laplace"], [3.918, 1.0]
    recons = [torch.randn(Kk, B, *s, generator=gen) for s in shapes]
    xs = [torch.rand(B, *s, generator=gen) for s in shapes]
    rows = ns["recon_nll_rows"]([r.to(d) for r in recons], [x.to(d) for x in xs], [_lib.DIST[n] for n in dists], rescales, Kk, B)
    torch.cuda.synchronize()
    assert len(rows) == 2
    for r, x, n, rs, got in zip(recons, xs, dists, rescales, rows):
        close(got, elbo._row_nll(n, r, x, rs, 1.0), what=f"INTEGRATION.md recon_nll_rows ({n})")
    # the collective binding of the same document: one rank is a valid world (the W > 1 form needs W devices)
    import torch.distributed as dist

    if not dist.is_initialized():
        import os
        import socket

        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=0, world_size=1)
        made = True
    else:
        made = False
    try:
        comm = ns["make_comm"](0, 1)
        buf = torch.randn(5000, generator=gen).to(d)
        want = buf.clone()
        ns["average_gradients"](buf, comm)
        torch.cuda.synchronize()
        assert torch.equal(buf, want)  # the mean over one rank
        assert _lib.load().mvk_comm_destroy(comm) == 0
    finally:
        if made:
            dist.destroy_process_group()
