"""bf16-piece kernels vs the scaled-fp16 form of the 4x4 / stride-2 layers (csrc/imgconv.hip) at the headline batch
(n = K B = 5120 images): HIP events around 10 launches incl. the Python call.  Run on the GPU box."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multivae_amd import kernels as K

d = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5120


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


print(f"n = {n}")
print("| layer | form | bf16x3 us | fp16x2 us | bf16x3 + y_amax us |")
for h, Cu, Cv in [(8, 32, 64), (4, 64, 128)]:
    gen = torch.Generator().manual_seed(h)
    Ud = torch.randn(n, 2 * h, 2 * h, Cu, generator=gen).to(d)
    Vd = torch.randn(n, h, h, Cv, generator=gen).to(d)
    Us, Vs = torch.randn(n, 2 * h, 2 * h, Cu, generator=gen).to(d), torch.randn(n, h, h, Cv, generator=gen).to(d)
    Wc = (torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(16 * Cu)).to(d)
    bu, bv = torch.randn(Cu, generator=gen).to(d), torch.randn(Cv, generator=gen).to(d)
    wd, wu = K.pack_conv(Wc)
    pb_u, pb_v = torch.nn.Parameter(torch.zeros(Cu, device=d)), torch.nn.Parameter(torch.zeros(Cv, device=d))
    pool = K.AmaxPool(Ud, 8)
    uam, vam, y = K.amax_of(Ud, pool.take()), K.amax_of(Vd, pool.take()), pool.take()
    rows = [("up forward", lambda a: K.conv_up(Vd, wu, bu, n, h, h, Cu, Cv, act=1, amax=a), vam, wu),
            ("up masked", lambda a: K.conv_up(Vd, wu, None, n, h, h, Cu, Cv, u_act_src=Us, u_act=1, out_bias=pb_u, amax=a), vam, wu),
            ("down forward", lambda a: K.conv_down(Ud, wd, bv, n, h, h, Cu, Cv, act=1, amax=a), uam, wd),
            ("down masked", lambda a: K.conv_down(Ud, wd, None, n, h, h, Cu, Cv, v_act_src=Vs, v_act=1, out_bias=pb_v, amax=a), uam, wd)]
    for name, fn, xam, wp in rows:
        t3 = t_us(lambda: fn(None))
        t2 = t_us(lambda: fn((xam, wp.mvk_amax, y)))
        t3y = t_us(lambda: fn((None, None, y)))
        print(f"| {Cu}<->{Cv} @{2 * h}<->{h} | {name} | {t3:.0f} | {t2:.0f} | {t3y:.0f} |")
    K.DIRECT_GRAD = False
    t3 = t_us(lambda: K.conv_wgrad(Ud, Vd, Wc, n, h, h, Cu, Cv))
    t2 = t_us(lambda: K.conv_wgrad(Ud, Vd, Wc, n, h, h, Cu, Cv, amax=(uam, vam)))
    print(f"| {Cu}<->{Cv} @{2 * h}<->{h} | weight gradient | {t3:.0f} | {t2:.0f} | |")
