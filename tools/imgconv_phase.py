"""Per-wave cycle split of the register-stationary convolution kernels (ICPROF build of the library, run ON the GPU box):
total cycles, cycles waiting at the per-tile barrier, cycles inside the k-loops.  Build + run: tools/imgconv_phase.sh"""
import ctypes
import math
import sys

import torch

sys.path.insert(0, ".")
from multivae_amd import _lib, kernels as K  # noqa: E402

lib = _lib.load()
lib.mvk_debug_set_flags.argtypes = [ctypes.c_int]
lib.mvk_imgconv_debug_buffer.argtypes = [ctypes.c_void_p]
d = torch.device("cuda:0")
n = 5120
buf = torch.zeros(256 * 4 * 4, dtype=torch.int64, device=d)
lib.mvk_imgconv_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
lib.mvk_debug_set_flags(0x200)
K.DIRECT_GRAD = False
for h, Cu, Cv in [(8, 32, 64), (4, 64, 128)]:
    gen = torch.Generator().manual_seed(h)
    Ud = torch.randn(n, 2 * h, 2 * h, Cu, generator=gen).to(d)
    Vd = torch.randn(n, h, h, Cv, generator=gen).to(d)
    Wc = (torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(16 * Cu)).to(d)
    bu = torch.randn(Cu, generator=gen).to(d)
    wd, wu = K.pack_conv(Wc)
    pb = torch.nn.Parameter(torch.zeros(Cv, device=d))
    pool = K.AmaxPool(Ud, 8)
    uam, vam, ya = K.amax_of(Ud, pool.take()), K.amax_of(Vd, pool.take()), pool.take()
    for name, fn in (("up  ", lambda: K.conv_up(Vd, wu, bu, n, h, h, Cu, Cv, act=1)),
                     ("down", lambda: K.conv_down(Ud, wd, None, n, h, h, Cu, Cv, v_act_src=Vd, v_act=1, out_bias=pb)),
                     ("up   fp16", lambda: K.conv_up(Vd, wu, bu, n, h, h, Cu, Cv, act=1, amax=(vam, wu.mvk_amax, ya))),
                     ("down fp16", lambda: K.conv_down(Ud, wd, None, n, h, h, Cu, Cv, v_act_src=Vd, v_act=1, out_bias=pb,
                                                       amax=(uam, wd.mvk_amax, ya)))):
        fn()
        torch.cuda.synchronize()
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"-- {name}: {e0.elapsed_time(e1) * 1e3:.1f} us incl. launch; cycle counter rate = "
              f"{buf.view(256, 4, 4)[..., 0].double().mean().item() / (e0.elapsed_time(e1) * 1e3) / 1e3:.2f} GHz (lower bound)")
        t = buf.view(256, 4, 4).double().cpu()
        tot, bar, kl = t[..., 0], t[..., 1], t[..., 2]
        raw3 = buf.view(256, 4, 4)[..., 3].cpu()
        pre, post = (raw3 >> 32).double(), (raw3 & 0xffffffff).double()
        print(f"   one-tile-latency loop: after k-loop -> barrier {100 * (pre / tot).mean():.1f} % | barrier -> end of tile "
              f"{100 * (post / tot).mean():.1f} %   (two-tile-latency loop: 'k-loops' = pairs 0-6, second figure = pair 7)")
        print(f"h={h} {name}: cycles/wave mean {tot.mean():.0f} max {tot.max():.0f} | barrier wait mean {100 * (bar / tot).mean():.1f} % "
              f"(per wave of a workgroup: {[round(float(x), 1) for x in (100 * bar / tot).mean(0)]}) | k-loops {100 * (kl / tot).mean():.1f} % "
              f"| rest {100 * ((tot - bar - kl) / tot).mean():.1f} %  | MFMA floor {61440 if 'fp16' in name else 122880} = {100 * (61440 if 'fp16' in name else 122880) / tot.mean():.0f} %")
