"""Fixed cost and per-image slope of the register-stationary kernels: time vs batch."""
import ctypes, math, sys
import torch
sys.path.insert(0, ".")
from multivae_amd import _lib, kernels as K
lib = _lib.load(); lib.mvk_debug_set_flags.argtypes = [ctypes.c_int]; lib.mvk_debug_set_flags(0x200)
d = torch.device("cuda:0")
K.DIRECT_GRAD = False
for h, Cu, Cv in ((8, 32, 64), (4, 64, 128)):
    gen = torch.Generator().manual_seed(h)
    Wc = (torch.randn(Cv, Cu, 4, 4, generator=gen) / math.sqrt(16 * Cu)).to(d)
    wd, wu = K.pack_conv(Wc)
    bu = torch.randn(Cu, generator=gen).to(d)
    pb = torch.nn.Parameter(torch.zeros(Cv, device=d))
    for n in (256, 512, 1024, 2048, 5120, 10240):
        Ud = torch.randn(n, 2 * h, 2 * h, Cu, generator=gen).to(d)
        Vd = torch.randn(n, h, h, Cv, generator=gen).to(d)
        cases = {"up": lambda: K.conv_up(Vd, wu, bu, n, h, h, Cu, Cv, act=1),
                 "down": lambda: K.conv_down(Ud, wd, None, n, h, h, Cu, Cv, v_act_src=Vd, v_act=1, out_bias=pb),
                 "wgrad": lambda: K.conv_wgrad(Ud, Vd, Wc, n, h, h, Cu, Cv)}
        out = []
        for nm, fn in cases.items():
            for _ in range(3): fn()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
            out.append(f"{nm} {best:7.1f}")
        print(f"h={h} n={n:6d}: " + "  ".join(out), flush=True)
