"""Per-wave cycle split of the register-stationary 3x3 kernels (a -DMVK_C3PROF variant build, tools/conv3_variants.sh; run ON the
GPU box with MVK_LIB_PATH=build/v/libmvk_<name>.so): total cycles, cycles waiting at the per-tile barrier."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from multivae_amd import _lib, kernels as K  # noqa: E402

lib = _lib.load()
lib.mvk_c3_debug_buffer.argtypes = [ctypes.c_void_p]
d = torch.device("cuda:0")
buf = torch.zeros(256 * 4 * 2, dtype=torch.int64, device=d)
lib.mvk_c3_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
n, H, Cin, Cout = 128, 64, 64, 64
x = torch.randn(n, H, H, Cin, device=d)
w = torch.randn(Cout, Cin, 3, 3, device=d) / (3 * Cin ** 0.5)
src = torch.randn(n, H, H, Cout, device=d)
(wf, wb), = K.pack_weights([(w, "c3", True, True)])
pool = K.AmaxPool(x, 64)
xam = K.amax_of(x, pool.take())
tiles = (n * (H + 1) * (H + 1) + 31) // 32 / 256
for name, fn in (("bf16x3 masked", lambda: K.conv3x3_f(x, wf, None, n, H, H, Cin, Cout, y_act_src=src, y_src_act=K.LEAKY, pre_scale=0.5)),
                 ("fp16x2 masked", lambda: K.conv3x3_s(x, wf, None, n, H, H, Cin, Cout, xam, wf.mvk_amax, pool.take(), y_act_src=src, y_src_act=K.LEAKY)),
                 ("bf16x3 plain ", lambda: K.conv3x3_f(x, wf, None, n, H, H, Cin, Cout, pre_scale=0.5)),
                 ("fp16x2 plain ", lambda: K.conv3x3_s(x, wf, None, n, H, H, Cin, Cout, xam, wf.mvk_amax, pool.take()))):
    fn()
    torch.cuda.synchronize()
    buf.zero_()
    fn()
    torch.cuda.synchronize()
    t = buf.view(256, 4, 2).double().cpu()
    tot, bar = t[..., 0], t[..., 1]
    print(f"{name}: cycles/wave mean {tot.mean():.0f} max {tot.max():.0f} ({tot.mean() / tiles:.0f} per tile) | barrier wait "
          f"{100 * (bar / tot).mean():.1f} % ({bar.mean() / tiles:.0f} cycles per tile; per wave of a workgroup "
          f"{[round(float(v), 1) for v in (100 * bar / tot).mean(0)]})")
