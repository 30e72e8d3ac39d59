import ctypes as C, sys, torch
sys.path.insert(0, ".")
from multivae_amd import _lib
import os
lib = _lib.load(os.environ.get("LIB") or ("multivae_amd/libmvk_phases.so" if os.environ.get("PH") else None))
from multivae_amd import kernels as K
d = torch.device("cuda:0")
dbg = torch.zeros(8, dtype=torch.int64, device=d)
if os.environ.get("PH"): lib.mvk_debug_set_phase_buffer.argtypes = [C.c_void_p]
if os.environ.get("PH"): lib.mvk_debug_set_phase_buffer(C.c_void_p(dbg.data_ptr()))
def probe(name, fn):
    fn(); torch.cuda.synchronize(); dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    v = dbg.tolist(); nw = max(v[7], 1); nt = max(v[5], 1)
    print(f"{name}: {e0.elapsed_time(e1)*1e3:.0f} us | per-iteration cycles: load-issue {v[0]/nt:.0f} compute {v[1]/nt:.0f} wait+store {v[2]/nt:.0f} barrier {v[3]/nt:.0f} | vmwait+prologue/iter {v[4]/nt:.0f} total/wave {v[6]/nw:.0f} iters/wave {nt/nw:.1f}")
if os.environ.get("PH") or os.environ.get("LIB"):
    lib.mvk_debug_set_flags.argtypes = [C.c_int]
    lib.mvk_debug_set_flags(int(os.environ.get("FLAGS", "0")))
n = 5120
dg3 = torch.randn(n, 16, 16, 32, device=d); g2 = torch.relu(torch.randn(n, 8, 8, 64, device=d))
w2 = torch.randn(64, 32, 4, 4, device=d) * 0.05
wd2, wu2 = K.pack_conv(w2)
probe("dgrad dec.4 (down 32->64, N=64,K=512)", lambda: K.conv_down(dg3, wd2, None, n, 8, 8, 32, 64, 0, v_act_src=g2, v_act=1))
probe("fwd up2 (64->32, N=32,K=256 x4 parities)", lambda: K.conv_up(g2, wu2, None, n, 8, 8, 32, 64, 1))
probe("wgrad dec.4 (M=512,N=64,K=327680)", lambda: K.conv_wgrad(dg3, g2, w2, n, 8, 8, 32, 64))
for nn in (256, 512, 1024, 2048):
    a_ = torch.randn(nn, 16, 16, 32, device=d); m_ = torch.relu(torch.randn(nn, 8, 8, 64, device=d))
    probe(f"dgrad dec.4 with n={nn} ({nn*64//128} blocks)", lambda: K.conv_down(a_, wd2, None, nn, 8, 8, 32, 64, 0, v_act_src=m_, v_act=1))
x = torch.randn(5120, 512, device=d); w = torch.randn(784, 512, device=d); b = torch.zeros(784, device=d)
probe("linear fwd 5120x784x512", lambda: K.linear_fwd(x, w, b, 2))
