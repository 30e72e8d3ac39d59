"""Time the six large convolution GEMMs of the step under experiment flags (run on the GPU box).
   LIB=multivae_amd/libmvk_exper.so FLAGS=<bits> python tools/bf_probe.py
   bits: 1 no split arithmetic, 2 no global loads, 16 no split + no LDS writes, 32 no fragment reads/MFMA"""
import ctypes as C, os, sys, torch
sys.path.insert(0, ".")
from multivae_amd import _lib
lib = _lib.load(os.environ.get("LIB"))
from multivae_amd import kernels as K
d = torch.device("cuda:0")
if os.environ.get("LIB"):
    lib.mvk_debug_set_flags.argtypes = [C.c_int]
    lib.mvk_debug_set_flags(int(os.environ.get("FLAGS", "0")))
dbg = torch.zeros(8, dtype=torch.int64, device=d)
if os.environ.get("LIB"):
    lib.mvk_debug_set_phase_buffer.argtypes = [C.c_void_p]
    lib.mvk_debug_set_phase_buffer(C.c_void_p(dbg.data_ptr()))
def probe(name, fn, gf=21.47):
    fn(); torch.cuda.synchronize(); dbg.zero_(); fn(); torch.cuda.synchronize()
    v = dbg.tolist()
    if v[5]:
        nt = v[5]
        print(f"      per wave-k-tile cycles: mfma+split(+load wait) {v[0]/nt:.0f} | barrier1 {v[1]/nt:.0f} | lds write {v[2]/nt:.0f} | barrier2 {v[3]/nt:.0f}   (k-tiles/wave {nt/max(v[7],1):.1f})")
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print(f"{name:44s} {us:7.1f} us  {gf/us*1e3:6.1f} TF")
n = 5120
dg3 = torch.randn(n, 16, 16, 32, device=d); g2 = torch.relu(torch.randn(n, 8, 8, 64, device=d))
dg2 = torch.randn(n, 8, 8, 64, device=d); g1 = torch.relu(torch.randn(n, 4, 4, 128, device=d))
w2 = torch.randn(64, 32, 4, 4, device=d) * 0.05; w1 = torch.randn(128, 64, 4, 4, device=d) * 0.05
wd2, wu2 = K.pack_conv(w2); wd1, wu1 = K.pack_conv(w1)
print("FLAGS", os.environ.get("FLAGS", "0"))
probe("F1 up  128->64 4x4  (M=81920x4 N=64 K=512)", lambda: K.conv_up(g1, wu1, None, n, 4, 4, 64, 128, 1))
probe("F2 up  64->32  8x8  (M=327680x4 N=32 K=256)", lambda: K.conv_up(g2, wu2, None, n, 8, 8, 32, 64, 1))
probe("B1 wgrad 64,32      (M=1024 N=32 K=327680)", lambda: K.conv_wgrad(dg3, g2, w2, n, 8, 8, 32, 64))
probe("B2 down 32->64      (M=327680 N=64 K=512)", lambda: K.conv_down(dg3, wd2, None, n, 8, 8, 32, 64, 0, v_act_src=g2, v_act=1))
probe("B3 wgrad 128,64     (M=2048 N=64 K=81920)", lambda: K.conv_wgrad(dg2, g1, w1, n, 4, 4, 64, 128))
probe("B4 down 64->128     (M=81920 N=128 K=1024)", lambda: K.conv_down(dg2, wd1, None, n, 4, 4, 64, 128, 0, v_act_src=g1, v_act=1))
if hasattr(K, "to_bf3"):
    g1s, g2s, dg3s, dg2s = K.to_bf3(g1), K.to_bf3(g2), K.to_bf3(dg3), K.to_bf3(dg2)
    def chk(name, a, b):
        print(f"   {name}: max|diff|/max|ref| = {(a - b).abs().max().item() / b.abs().max().item():.2e}")
    chk("F1", K.conv_up(g1s, wu1, None, n, 4, 4, 64, 128, 1, in_bf3=True), K.conv_up(g1, wu1, None, n, 4, 4, 64, 128, 1))
    chk("B2", K.conv_down(dg3s, wd2, None, n, 8, 8, 32, 64, 0, v_act_src=g2, v_act=1, in_bf3=True), K.conv_down(dg3, wd2, None, n, 8, 8, 32, 64, 0, v_act_src=g2, v_act=1))
    probe("F1 pre-split A", lambda: K.conv_up(g1s, wu1, None, n, 4, 4, 64, 128, 1, in_bf3=True))
    probe("F2 pre-split A", lambda: K.conv_up(g2s, wu2, None, n, 8, 8, 32, 64, 1, in_bf3=True))
    probe("B2 pre-split A", lambda: K.conv_down(dg3s, wd2, None, n, 8, 8, 32, 64, 0, v_act_src=g2, v_act=1, in_bf3=True))
    probe("B4 pre-split A", lambda: K.conv_down(dg2s, wd1, None, n, 4, 4, 64, 128, 0, v_act_src=g1, v_act=1, in_bf3=True))
