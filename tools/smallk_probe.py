#!/usr/bin/env python3
"""The K <= 32 first decoder layer (mvk_gemm_smallk_amax) alone at the headline shape: M = 5120, N = 2048, K = 20."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multivae_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda:0")
M, N, K = 5120, 2048, 20
z = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) * 0.2; b = torch.randn(128, device=dev)
y = torch.empty(M, N, device=dev); am = torch.zeros(1, device=dev)
def run(amax):
    call("mvk_gemm_smallk_amax", ptr(z), ptr(w), ptr(y), M, N, K, 0, ptr(b), 128, 1, ptr(am) if amax else None, stream_ptr())
for amax in (True, False):
    for _ in range(3): run(amax)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(amax)
    e1.record(); torch.cuda.synchronize()
    print(f"amax={amax}: {e0.elapsed_time(e1) * 50:.1f} us per launch")
ref = torch.relu(z.double() @ w.double() + b.double().repeat(N // 128))
print("rel err", float((y.double() - ref).abs().max() / ref.abs().max()))
