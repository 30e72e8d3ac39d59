"""Error of the GEMM engines against an fp64 product (run on the GPU box): python tools/gemm_accuracy.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multivae_amd import kernels

torch.manual_seed(0)
dev = torch.device("cuda")
for (M, N, K) in [(5120, 512, 784), (40960, 128, 1024), (4096, 64, 2048), (327680, 32, 1024)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    ref = (x.double() @ w.double().t())
    y = kernels.linear_fwd(x, w, None, 0)
    yt = x @ w.t()
    den = ref.abs().max().item()
    print(f"M={M} N={N} K={K} engine={os.environ.get('MVK_ENGINE','f32')}: max|err|/max|ref| mvk={((y.double()-ref).abs().max().item()/den):.3e} "
          f"torch={((yt.double()-ref).abs().max().item()/den):.3e}  rms mvk={((y.double()-ref).pow(2).mean().sqrt().item()/ref.pow(2).mean().sqrt().item()):.3e} "
          f"torch={((yt.double()-ref).pow(2).mean().sqrt().item()/ref.pow(2).mean().sqrt().item()):.3e}")
