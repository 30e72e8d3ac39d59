import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_trainer as T
from multivae_amd import kernels
from multivae_amd.data.datasets.base import DatasetOutput
from multivae_amd.trainers import FlatParams
d = torch.device("cuda:0")
def grads(streams, B=64, K=3, L=8):
    kernels.BRANCH_STREAMS = streams
    model = T._mnist_svhn_mopoe(d, K=K, L=L)
    flat = FlatParams(model)
    g = torch.Generator().manual_seed(3)
    inputs = DatasetOutput(data=dict(mnist=torch.rand(B, 1, 28, 28, generator=g).to(d), svhn=torch.rand(B, 3, 32, 32, generator=g).to(d)))
    eps = torch.randn(K, B, L, generator=g).to(d)
    flat.zero_grad(); out = model(inputs, noise=eps); out.loss.backward(); torch.cuda.synchronize()
    names = [n for n, p in model.named_parameters()]
    return flat.grad.detach().cpu().clone(), names, [p.numel() for p in model.parameters()]
def cmp(a, b, name, names, sizes):
    diff = (a - b).abs()
    print(f"{name}: max abs diff {diff.max().item():.3e} rel-to-max {diff.max().item()/a.abs().max().item():.3e} differing {int((diff>0).sum())}")
    off = 0
    for n, s in zip(names, sizes):
        dd = diff[off:off+s]; aa = a[off:off+s]
        if dd.max() > 0: print(f"    {n:45s} max diff {dd.max().item():.3e}  |g|max {aa.abs().max().item():.3e}")
        off += s
a, names, sizes = grads(False); b, _, _ = grads(False); c, _, _ = grads(True); e, _, _ = grads(True)
cmp(a, b, "one stream twice", names, sizes)
cmp(c, e, "streams twice", names, sizes)
cmp(a, c, "one vs streams", names, sizes)
