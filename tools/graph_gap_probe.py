"""Where do the ~0.2 ms per step between the kernels go?  (run on the GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multivae_amd.data.datasets.base import DatasetOutput
from multivae_amd.trainers import FlatParams, FusedAdam, GraphedStep
dev = torch.device("cuda", 0); K, B, L = 10, 512, 20
model = bench.build_model(K, L, dev); flat = FlatParams(model); opt = FusedAdam(flat, lr=1e-3)
inputs = DatasetOutput(data=bench.synthetic_batch(B, dev))
gen = torch.Generator(device=dev).manual_seed(1)
gs = GraphedStep(model, flat, inputs, noise=torch.zeros(K, B, L, device=dev))
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print("replay only                :", round(timeit(lambda: gs.graph.replay()), 4), "ms")
print("replay + adam              :", round(timeit(lambda: (gs.graph.replay(), opt.step())), 4), "ms")
def full():
    eps = torch.randn(K, B, L, device=dev, generator=gen); gs(inputs, eps); opt.step()
print("randn + copy + replay + adam:", round(timeit(full), 4), "ms")
