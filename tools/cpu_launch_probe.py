"""How long does the host need to enqueue one training step?  (run on the GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multivae_amd.data.datasets.base import DatasetOutput
from multivae_amd.trainers import FlatParams, FusedAdam
dev = torch.device("cuda", 0)
model = bench.build_model(10, 20, dev)
flat = FlatParams(model); opt = FusedAdam(flat, lr=1e-3)
inputs = DatasetOutput(data=bench.synthetic_batch(512, dev))
gen = torch.Generator(device=dev).manual_seed(1)
def step():
    eps = torch.randn(10, 512, 20, device=dev, generator=gen)
    opt.zero_grad(); out = model(inputs, noise=eps); out.loss.backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/N:.3f} ms/step, total {1e3*(t2-t0)/N:.3f} ms/step")
# host-only cost: same loop with the GPU idle in between
ts = []
for _ in range(10):
    torch.cuda.synchronize(); a = time.perf_counter(); step(); ts.append(time.perf_counter() - a)
print(f"host time for one step with an idle GPU: {1e3*sum(ts)/len(ts):.3f} ms")
