"""Eager vs hipGraph-replayed training: same losses?  how fast?  (run on the GPU box)"""
import copy, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multivae_amd.data.datasets.base import DatasetOutput
from multivae_amd.trainers import FlatParams, FusedAdam, GraphedStep
dev = torch.device("cuda", 0)
K, B, L = 10, 512, 20
def run(graphed, steps=12):
    model = bench.build_model(K, L, dev, seed=0)
    flat = FlatParams(model); opt = FusedAdam(flat, lr=1e-3)
    inputs = DatasetOutput(data=bench.synthetic_batch(B, dev))
    gen = torch.Generator(device=dev).manual_seed(1)
    gs = None
    if graphed:
        gs = GraphedStep(model, flat, inputs, noise=torch.zeros(K, B, L, device=dev))
    losses = []
    for i in range(steps):
        eps = torch.randn(K, B, L, device=dev, generator=gen)
        if gs is not None:
            out = gs(inputs, eps)
        else:
            opt.zero_grad(); out = model(inputs, noise=eps); out.loss.backward()
        opt.step()
        losses.append(float(out.loss))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50):
        eps = torch.randn(K, B, L, device=dev, generator=gen)
        if gs is not None:
            out = gs(inputs, eps)
        else:
            opt.zero_grad(); out = model(inputs, noise=eps); out.loss.backward()
        opt.step()
    torch.cuda.synchronize()
    return losses, 1e3 * (time.perf_counter() - t0) / 50
le, te = run(False)
lg, tg = run(True)
print("eager  :", [round(x, 3) for x in le[:6]], f"{te:.3f} ms/step")
print("graphed:", [round(x, 3) for x in lg[:6]], f"{tg:.3f} ms/step")
print("max rel diff:", max(abs(a - b) / abs(a) for a, b in zip(le, lg)))
