import torch, inspect
print(torch.__version__)
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True); e1 = torch.cuda.Event(enable_timing=True, external=True)
except TypeError as ex:
    print("no external kw:", ex); raise SystemExit
x = torch.randn(1 << 24, device="cuda")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x * 2
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    e0.record()
    y = x * 2 + 1
    e1.record()
    z = y.sum()
for _ in range(3):
    g.replay(); torch.cuda.synchronize()
    print("elapsed ms", e0.elapsed_time(e1))
