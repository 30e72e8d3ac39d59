"""Capture + first replay of the full-size MoPoE step (the path that crashed in hip::Graph::UpdateStreams while side streams came from
torch's round-robin pool).  python tools/graph_replay_probe.py {0|1: fused tail} {eager_first|direct}"""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import test_gpu_trainer as T
from multivae_amd.data.datasets.base import DatasetOutput
from multivae_amd.trainers import FlatParams, GraphedStep
fused = sys.argv[1] == "1"
order = sys.argv[2]
d = torch.device("cuda:0")
B, K, L = 512, 10, 20
model = T._mnist_svhn_mopoe(d, K=K, L=L)
model.fused_decoder_tail = fused
g = torch.Generator().manual_seed(5)
inputs = DatasetOutput(data=dict(mnist=torch.rand(B, 1, 28, 28, generator=g).to(d), svhn=torch.rand(B, 3, 32, 32, generator=g).to(d)))
eps = torch.randn(K, B, L, generator=g).to(d)
if order == "eager_first":
    for kk in (K, 1, K):
        model.zero_grad(set_to_none=True)
        out = model(inputs, noise=eps[:kk].contiguous(), K=kk)
        out.loss.backward()
    torch.cuda.synchronize()
    print("eager ok", float(out.loss))
flat = FlatParams(model)
gs = GraphedStep(model, flat, inputs, noise=eps)
print("captured")
out_g = gs(inputs, eps)
torch.cuda.synchronize()
print("replay ok", float(out_g.loss))
