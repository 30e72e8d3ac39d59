"""mvk_probe_mfma_bf16 in its three operand modes (constants / hashed bf16 / hashed fp16 on the f16 instruction) at several launch lengths."""
import sys

import torch

sys.path.insert(0, ".")
from multivae_amd import _lib

d = torch.device("cuda:0")
out = torch.empty(65536, device=d)
per_iter = 256 * 4 * 64 * 32768.0
for iters in (16, 64, 256):
    for mode in (0, 1, 2, 1, 2):
        def run(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                _lib.call("mvk_probe_mfma_bf16", _lib.ptr(out), iters, mode, _lib.stream_ptr())
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / reps
        run(3)
        sec = run(20)
        print(f"iters {iters:4d} mode {mode}: {sec * 1e6:7.1f} us per launch, {per_iter * iters / sec / 1e12:7.1f} TFLOP/s", flush=True)
