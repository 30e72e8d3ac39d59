"""hipGraph replay of the fixed-shape part of a training step.

One step of the models here is ~100 short kernel launches driven from Python (ctypes + autograd): the host needs
about as long to enqueue them as the GPU needs to run them.  `GraphedStep` captures forward + backward
once (torch.cuda.CUDAGraph == hipGraph on ROCm, including the fork/join of the modality-branch streams) and
replays it with one launch per step; `zero_grad` runs in front of the replay and costs nothing when the optimizer cleared
the gradients while consuming them (`FusedAdam(zero_grad_in_step=True)`).  The gradient all-reduce and the fused Adam kernel stay outside the graph, so
the distributed step is: copy batch -> replay -> all_reduce(flat.grad) -> adam.

Capture needs static shapes and addresses: the batch and the noise are copied into buffers owned by this object;
a batch of another shape (the last one of an epoch) must go through the eager path.
"""
import os

import torch

from .. import kernels
from ..data.datasets.base import DatasetOutput


class GraphedStep:
    def __init__(self, model, flat, inputs, noise=None, warmup=3, capture_error_mode="global", **fwd_kwargs):
        dev = flat.flat.device
        if dev.type != "cuda":
            raise RuntimeError("GraphedStep needs a GPU")
        self.model, self.flat, self.fwd_kwargs = model, flat, fwd_kwargs
        self._seed = None
        self.data = {m: v.detach().clone() for m, v in inputs.data.items()}
        extra = {}
        if hasattr(inputs, "masks"):
            self.masks = {m: v.detach().clone() for m, v in inputs.masks.items()}
            extra["masks"] = self.masks
        else:
            self.masks = None
        self.inputs = DatasetOutput(data=self.data, **extra)
        self.noise = self._clone(noise)
        prof = kernels.PROFILE.pop("recon_nll", None)  # host-timed events cannot be recorded into a graph
        try:
            cur = torch.cuda.current_stream(dev)
            side = kernels._side_stream(dev, 62)  # a dedicated stream (not one of torch's pooled ones: _lib.new_stream)
            side.wait_stream(cur)
            with torch.cuda.stream(side):  # eager warm-up: fills every cache (scratch, packed masks, autotuned paths)
                for _ in range(max(int(warmup), 1)):
                    self.flat.zero_grad()
                    self._body()
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            # (stream priorities were tried: capturing the main branch, or the side branches, on a priority -1 stream
            # makes the replayed step 3.0-3.1 ms instead of 1.9)
            with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
                self.out = self._body()
        finally:
            if prof is not None:
                kernels.PROFILE["recon_nll"] = prof

    @staticmethod
    def _clone(noise):
        if noise is None:
            return None
        if isinstance(noise, dict):
            return {k: v.detach().clone() for k, v in noise.items()}
        return noise.detach().clone()

    def _body(self):
        kw = dict(self.fwd_kwargs)
        if self.noise is not None:
            kw["noise"] = self.noise
        with kernels.deferred_reductions(self.flat):
            out = self.model(self.inputs, **kw)
            # the registered unit seed: filled once (not one launch per replay), and ReconLossFn.backward launches nothing
            out.loss.backward(gradient=kernels.unit_seed(out.loss))
        return out

    def matches(self, inputs):
        return (set(inputs.data.keys()) == set(self.data.keys())
                and all(inputs.data[m].shape == self.data[m].shape for m in self.data)
                and (hasattr(inputs, "masks") == (self.masks is not None)))

    def __call__(self, inputs=None, noise=None):
        """Copy the batch (and noise) into the captured buffers, replay, return the captured ModelOutput (its tensors
        are overwritten by the next call)."""
        if inputs is not None and inputs is not self.inputs:
            for m, v in inputs.data.items():
                self.data[m].copy_(v, non_blocking=True)
            if self.masks is not None:
                for m, v in inputs.masks.items():
                    self.masks[m].copy_(v, non_blocking=True)
        if noise is not None and self.noise is not None:
            if isinstance(noise, dict):
                for k, v in noise.items():
                    self.noise[k].copy_(v, non_blocking=True)
            elif noise is not self.noise:
                self.noise.copy_(noise, non_blocking=True)
        self.flat.zero_grad()  # free after FusedAdam(zero_grad_in_step=True).step()
        self.graph.replay()
        return self.out
