"""hipGraph replay of the fixed-shape part of a training step.

One step of the models here is ~100 short kernel launches driven from Python (ctypes + autograd): the host needs
about as long to enqueue them as the GPU needs to run them.  `GraphedStep` captures forward + backward
once (torch.cuda.CUDAGraph == hipGraph on ROCm, including the fork/join of the modality-branch streams) and
replays it with one launch per step; `zero_grad` runs in front of the replay and costs nothing when the optimizer cleared
the gradients while consuming them (`FusedAdam(zero_grad_in_step=True)`).

Single GPU (`optimizer=` given): the fused Adam is INSIDE the graph — its scalars (step, lr, ...) live in device memory
(mvk_adam_prepare / mvk_adam_step_dev), the preparation launch rides on a side branch, the update is the graph's last node:
the step is copy batch -> ONE replay.

Data parallel (`overlap=True`, one process per GPU): the collective stays outside the graph (RCCL launches are not captured), but
it no longer waits for the whole backward pass: the graph carries an external event-record node where everything but the last
encoder's gradients is final (kernels.OverlapPoint), `reduce_and_step` starts that part's all-reduce on a communication stream
behind the event — beside the rest of the backward pass —, the remainder behind the replay, then Adam:
copy batch -> replay || all-reduce(early ranges) -> all-reduce(late ranges) -> adam.  The reference's DDP overlaps its bucket
reductions with loss.backward() the same way (trainers/base/base_trainer.py:116-117,359).

Rotated (`rotate=optimizer`, single GPU): the decoders' weight-gradient leaves of step N, their finishes and their share of the
optimizer run at the HEAD of replay N + 1 on a branch of their own, beside the encoders' forward pass (kernels.Rotation); the
caller's optimizer.step() behind replay N covers everything else and leaves its scalars on the device for that branch.
`drain()` applies what is pending (end of an epoch, before evaluation / a checkpoint / an eager step); parameters after N
replays + drain are bit for bit those of N unrotated steps.

Capture needs static shapes and addresses: the batch and the noise are copied into buffers owned by this object;
a batch of another shape (the last one of an epoch) must go through the eager path.
"""
import os

import torch

from .. import _lib, kernels
from ..data.datasets.base import DatasetOutput


# MVK_ADAM_PRELUDE=0: the optimizer's scalar preparation directly in front of the update (end of the chain) instead of at the
# head of a side branch (A/B)
_ADAM_PRELUDE = _lib.tune("MVK_ADAM_PRELUDE", "1") != "0"


class GraphedStep:
    def __init__(self, model, flat, inputs, noise=None, warmup=3, capture_error_mode="global", optimizer=None, overlap=False,
                 rotate=None, **fwd_kwargs):
        """optimizer: a FusedAdam(zero_grad_in_step=True) to capture behind the backward pass (single-GPU steps; `includes_optimizer`
        says whether it was).  overlap: record the point where the early part of the gradient buffer is final (data-parallel
        steps; `reduce_and_step` uses it).  rotate: the FusedAdam the caller steps behind every replay — the decoders' late
        leaves and their part of the update move to the head of the next replay (`rotated` says whether the model had any)."""
        dev = flat.flat.device
        if dev.type != "cuda":
            raise RuntimeError("GraphedStep needs a GPU")
        self.model, self.flat, self.fwd_kwargs = model, flat, fwd_kwargs
        if rotate is not None and (optimizer is not None or overlap or not hasattr(rotate, "set_rotation")):
            raise ValueError("GraphedStep(rotate=...) excludes optimizer= / overlap= and needs a trainers.FusedAdam")
        self.rot_opt = rotate
        self.rotation = kernels.Rotation(dev) if rotate is not None else None
        self.rotated = False
        self.optimizer = optimizer if (optimizer is not None and getattr(optimizer, "zero_grad_in_step", False)
                                       and hasattr(optimizer, "step_captured") and not overlap) else None
        self.includes_optimizer = self.optimizer is not None
        self.overlap_point = kernels.OverlapPoint(dev) if overlap else None
        self.early_ranges = self.late_ranges = None
        if self.optimizer is not None:
            self.optimizer.sync_device_state()  # allocates the device scalars the captured launches point at
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
                for _ in range(max(int(warmup), 2 if self.rotation is not None else 1)):
                    self.flat.zero_grad()
                    self._body()
                if self.rotation is not None:  # the leaves the last pass registered stay unapplied: warm-up moves no parameter
                    self.flat.zero_grad()
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            if self.rotation is not None:
                ranges = flat.ranges_of(self.rotation.params)
                if ranges:
                    self.rot_opt.set_rotation(ranges)
                    self.rotated = True
                else:  # nothing in this model registers rotatable leaves: an ordinary captured step
                    self.rotation = None
            for st in (cur, kernels._side_stream(dev, 30)):  # the loss assembly's workspace: eager memory, made outside the capture
                kernels.terms_workspace(dev, st)
            self.graph = torch.cuda.CUDAGraph()
            # (stream priorities were tried: capturing the main branch, or the side branches, on a priority -1 stream
            # makes the replayed step 3.0-3.1 ms instead of 1.9)
            with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
                self.out = self._body(capture=True)
            op = self.overlap_point
            if op is not None and op.recorded:
                # strict: a late parameter that is not a view of the flat buffer leaves both None -> one collective behind the replay
                self.late_ranges = flat.ranges_of(op.late_params, strict=True)
                self.early_ranges = flat.complement(self.late_ranges) if self.late_ranges else None
                if not self.early_ranges:
                    self.early_ranges = self.late_ranges = None
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

    def _body(self, capture=False):
        kw = dict(self.fwd_kwargs)
        if self.noise is not None:
            kw["noise"] = self.noise
        dev = self.flat.flat.device
        opt = self.optimizer if capture else None  # the eager warm-up passes must not move the parameters
        if opt is not None and _ADAM_PRELUDE:
            kernels.set_prelude(dev, opt.prepare_captured)  # depends on nothing of the step: head of the first side branch
        if self.overlap_point is not None:
            self.overlap_point.begin()
        rot = self.rotation
        try:
            with kernels.deferred_reductions(self.flat):
                if rot is not None:  # the head branch: the previous step's leaves, finishes, update, packs
                    (rot.arm if kernels.ROT_ARM else rot.begin_step)(self.rot_opt.rot_update if (capture and self.rotated) else None)
                out = self.model(self.inputs, **kw)
                # the registered unit seed: filled once (not one launch per replay), and ReconLossFn.backward launches nothing
                out.loss.backward(gradient=kernels.unit_seed(out.loss))
            if opt is not None:
                kernels.run_prelude(dev)  # no branch took it: here, in front of the update
                if not _ADAM_PRELUDE:
                    opt.prepare_captured()
                opt.step_captured()
        finally:
            kernels._PRELUDE.pop(dev, None)
            if self.overlap_point is not None:
                self.overlap_point.end()
            if rot is not None:
                rot.end_step()
        return out

    def matches(self, inputs):
        return (set(inputs.data.keys()) == set(self.data.keys())
                and all(inputs.data[m].shape == self.data[m].shape for m in self.data)
                and (hasattr(inputs, "masks") == (self.masks is not None)))

    def __call__(self, inputs=None, noise=None):
        """Copy the batch (and noise) into the captured buffers, replay, return the captured ModelOutput (its tensors
        are overwritten by the next call)."""
        if inputs is not None and inputs is not self.inputs:
            pairs = [(self.data[m], v) for m, v in inputs.data.items() if v is not self.data[m]]
            if self.masks is not None:
                pairs += [(self.masks[m], v) for m, v in inputs.masks.items() if v is not self.masks[m]]
            self._copy_in(pairs)
        if noise is not None and self.noise is not None:
            if isinstance(noise, dict):
                for k, v in noise.items():
                    self.noise[k].copy_(v, non_blocking=True)
            elif noise is not self.noise:
                self.noise.copy_(noise, non_blocking=True)
        if self.optimizer is not None:
            self.optimizer.sync_device_state()  # a copy only when lr / step changed behind the graph's back
        self.flat.zero_grad()  # free after FusedAdam(zero_grad_in_step=True).step()
        self.graph.replay()
        if self.optimizer is not None:
            self.optimizer.note_replayed_step()
        if self.rotated:
            self.rot_opt.arm_rotation()  # the caller's optimizer.step() covers everything but the rotated ranges
        return self.out

    @staticmethod
    def _copy_in(pairs):
        """The batch into the captured buffers: ONE launch (mvk_copy_batch) where every pair is a same-typed, contiguous,
        16-byte-granular device tensor — two blits of 1.6 + 6.3 MB were 12 us back to back in front of every replay of the
        headline step —, torch's copy_ otherwise (dtype conversion, host tensors, odd sizes)."""
        import ctypes as C

        fast = [(d, s) for d, s in pairs if (torch.is_tensor(s) and s.is_cuda and s.device == d.device and s.dtype == d.dtype
                                             and s.shape == d.shape and s.is_contiguous() and d.is_contiguous()
                                             and (d.numel() * d.element_size()) % 16 == 0 and d.data_ptr() % 16 == 0
                                             and s.data_ptr() % 16 == 0)]
        if 2 <= len(fast) <= 8 and len(fast) == len(pairs) and os.environ.get("MVK_TWO_BLITS") != "1":  # (the switch: A/B only)
            descs = (_lib.CopyDesc * len(fast))()
            for e, (d, s) in zip(descs, fast):
                e.dst, e.src, e.bytes = d.data_ptr(), s.data_ptr(), d.numel() * d.element_size()
            _lib.call("mvk_copy_batch", descs, len(fast), _lib.stream_ptr())
            return
        for d, s in pairs:
            d.copy_(s, non_blocking=True)

    def drain(self):
        """Rotated steps only: apply what the last replay left pending — the late leaves of its step, their finishes and the
        update of the rotated parameters — on the current stream.  Afterwards every parameter (and optimizer moment) has seen
        the same number of updates; the next replay's head branch finds identity scalars and changes nothing.  Call it before
        anything but another replay reads the parameters or the optimizer state (evaluation, checkpoint, eager step)."""
        if not self.rotated or not self.rot_opt._rot_dirty:
            return
        with kernels.deferred_reductions(self.flat):
            self.rotation.run_pending()
        self.rot_opt.rot_update()
        self.rot_opt.rot_drained()
        self.flat.grads_zero = self.rot_opt.zero_grad_in_step

    def reduce_and_step(self, optimizer):
        """Data-parallel tail of a replayed step: the gradient collective (overlapped with the end of the backward pass when the
        capture recorded an early point), then the optimizer.  Every rank calls it after every replay, with the same ranges in
        the same order."""
        flat, dev = self.flat, self.flat.flat.device
        cur = torch.cuda.current_stream(dev)
        if self.early_ranges and self.late_ranges and os.environ.get("MVK_OVERLAP") != "2":  # 2: event node captured, serial collective (A/B)
            comm = kernels._side_stream(dev, 61)
            kernels.call("mvk_stream_wait_event", kernels.C.c_void_p(comm.cuda_stream), self.overlap_point.event)
            with torch.cuda.stream(comm):
                flat.all_reduce_mean_ranges(self.early_ranges)  # starts where the graph's event node fires
            comm.wait_stream(cur)  # ... the rest only behind the whole replay
            with torch.cuda.stream(comm):
                scale = flat.all_reduce_mean_ranges(self.late_ranges)
            cur.wait_stream(comm)
        else:
            scale = flat.all_reduce_mean()
        optimizer.step(grad_scale=scale)
