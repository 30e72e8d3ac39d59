"""torch.autograd.Function wrappers over the C ABI of libmvk.so.

Every function here launches hand-written HIP kernels on the current PyTorch-ROCm stream; PyTorch is used
for device memory and autograd ordering only.  Whole networks (MLP / SVHN encoder and decoder) are single
autograd nodes so that a training step is a few dozen kernel launches with almost no Python in between,
and the backward passes use the producer-epilogue fusion of the engine (a layer's backward-data GEMM
multiplies by the previous layer's activation derivative and writes the pre-activation gradient directly).
"""
import ctypes as C
import math

import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import ACT, DIST, FAMILY, PackDesc, ReconDesc, SeedDesc, TermDesc, call, ptr, ptr_array, stream_ptr

RELU, SIGMOID, NONE = ACT["relu"], ACT["sigmoid"], ACT["none"]

# bench.py sets PROFILE["recon_nll"] = [] to collect (start, end) HIP events around the fused reconstruction-NLL
# launch on the stream it is launched on (roofline measurement inside the timed region).
PROFILE = {}
# accumulate parameter gradients directly into pre-existing .grad buffers (see _grad_target)
DIRECT_GRAD = True


def _c(t):
    """contiguous fp32 CUDA tensor (no copy when already so)."""
    _lib.require_gpu_tensor(t)
    return t if t.is_contiguous() else t.contiguous()


def _new(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _zeros(shape, like):
    return torch.zeros(shape, dtype=torch.float32, device=like.device)


FMT_IN_BF3 = 1  # mvk.h MVK_FMT_IN_BF3


def to_bf3(x):
    """fp32 tensor -> pre-split tensor [3, *x.shape] of bf16 (mvk_f32_to_bf3)."""
    x = x.contiguous()
    out = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    call("mvk_f32_to_bf3", ptr(x), x.numel(), out.data_ptr(), stream_ptr())
    return out


_WS = {}
WS_FLOATS = 16 * 1024 * 1024  # 64 MB of split-K slab scratch per device (largest need on the path: 8.4M floats)


def _ws(like):
    """Caller-owned split-K scratch (see mvk.h): one persistent buffer per (device, stream), reused by every launch
    on that stream (launches are stream-ordered, so reuse is safe; two streams must not share one)."""
    key = (like.device, torch.cuda.current_stream(like.device).cuda_stream)
    t = _WS.get(key)
    if t is None:
        t = torch.empty(WS_FLOATS, dtype=torch.float32, device=like.device)
        _WS[key] = t
    return t


# -----------------------------------------------------------------------------------------------------
# deferred leaf reductions (mvk_defer_begin, include/mvk.h)
# -----------------------------------------------------------------------------------------------------
# Weight / bias gradients are leaves: between `deferred_reductions(flat)` and its exit the ordered finishes of every
# gradient that accumulates into the flat buffer (split-K slabs, column-sum partials, convolution weight-gradient slabs)
# are queued and run in ONE launch at exit instead of one small launch each in the middle of the backward chain.
DEFER = _lib.tune("MVK_DEFER", "1") != "0"
# MVK_DEFER_SIDE=1: flush the decoders' finishes on a side stream when the posterior backward starts.  Measured on the
# headline step: 1.528 ms vs 1.50 ms with one flush at the end (the streaming finish kernel delays every launch of the
# latency-bound encoder backward it runs beside): off.
DEFER_SIDE = _lib.tune("MVK_DEFER_SIDE", "0") == "1"
# The arena starts at 64 MB and grows between steps to what the last step asked for (mvk_defer_wanted), up to MVK_DEFER_MB
# (default 1024): an MLP model keeps 64 MB, the MnistSvhn models settle at ~200 MB, the ResNet configurations at the cap
# (a fixed 512 MB per device was wasteful for small models and too small for cfg5).  Grown only outside a stream
# capture; a replaced arena stays alive (a captured graph may hold its address).
DEFER_ARENA_CAP_FLOATS = int(os.environ.get("MVK_DEFER_MB", "1024")) * (1 << 18)
DEFER_ARENA_START_FLOATS = min(64 * (1 << 18), DEFER_ARENA_CAP_FLOATS)
_ARENA = {}
_ARENA_RETIRED = []
_ARENA_CAPTURED = set()  # data_ptr of every arena that was live during a stream capture


_DEFER_ACTIVE = set()  # devices inside deferred_reductions
_LATE_USED = {}  # device -> streams that hold late leaves / partial flushes since mvk_defer_begin (joined at its end)
# MVK_LATE_LEAVES=1: the weight gradients of the large decoder (two ~100 us launches that fill the chip) are enqueued AFTER
# its backward-data chain, on a stream that is joined only where the deferred finishes run: they execute beside the
# launch-latency-bound encoder backward (the ~300 us tail of the step in which the chip is mostly idle) instead of in
# front of it.
# MEASURED (headline step, same box, two pairs): 1.347 / 1.336 ms without vs 1.320 / 1.315 ms with; a smaller grid for the
# weight-gradient kernels (MVK_IMGWGRAD_GRID=224 / 192 / 128: compute units left free for the chain) does not help
# (1.331 / 1.348 / 1.414 ms).  MVK_LATE_LEAVES=0 disables.
LATE_LEAVES = _lib.tune("MVK_LATE_LEAVES", "1") != "0"


class late_leaves:
    """with late_leaves(device, *tensors_read): the enclosed launches go to the late-leaf stream, ordered behind what the
    current stream holds now; nothing waits for them until deferred_reductions ends.  Outside deferred_reductions (or with
    MVK_LATE_LEAVES=0, or on the CPU) the launches stay on the current stream."""

    def __init__(self, device, *reads):
        self.on = LATE_LEAVES and device.type == "cuda" and device in _DEFER_ACTIVE
        self.device, self.reads, self._ctx = device, reads, None

    def __enter__(self):
        if self.on:
            st = _side_stream(self.device, 30)
            st.wait_event(torch.cuda.current_stream(self.device).record_event())
            for t in self.reads:
                t.record_stream(st)
            _LATE_USED.setdefault(self.device, []).append(st)
            self._ctx = torch.cuda.stream(st)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
            self._ctx = None
        return False


_LATE_CALLS = {}  # device -> closures run on the late-leaf stream when deferred_reductions ends (behind every other late leaf)
# MVK_LATE_DENSE=1: the MLP decoder's weight gradients are postponed behind the convolutional late leaves, into the tail of the
# step.  MEASURED (three same-box pairs): 1.099 / 1.093 / 1.094 ms without vs 1.114 / 1.104 / 1.106 ms with — the tail (the
# encoders' backward: a dependent chain of small kernels) slows down by what the decoder window gains; off.
LATE_DENSE = _lib.tune("MVK_LATE_DENSE", "0") == "1"


# Only the first layer's weight gradient (a split-K launch of 16 us alone, 50-60 us beside the convolutions) is postponed: autograd
# orders the posterior's backward behind EVERYTHING the decoder's backward node enqueued on its stream, leaves included.
# MEASURED (three same-box pairs): 1.046 / 1.050 / 1.039 -> 1.042 / 1.040 / 1.032 ms.  MVK_LATE_DW0=0: inside the node.
LATE_DW0 = _lib.tune("MVK_LATE_DW0", "1") != "0"
# MVK_SKIP_LATE=1 (TIMING ONLY, wrong gradients): the SVHN decoder's late weight gradients are not launched at all — the bound of
# what moving them out of the step's tail can gain (NOTES_r06 section 1)
SKIP_LATE = _lib.tune("MVK_SKIP_LATE", "0") == "1"


# MVK_LATE_GATE=1: the large decoder's register-stationary weight gradients (one 512-register wave per SIMD on every CU for ~150 us)
# start only behind the convolutional encoder's LAST backward-data launch (late_gate): the encoder's own 512-register launches
# cannot start on a CU that holds such a workgroup, its small-register ones (heads, split-K weight gradients) can run beside them
LATE_GATE = _lib.tune("MVK_LATE_GATE", "0") == "1"
_LATE_GATE_EV = {}  # device -> event recorded by late_gate (main stream)
_LATE_GATED = {}    # device -> closures that wait for it


def late_gate(device):
    """SVHNEncoderFn.backward, behind its last backward-data launch (the caller's stream must be the step's main stream)."""
    if LATE_GATE and device in _LATE_GATED:
        _LATE_GATE_EV[device] = torch.cuda.current_stream(device).record_event()


def run_gated(device, fn, *reads):
    """Inside deferred_reductions: fn() on the late-leaf stream, enqueued at the end of the scope behind every other postponed
    leaf and behind the late_gate event (when one was recorded).  False: the caller runs fn itself."""
    if not (LATE_GATE and LATE_LEAVES) or device.type != "cuda" or device not in _DEFER_ACTIVE:
        return False
    _LATE_GATED.setdefault(device, []).append((fn, reads))
    return True


def run_last(device, fn, *reads, force=False, params=()):
    """Inside deferred_reductions: run fn() on the late-leaf stream AFTER everything else the backward pass puts there (the
    enqueue itself is postponed to the end of the scope), i.e. in the launch-latency-bound tail of the step where the chip is
    mostly idle, instead of beside the decoders' backward-data chains whose window is throughput-bound.  Returns False when
    there is no such scope (the caller runs fn itself)."""
    if not (LATE_LEAVES and (LATE_DENSE or force)) or device.type != "cuda" or device not in _DEFER_ACTIVE:
        return False
    _LATE_CALLS.setdefault(device, []).append((fn, reads))
    op = _OVERLAP.get(device)
    if op is not None:  # these gradients are written in the tail of the step: not part of the early collective
        op.late_params.extend(params)
    return True


# -----------------------------------------------------------------------------------------------------
# where the backward pass says "this part of the gradient buffer is final" (overlapped data-parallel collective)
# -----------------------------------------------------------------------------------------------------
_OVERLAP = {}  # device -> OverlapPoint, while trainers.graph enqueues / captures a data-parallel step


class OverlapPoint:
    """The reference's DDP reduces gradient buckets while loss.backward() still runs (trainers/base/base_trainer.py:116-117,359).
    Here the step is ONE replayed hipGraph, so the overlap needs a point INSIDE the graph that a stream OUTSIDE can wait for: an
    external event-record node (mvk_event_record(ev, 1, stream)), placed where the finishes of everything but the last backward
    node have run (defer_flush_sibling: the decoders', the posterior's and the short encoder's gradients are final there, the
    long encoder's backward chain — ~300 us of the headline step — is still ahead).  `late_params` collects the parameters whose
    gradients are NOT final at that point: the node that called defer_flush_sibling and every leaf postponed by run_last."""

    def __init__(self, device):
        self.device = device
        self.event = C.c_void_p()
        call("mvk_event_create", C.byref(self.event))
        self.late_params = []
        self.recorded = False

    def begin(self):
        self.late_params, self.recorded = [], False
        _OVERLAP[self.device] = self

    def end(self):
        _OVERLAP.pop(self.device, None)

    def __del__(self):
        try:
            if self.event:
                _lib.load().mvk_event_destroy(self.event)
        except Exception:
            pass


# -----------------------------------------------------------------------------------------------------
# the rotated step: leaf gradients of step N at the head of step N + 1 (trainers/graph.py GraphedStep(rotate=True))
# -----------------------------------------------------------------------------------------------------
# The reference's loop is strictly sequential (trainers/base/base_trainer.py:350-361: zero_grad, forward, backward, step), but
# nothing in its semantics ties a DECODER's weight gradients and their share of optimizer.step() of step N to anything before
# that decoder's forward pass of step N + 1.  In a rotated step the decoders' weight-gradient launches (leaves of the backward
# pass: nothing waits for them) are not enqueued by the backward pass; their closures are kept and run at the head of the
# next step on a branch of their own — beside the encoders' forward pass and the posterior, the launch-latency-bound head of
# the step in which the chip is mostly idle — followed by their ordered finishes, the Adam update of exactly those
# parameters (mvk_adam_step_dev with the scalars the step's main update published: mvk_adam_step_pub) and the weight packs
# that read them; the decoders' forward pass waits for that branch.  Every parameter is updated once per step, with the same
# gradient and the same scalars, before it is next read: the parameters after N steps + a drain are bit for bit those of N
# unrotated steps (tests/test_gpu_trainer.py).  What a leaf reads must survive into the next replay at a fixed address:
# `Rotation.buf` hands out persistent buffers, `Rotation.slots` persistent amax slots (zeroed by the branch behind the leaves).
# MVK_ROT_SVHN / MVK_ROT_MLP: 0 = that decoder's leaves stay in the step they belong to; 1 = all of its weight-gradient leaves are
# rotated (the large, register-stationary launches too); 2 = only its FIRST layer's (a split-K GEMM of 10-25 us whose workgroups
# fit beside anything: the kind of launch the head of the step has room for).
ROT_SVHN = int(_lib.tune("MVK_ROT_SVHN", "1"))
ROT_MLP = int(_lib.tune("MVK_ROT_MLP", "1"))
# MVK_ROT_ARM=1: the head branch forks BEHIND the first launch of the forward pass (so that the main chain is the root chain that is
# enqueued first); measured slower than forking in front of it (0.955 vs 0.948 ms, three pairs; unrotated 0.938): off
ROT_ARM = _lib.tune("MVK_ROT_ARM", "0") != "0"
_ROTATE = {}  # device -> Rotation, while trainers.graph enqueues / captures a rotated step
_ROTATE_ARMED = {}  # device -> Rotation whose head branch starts behind the forward pass's first launch (Rotation.arm)


def rotation(device):
    """The open Rotation of a device, or None."""
    return _ROTATE.get(device) if _ROTATE else None


class Rotation:
    def __init__(self, device):
        self.device = device
        self.bufs = {}
        self.amax = torch.zeros(64, dtype=torch.float32, device=device)
        self.amax_pos = 0
        self.leaves = []    # registered by the backward pass that is being enqueued: (closure, parameters, prepare)
        self.pending = []   # those of the previous pass: what the head of the next one runs
        self.cache = {}     # pack key -> what the head's `prepare` closures packed BEHIND the update of the rotated parameters
        self.event = None
        self.waited = set()
        self.stream = _side_stream(device, 29)
        self.open = False

    def buf(self, name, shape, dtype=torch.float32):
        """A buffer that keeps its address from pass to pass (and from replay to replay)."""
        t = self.bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            if t is not None and torch.cuda.is_current_stream_capturing():
                raise _lib.MvkError(f"rotated step: buffer {name} changes shape inside a capture")
            t = torch.zeros(tuple(shape), dtype=dtype, device=self.device)
            self.bufs[name] = t
        return t

    def slots(self, n):
        """n amax slots that are zero now and stay untouched until the head of the next step has run its leaves."""
        if self.amax_pos + n > self.amax.numel():
            raise _lib.MvkError("rotated step: out of amax slots")
        self.amax_pos += n
        return self.amax[self.amax_pos - n:self.amax_pos]

    @property
    def params(self):
        """The parameters whose gradients (and update) are rotated, in registration order."""
        out, seen = [], set()
        for _, ps, _ in self.pending:
            for q in ps:
                if id(q) not in seen:
                    seen.add(id(q))
                    out.append(q)
        return out

    def push(self, fn, params, prepare=None):
        self.leaves.append((fn, tuple(params), prepare))

    def arm(self, update=None):
        """Inside deferred_reductions, before the forward pass: begin_step(update) will run behind the FIRST launch of the
        forward pass on the caller's stream (pack_scope's fill of the amax arena), not in front of it — in a captured graph the
        root chain that is enqueued first gets the launching hardware queue, and that has to be the step's main chain (with the
        head branch enqueued first the whole forward chain started ~20 us later: profiles/r06_rotated_small_step_timeline.txt)."""
        self.armed = (update,)
        _ROTATE_ARMED[self.device] = self

    def begin_if_armed(self):
        if self.armed is not None:
            update, = self.armed
            self.armed = None
            _ROTATE_ARMED.pop(self.device, None)
            self.begin_step(update)

    armed = None

    def begin_step(self, update=None):
        """Inside deferred_reductions, at the head of the forward pass: the branch of the previous step's leaves.  update: the
        closure that applies the optimizer to the rotated parameters (captured passes only; eager warm-up passes leave the
        parameters alone)."""
        cur = torch.cuda.current_stream(self.device)
        st = self.stream
        st.wait_event(cur.record_event())
        self.cache, self.waited, self.open = {}, set(), True
        with torch.cuda.stream(st):
            for fn, _, _ in self.pending:
                fn()
            if self.pending:
                call("mvk_defer_flush", stream_ptr())  # their ordered finishes: the gradients of the rotated parameters are final
                if update is not None:
                    update()
            self.amax.zero_()  # the slots the leaves read are free for this step's producers
            self.amax_pos = 0
            for _, _, prepare in self.pending:
                if prepare is not None:
                    prepare(self)
            self.event = st.record_event()
        self.leaves = []
        _ROTATE[self.device] = self

    def wait(self):
        """Order the current stream behind the head branch (whoever reads a rotated parameter, or a pack of one, calls this)."""
        if not self.open or self.event is None:
            return
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream not in self.waited:
            cur.wait_event(self.event)
            self.waited.add(cur.cuda_stream)
            for t in _tensors_of(list(self.cache.values())):  # allocated on the branch's stream, read on this one
                t.record_stream(cur)
                if getattr(t, "mvk_frag", None) is not None:
                    t.mvk_frag.record_stream(cur)

    def end_step(self):
        """Behind the backward pass: the branch is joined (it is, transitively, when a decoder waited for it), the leaves the
        backward pass registered become the next head's."""
        self.begin_if_armed()  # (a forward pass without a pack_scope never started it)
        _ROTATE.pop(self.device, None)
        if self.open:
            self.wait()
            self.open = False
        self.pending, self.leaves = self.leaves, []

    def run_pending(self):
        """The drain: the pending leaves on the current stream (inside deferred_reductions), no update, no packs."""
        for fn, _, _ in self.pending:
            fn()


_LATE_READY = {}


def late_ready(device):
    """Called on the main stream where every decoder's backward has been joined (the posterior's backward): the postponed
    leaves of run_last may start behind this point."""
    if device in _LATE_CALLS:
        _LATE_READY[device] = torch.cuda.current_stream(device).record_event()


_LOSS_EVENT = {}  # device -> event behind the loss assembly when it ran on the late-leaf stream (ReconLossFn, async_ok)
# MVK_ASYNC_LOSS=0: the loss assembly stays on the caller's stream, between the last forward and the first backward launch
ASYNC_LOSS = _lib.tune("MVK_ASYNC_LOSS", "1") != "0"


def orders_behind_loss(rows):
    """True when the backward node that produced `rows` (a fused decoder tail's NLL row sums) is one of this package's, i.e.
    known to call wait_loss before it reads a row gradient the assembly launch fills.  A user decoder may implement
    `reconstruction_nll` with ordinary autograd ops: its backward would read the row-gradient buffer with NO ordering against an
    assembly launch on the late-leaf stream — in a captured graph a missing dependency, silently wrong gradients.
    Models set spec["async_ok"] only when this holds for every fused term."""
    fn = getattr(rows, "grad_fn", None)
    if fn is None:
        return not getattr(rows, "requires_grad", False)
    return isinstance(fn, (MLPDecoderFn._backward_cls, SVHNDecoderFn._backward_cls))


_LOSS_POSTPONED = {}  # device -> the assembly launch as a closure, while it waits among the postponed leaves (ReconLossFn, assembly_last)
# MVK_ASSEMBLY_LAST=0: the assembly launch at the head of the backward pass on the late-leaf stream (the round-5 place; A/B)
ASSEMBLY_LAST = _lib.tune("MVK_ASSEMBLY_LAST", "1") != "0"


def wait_loss(device):
    """Order the current stream behind the loss assembly launch.  Every backward node that READS a gradient buffer the assembly
    fills (the KL rows' gradients, a fused tail's row gradients on its general path) calls this first; nodes that take the
    constant from `const_grad` do not wait for anything.  An assembly that was postponed to the end of the step (nobody was
    expected to read what it fills) runs HERE, on the caller's stream, when somebody does."""
    fn = _LOSS_POSTPONED.pop(device, None)
    if fn is not None:
        calls = _LATE_CALLS.get(device, [])
        calls[:] = [c for c in calls if c[0] is not fn]
        fn()
        return
    ev = _LOSS_EVENT.get(device)
    if ev is not None:
        torch.cuda.current_stream(device).wait_event(ev)


_CONST_ROWS = {}  # (device, n, value) -> a tensor of n floats holding `value` (made in an eager pass, kept: captured graphs read it)


def const_rows(device, n, value):
    key = (device, int(n), float(value))
    t = _CONST_ROWS.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return None  # never allocate a cached buffer inside a capture (it would live in that graph's pool)
        t = torch.full((int(n),), float(value), dtype=torch.float32, device=device)
        _CONST_ROWS[key] = t
    return t


class deferred_reductions:
    """with deferred_reductions(flat): forward + backward.  On exit the queued finishes run on the current stream; the
    gradient buffer is complete after that (before: NOT).  A no-op on CPU tensors or with MVK_DEFER=0."""

    def __init__(self, flat):
        g = getattr(flat, "grad", None)
        self.on = DEFER and g is not None and g.is_cuda and g.numel() > 0
        self.grad = g

    def __enter__(self):
        if self.on:
            dev = self.grad.device
            arena = _ARENA.get(dev)
            if arena is None:
                arena = torch.empty(DEFER_ARENA_START_FLOATS, dtype=torch.float32, device=dev)
                _ARENA[dev] = arena
            if torch.cuda.is_current_stream_capturing():
                _ARENA_CAPTURED.add(arena.data_ptr())  # a hipGraph holds pointers into it: it must outlive the graph
            call("mvk_defer_begin", ptr(arena), arena.numel(), ptr(self.grad), self.grad.numel())
            _DEFER_ACTIVE.add(dev)
        return self

    def __exit__(self, et, ev, tb):
        if self.on:
            dev = self.grad.device
            late = list(_LATE_CALLS.pop(dev, ()))
            gated = _LATE_GATED.pop(dev, ())
            gate_ev = _LATE_GATE_EV.pop(dev, None)
            if gated and gate_ev is None:
                late += list(gated)  # nobody recorded the gate: ordinary postponed leaves
                gated = ()
            if late or gated:  # postponed leaves: enqueued now, on the late-leaf stream, behind the other late leaves
                # Ordered behind a MAIN-stream event (late_ready: recorded where the posterior's backward starts, i.e. behind the
                # join of every decoder's backward), never behind the producer's branch stream itself: the partial flush on a branch
                # stream waits for the late-leaf stream, a late-leaf stream that waited for that branch stream made the two
                # "parallel capture streams" of each other and hip::Stream::EndCapture recursed until the stack overflowed; a
                # stream of its own changed the queue assignment of the replayed graph (the encoders' backward chain landed
                # behind the postponed launches: 1.10 -> 1.19 ms).
                st = _side_stream(dev, 30)
                ev = _LATE_READY.pop(dev, None)
                st.wait_event(ev if ev is not None else torch.cuda.current_stream(dev).record_event())
                _LATE_USED.setdefault(dev, []).append(st)
                with torch.cuda.stream(st):
                    for fn, reads in late:
                        for t in reads:
                            t.record_stream(st)
                        fn()
                    if gated:
                        st.wait_event(gate_ev)
                        for fn, reads in gated:
                            for t in reads:
                                t.record_stream(st)
                            fn()
            _DEFER_ACTIVE.discard(dev)
            _LATE_READY.pop(dev, None)
            _LOSS_EVENT.pop(dev, None)  # the late-leaf stream is joined below
            _LOSS_POSTPONED.pop(dev, None)
            _BIG_LATE.discard(dev)
            _LATE_GATE_EV.pop(dev, None)
            _LATE_GATED.pop(dev, None)
            cur = torch.cuda.current_stream(dev)
            for st in dict.fromkeys(_LATE_USED.pop(dev, ())):  # the late leaves (below) and sibling flushes end here
                if st != cur:
                    cur.wait_stream(st)
            call("mvk_defer_end", stream_ptr())
            wanted = int(_lib.load().mvk_defer_wanted())
            arena = _ARENA[dev]
            if wanted > arena.numel() and arena.numel() < DEFER_ARENA_CAP_FLOATS and not torch.cuda.is_current_stream_capturing():
                if arena.data_ptr() in _ARENA_CAPTURED:  # only an arena a captured graph points into is kept alive
                    _ARENA_RETIRED.append(arena)
                else:
                    arena.record_stream(cur)  # the finish launch above still reads it: no reuse before that has run
                grown = 1 << (wanted + wanted // 8 - 1).bit_length()  # powers of two: a creeping demand grows it log(n) times
                _ARENA[dev] = torch.empty(min(DEFER_ARENA_CAP_FLOATS, grown), dtype=torch.float32, device=dev)
        return False


# MVK_FLUSH_SIBLING=1: when the LAST backward node of the step starts (the convolutional encoder: ~140 us of launch-latency-bound
# chain on its own stream), everything queued so far is finished on the other branch stream, which is idle by then, instead of
# in one launch behind the chain.
FLUSH_SIBLING = _lib.tune("MVK_FLUSH_SIBLING", "1") != "0"
# MVK_FLUSH_ON_LATE=1: that partial flush on the late-leaf stream, directly behind the large decoder's late weight gradients (A/B)
FLUSH_ON_LATE = _lib.tune("MVK_FLUSH_ON_LATE", "0") == "1"
_BRANCH_SET = {}  # device -> the streams of the last run_branches call (main first)


def defer_flush_sibling(device, node_params=()):
    """node_params: the parameters of the backward node that calls this (the LAST one of the step): with an OverlapPoint open,
    everything else's gradient is final behind the flush, and the point is recorded there."""
    if not (FLUSH_SIBLING and DEFER) or device.type != "cuda" or device not in _DEFER_ACTIVE:
        return
    cur = torch.cuda.current_stream(device)
    sib = next((st for st in _BRANCH_SET.get(device, ()) if st != cur), None)
    if FLUSH_ON_LATE and device in _BIG_LATE and _OVERLAP.get(device) is None:
        # the partial flush has to wait for the decoder's late weight gradients anyway (their slabs are most of what it adds): on
        # THEIR stream it starts the moment they end — beside the encoders' backward-data launches — instead of behind the other
        # encoder's whole backward chain on the sibling stream, where its 165 MB ran beside the last launches of the step's chain
        sib = _side_stream(device, 30)
    if sib is None or _lib.load().mvk_defer_pending() == 0:
        return
    with torch.cuda.stream(sib):
        call("mvk_defer_flush", stream_ptr())
        op = _OVERLAP.get(device)
        if op is not None and not op.recorded:
            # the flush already waits for every stream a gradient producer RAN ON (csrc/igemm.hip defer_flush_locked); ordered
            # explicitly behind the caller's stream and the late-leaf streams as well: a gradient written straight into the
            # buffer by a launch that queued no finish would otherwise be unordered against the collective
            # (... and behind EVERY branch stream of the step, not only the caller's: a backward node of another branch
            # that is enqueued already must have written its gradients before the collective reads them)
            for st in dict.fromkeys([cur] + list(_BRANCH_SET.get(device, ())) + list(_LATE_USED.get(device, ()))):
                if st != sib:
                    sib.wait_event(st.record_event())
            call("mvk_event_record", op.event, 1, stream_ptr())
            op.late_params.extend(node_params)
            op.recorded = True  # from here on every gradient target handed out is late (_grad_target): whatever node asks
    _LATE_USED.setdefault(device, []).append(sib)


def defer_flush_side(device):
    """Inside deferred_reductions: run the finishes queued so far on a side stream, beside whatever follows on the current
    one (called where the decoders' backward is complete and the launch-latency-bound encoder backward begins)."""
    if not DEFER_SIDE or not DEFER or device.type != "cuda" or _lib.load().mvk_defer_pending() == 0:
        return
    with torch.cuda.stream(_side_stream(device, 63)):
        call("mvk_defer_flush", stream_ptr())


# -----------------------------------------------------------------------------------------------------
# modality branches on separate HIP streams
# -----------------------------------------------------------------------------------------------------
# The encoders (and the decoders) of different modalities are independent until the posterior (resp. the
# reconstruction loss) joins them.  The small-modality branch is a string of short, launch-latency-bound
# kernels; on its own stream it runs beside the large modality's convolutions instead of in front of them.
# Autograd replays each node's backward on the stream its forward ran on, so the backward overlaps too.
BRANCH_STREAMS = _lib.tune("MVK_BRANCH_STREAMS", "1") != "0"
_SIDE = {}


def _side_stream(device, i):
    key = (device, i)
    st = _SIDE.get(key)
    if st is None:
        # default priority: a priority -1 side stream costs +75 % step time (1.38 -> 2.4 ms, measured in round 2)
        st = _lib.new_stream(device)  # dedicated: never one of torch's 32 pooled streams (see _lib.new_stream)
        _SIDE[key] = st
    return st


def _tensors_of(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)


# Weight / bias gradients of a launch-latency-bound backward chain (the encoders at the training batch: ~20 launches of
# 5-30 us each) are leaves: on their own stream they could run beside the backward-data chain instead of inside it.
# MEASURED (MoPoE MnistSvhn step, hipGraph replay): 1.435 ms with the SVHN encoder's leaves on a third stream vs 1.385 ms
# without — every fork / join edge between streams of a captured graph costs more than the ~90 us of leaf work it moves
# off the chain (third result of this kind).  Off unless MVK_LEAF_STREAM=1.
LEAF_STREAM = _lib.tune("MVK_LEAF_STREAM", "0") == "1"
# MVK_WGRAD_PAIR=0: the convolutional encoder's two inner weight gradients as two launches inside the backward-data chain (A/B)
WGRAD_PAIR = _lib.tune("MVK_WGRAD_PAIR", "1") != "0"
# MVK_ENC_BWD_TILED (bit 0: the 64 <- 128 layer, bit 1: the 32 <- 64 layer): the convolutional encoder's backward-data launches on
# the tiled engine when register-stationary late leaves of a decoder are in flight (big_late_leaves): a 512-register workgroup
# waits for a CU those leaves have left, <= 128-register ones run beside them
ENC_BWD_TILED = int(_lib.tune("MVK_ENC_BWD_TILED", "0"))


class LeafStream:
    """lf = LeafStream(device); `with lf:` enqueues the enclosed launches on the leaf stream, ordered behind everything
    the current stream holds at that point; `lf.join()` orders the current stream behind the leaf stream (call it before
    the node returns: tensors the leaf launches read may be freed afterwards).  A no-op on the CPU or with
    MVK_LEAF_STREAM=0."""

    def __init__(self, device, index=31):
        self.on = LEAF_STREAM and device.type == "cuda"
        self.device = device
        self.st = _side_stream(device, index) if self.on else None
        self._ctx = None
        self.used = False

    def __enter__(self):
        if self.on:
            cur = torch.cuda.current_stream(self.device)
            self.st.wait_event(cur.record_event())
            self._ctx = torch.cuda.stream(self.st)
            self._ctx.__enter__()
            self.used = True
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
            self._ctx = None
        return False

    def join(self):
        if self.on and self.used:
            torch.cuda.current_stream(self.device).wait_stream(self.st)


# The decoders' side branches are CREATED before the main one (run_branches(side_first=True)), so that autograd enqueues the main
# branch's backward first and a side branch's backward can be ordered behind the main branch's HBM-bound tail kernel
# (tail_bwd_done / wait_tail_bwd).  Two memory-heavy launches that share the chip are slower than the same two in a row:
# small_up_bwd_bf_kernel took 139 us beside the MLP decoder's backward-data GEMM (95 us), 82 and 32 us alone.  MEASURED (three
# same-box pairs): 1.087 / 1.080 / 1.095 ms -> 1.040 / 1.055 / 1.050 ms.  MVK_TAIL_FIRST=0: the round-3 order.
TAIL_FIRST = _lib.tune("MVK_TAIL_FIRST", "1") != "0"
_TAIL_EVENT = {}


def tail_bwd_done(device):
    """SVHNDecoderFn.backward, right behind its image-layer launch."""
    if TAIL_FIRST and device in _DEFER_ACTIVE:
        _TAIL_EVENT[device] = torch.cuda.current_stream(device).record_event()


def wait_tail_bwd(device):
    """A side branch's decoder backward: start behind the main branch's tail kernel when that one was enqueued first."""
    ev = _TAIL_EVENT.pop(device, None)
    if ev is not None:
        torch.cuda.current_stream(device).wait_event(ev)


# A side branch's forward LAUNCHES may be postponed behind a point of the main branch's forward (its node is created first all the
# same): the MLP decoder's GEMMs read ~200 MB through L2 / the fabric and slow the 128 <-> 64-channel convolution of the SVHN
# decoder (4 workgroup types, 2-2.5x fabric traffic of its own) far more than its 64 <-> 32-channel one.  MEASURED (three same-box
# pairs): 1.060 / 1.034 / 1.048 ms postponed behind the first convolution vs 1.047 / 1.051 / 1.042 ms launched where the node is
# created — no difference in the forward pass; MVK_FWD_DEFER=1 enables it.
FWD_DEFER_AT = int(_lib.tune("MVK_FWD_DEFER", "0"))  # 1: behind the 128 -> 64 layer; 2: behind the 64 -> 32 layer; 3: behind the fused tail;
# 4: behind the FIRST layer, the side branch's z-independent preparation included
FWD_DEFER = FWD_DEFER_AT > 0
_FWD_DEFERRED = {}  # device -> [(stream, closure)], only while run_branches(side_first=True) builds the side branches
_FWD_DEFER_OPEN = set()


def defer_forward(device, fn):
    """Inside a side branch of run_branches(side_first=True): queue fn (launches only: every tensor is allocated already) until
    the main branch calls flush_deferred_forward, or the branches are joined.  False: the caller launches now."""
    if not FWD_DEFER or device not in _FWD_DEFER_OPEN or TAPS is not None:
        return False
    _FWD_DEFERRED.setdefault(device, []).append((torch.cuda.current_stream(device), fn))
    return True


def flush_deferred_forward(device):
    """Main branch: the postponed side launches may start behind what this stream holds now."""
    todo = _FWD_DEFERRED.pop(device, None)
    if not todo:
        return
    ev = torch.cuda.current_stream(device).record_event()
    for st, fn in todo:
        st.wait_event(ev)
        with torch.cuda.stream(st):
            fn()


_PRELUDE = {}  # device -> closure with launches that depend on nothing of the step (run at the head of the first side branch)


def set_prelude(device, fn):
    """Launches that the step needs LATER and that depend on nothing it computes (the optimizer's scalar preparation inside a
    captured step): enqueued at the head of the first side branch run_branches opens — a stream with slack, joined long before
    the optimizer — instead of on the critical chain.  run_prelude(device) runs it where it is if no branch took it."""
    _PRELUDE[device] = fn


def run_prelude(device):
    fn = _PRELUDE.pop(device, None)
    if fn is not None:
        fn()


# MVK_BRANCH_MAX=n: at most n streams for the modality branches (the caller's + n - 1 side streams; branches beyond that share
# the side streams round-robin, in order).  Five PolyMNIST stacks on five streams trample each other (cfg4: c3rs_kernel 188 us
# average, 1326 us maximum); 0 = one stream per branch.
BRANCH_MAX = int(_lib.tune("MVK_BRANCH_MAX", "0"))


def _branch_stream(device, i, n):
    """The stream of branch i (i >= 1) of n."""
    if BRANCH_MAX >= 2 and n > BRANCH_MAX:
        i = 1 + (i - 1) % (BRANCH_MAX - 1)
    return _side_stream(device, i)


def run_branches(names, fn, device, side_first=False):
    """{m: fn(m)} with every branch but the first on its own stream; joined before returning."""
    names = list(names)
    if not BRANCH_STREAMS or len(names) < 2 or device.type != "cuda":
        return {m: fn(m) for m in names}
    main = torch.cuda.current_stream(device)
    _BRANCH_SET[device] = [main] + list(dict.fromkeys(_branch_stream(device, i, len(names)) for i in range(1, len(names))))
    fork = main.record_event()
    # The first branch stays on the caller's stream and is enqueued FIRST: autograd runs backward nodes in reverse
    # creation order, so the side branches' backward is enqueued (and, in a captured graph, ordered) before the long
    # backward of the main branch instead of behind it.
    outs = {}
    if not (side_first and TAIL_FIRST):
        outs[names[0]] = fn(names[0])
    else:
        _FWD_DEFER_OPEN.add(device)
    sides = []
    for i, m in enumerate(names[1:], start=1):
        st = _branch_stream(device, i, len(names))
        if st not in sides:
            st.wait_event(fork)
            sides.append(st)
        with torch.cuda.stream(st):
            run_prelude(device)
            outs[m] = fn(m)
        for t in _tensors_of(outs[m]):
            t.record_stream(main)
    if names[0] not in outs:
        _FWD_DEFER_OPEN.discard(device)  # the main branch itself never defers
        try:
            outs[names[0]] = fn(names[0])
        finally:
            flush_deferred_forward(device)  # nobody asked for them earlier: now
    for st in sides:
        main.wait_stream(st)
    return {m: outs[m] for m in names}


# =====================================================================================================
# thin launch helpers (no autograd)
# =====================================================================================================
def linear_fwd(x2, w, b, act):
    M, K = x2.shape
    N = w.shape[0]
    y = _new((M, N), x2)
    ws = _ws(x2)
    call("mvk_linear_fwd", ptr(x2), ptr(w), ptr(b), ptr(y), M, N, K, act, ptr(ws), ws.numel(), stream_ptr())
    return y


def _bias_target(bias_param):
    """(buffer the kernel accumulates the bias gradient into, what autograd gets for the bias)."""
    if bias_param is None:
        return None, None
    return _grad_target(bias_param)


def linear_bwd_data(dy, w, y_out=None, y_act=NONE, prev_out=None, prev_act=NONE, out=None, accumulate=False,
                    prev_bias=None):
    """dx (the previous layer's pre-activation gradient when prev_out is given).  prev_bias: that layer's bias
    parameter — its gradient (column sums of dx) is produced by the same launch; returns (dx, grad for autograd)."""
    M, N = dy.shape
    K = w.shape[1]
    dx = out if out is not None else _new((M, K), dy)
    ws = _ws(dy)
    tb, rb = _bias_target(prev_bias)
    call("mvk_linear_bwd_data", ptr(dy), ptr(w), ptr(dx), M, N, K, ptr(y_out), y_act, ptr(prev_out), prev_act,
         1 if accumulate else 0, ptr(tb), ptr(ws), ws.numel(), stream_ptr())
    return dx if prev_bias is None else (dx, rb)


def _grad_target(p):
    """Where a parameter's gradient is accumulated.  If the parameter already owns a contiguous fp32 .grad
    (e.g. a view of the flat gradient buffer of trainers.FlatParams) the kernels accumulate straight into it
    and autograd gets None for this input (no zero-fill, no extra add kernel); otherwise a fresh zero buffer is
    returned to autograd as usual.  Both ways the visible semantics are `p.grad += dL/dp`."""
    g = p.grad
    if _OVERLAP:
        op = _OVERLAP.get(p.device)
        if op is not None and op.recorded:
            # a gradient produced behind the overlap point — by the node that recorded it or by ANY node enqueued
            # later (a second SVHN-type encoder, an encoder on a side branch) — is not part of the early collective
            op.late_params.append(p)
    if DIRECT_GRAD and g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.shape == p.shape \
            and g.device == p.device:
        return g, None
    z = _zeros(p.shape, p)
    return z, z


def _is_direct(p):
    """_grad_target(p) accumulates straight into p.grad (autograd gets None)."""
    g = p.grad
    return bool(DIRECT_GRAD and g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.shape == p.shape
                and g.device == p.device)


def linear_bwd_weight(dy, x2, w, b, y_out=None, y_act=NONE):
    """-> (grad to return for w, grad to return for b)"""
    M, N = dy.shape
    K = x2.shape[1]
    dw, rw = _grad_target(w)
    db, rb = _grad_target(b) if b is not None else (None, None)
    ws = _ws(dy)
    call("mvk_linear_bwd_weight", ptr(dy), ptr(x2), ptr(dw), ptr(db), M, N, K, ptr(y_out), y_act, ptr(ws),
         ws.numel(), stream_ptr())
    return rw, rb


def gemm(a, b, M, N, K, ta=False, tb=False, bias=None, bias_mod=0, act=NONE, out=None, accumulate=False,
         a_act_src=None, a_act=NONE, c_act_src=None, c_act=NONE):
    c = out if out is not None else _new((M, N), a)
    ws = _ws(a)
    call("mvk_gemm", ptr(a), ptr(b), ptr(c), M, N, K, int(ta), int(tb), ptr(bias), bias_mod, act,
         1 if accumulate else 0, ptr(a_act_src), a_act, ptr(c_act_src), c_act, ptr(ws), ws.numel(), stream_ptr())
    return c


def heads_fwd(h2, w_mu, b_mu, w_lv, b_lv, N, w_sk, w_sn):
    """(mu, lv) = h2 @ W_mu + b_mu, h2 @ W_lv + b_lv in one launch (mvk_heads_fwd); W(k, n) = W[k * w_sk + n * w_sn]."""
    M, K = h2.shape
    if N > 32 or K % 4 != 0:
        return None
    mu, lv = _new((M, N), h2), _new((M, N), h2)
    call("mvk_heads_fwd", ptr(h2), ptr(w_mu), ptr(b_mu), ptr(mu), ptr(w_lv), ptr(b_lv), ptr(lv), M, N, K, w_sk, w_sn,
         stream_ptr())
    return mu, lv


# MVK_NARROW_DZ=1: the decoders' gradient into the latent (N = L <= 32 outputs from K = 512 / 2048 over the decoder batch) as ONE
# exact-fp32 launch that reads its input once (narrow_linear / narrow_fwd_kernel) instead of a split-K launch of the tiled engine +
# its reduce.  MEASURED (four alternating pairs): 0.9382 / 0.9361 / 0.9307 / 0.9317 ms with, 0.9361 / 0.9288 / 0.9293 / 0.9271
# without (+0.4 %): the launch takes 21.5 / 28.7 us in the step beside the last decoder launches, no less than the pair it
# replaces (18.5 + 4.5 / 18.5 + 9).  Off.
NARROW_DZ = _lib.tune("MVK_NARROW_DZ", "0") == "1"


def narrow_linear(x2, w, M, N, K, w_sk, w_sn):
    """y[M, N] = x2[M, K] W with W(k, n) = w[k * w_sk + n * w_sn], 16 < N <= 32, M >= 1024 rows: ONE launch that reads x2 once
    (mvk_heads_fwd's one-head, many-rows form: narrow_fwd_kernel, exact fp32, fixed summation order).  None: not covered."""
    if not NARROW_DZ or not (16 < N <= 32) or K % 4 != 0 or M < 1024 or x2.data_ptr() % 16 != 0 or not x2.is_cuda:
        return None
    y = _new((M, N), x2)
    call("mvk_heads_fwd", ptr(x2), ptr(w), None, ptr(y), None, None, None, M, N, K, w_sk, w_sn, stream_ptr())
    return y


HEADS_BWD = _lib.tune("MVK_HEADS_BWD", "1") != "0"  # A/B switch: the heads' backward in one launch
# MEASURED (headline step, one box, 3-4 rounds each): 1.355 ms with the six separate launches, 1.326 ms with the fused launch
# for the convolutional (SVHN) encoder only, 1.343 ms with the fused launch for the MLP encoder too (its chain is not the
# critical one and the fused launch delays the other stream's kernels).  RE-MEASURED at the end of round 2, with the partial
# finish on the sibling stream and a replayed graph that now spreads over three hardware queues: 1.2480 ms with the fused launch
# for the MLP encoder too vs 1.2569 ms without (four same-box pairs) — on by default now, MVK_HEADS_BWD_MLP=0 disables.
# RE-MEASURED in round 6, with the loss assembly at the end of the step and the convolutional encoder's weight-gradient pair: at the
# headline (K = 10: the large decoder's register-stationary weight gradients are late leaves of the tail, one 512-register wave
# per SIMD on every CU) the six separate launches for the MLP encoder are ahead again — 0.9262 / 0.9299 / 0.9272 ms against
# 0.9407 / 0.9370 / 0.9361 with the fused launch, second box 0.9565 / 0.9615 / 0.9606 against 0.9638 / 0.9713 / 0.9708, third
# 0.9487 / 0.9558 / 0.9508 against 0.9574 / 0.9612 / 0.9587 (-0.9 %, nine alternating pairs: the fused launch's 128 x 4
# workgroups with 48-62 KB of LDS wait for room beside those leaves) — while WITHOUT such leaves (decoder batch < 1024 rows:
# MMVAE MnistSvhn 0.516 / 0.533 fused against 0.560 / 0.559, MoPoE at K = 1 0.509 / 0.521 against 0.537 / 0.540) the fused launch
# wins by 5-8 %.  "auto" (default): fused unless a backward node of this scope enqueued register-stationary late leaves
# (big_late_leaves); MVK_HEADS_BWD_MLP=0 / 1 force one form.
_HEADS_BWD_MLP_MODE = _lib.tune("MVK_HEADS_BWD_MLP", "auto")
_BIG_LATE = set()  # devices on which a backward node of the open deferred_reductions scope enqueued register-stationary late leaves


def big_late_leaves(device):
    _BIG_LATE.add(device)


def heads_bwd_mlp(device):
    """Whether an MLP encoder's heads take the one-launch backward now."""
    if _HEADS_BWD_MLP_MODE in ("0", "1"):
        return _HEADS_BWD_MLP_MODE == "1"
    return device not in _BIG_LATE


def heads_bwd(x, x_act, dys, ws_, bs, w_sk, w_sn, flat_c=0, want_dx=True, prev_bias=None, dw_params=None):
    """Backward of 1 or 2 narrow heads that read x [M,K] (mvk_heads_bwd).  dys: gradients [M,N]; ws_: the weights as the
    kernel reads them (W(k, n) = W[k * w_sk + n * w_sn]); dw_params: the parameters whose .grad receives dY^T x (default:
    ws_); bs: bias parameters (or None).  Returns None when the shape is not covered, else
    (dx or None, [grad for autograd per weight], [grad per bias], grad for prev_bias)."""
    M, K = x.shape
    N = dys[0].shape[1]
    if not HEADS_BWD or N > 32 or K % 16 != 0 or len(dys) > 2 or M < 1 or (M + 127) // 128 > 64:
        return None
    dw_params = ws_ if dw_params is None else dw_params
    tw, rw = zip(*[_grad_target(p) for p in dw_params])
    tb, rb = zip(*[_grad_target(b) if b is not None else (None, None) for b in bs])
    tp, rp = _grad_target(prev_bias) if prev_bias is not None else (None, None)
    dx = _new((M, K), x) if want_dx else None
    two = len(dys) == 2
    ws = _ws(x)
    # slabs of the launch: N * K * ceil(M / 128) floats per head + the bias rows; what does not fit the deferred arena must fit
    # the per-stream scratch, else the separate launches take over (the C side declines with MVK_EINVAL)
    rgs = (M + 127) // 128
    if (len(dys) * (N * K + N) + K) * rgs + 64 > ws.numel():
        return None
    call("mvk_heads_bwd", ptr(x), x_act, ptr(dys[0]), ptr(dys[1]) if two else None, ptr(ws_[0]), ptr(ws_[1]) if two else None,
         w_sk, w_sn, flat_c, ptr(dx), ptr(tw[0]), ptr(tw[1]) if two else None, ptr(tb[0]), ptr(tb[1]) if two else None,
         ptr(tp), M, N, K, ptr(ws), ws.numel(), stream_ptr())
    return dx, list(rw), list(rb), rp


def colsum(dy2, b, y_out=None, y_act=NONE):
    """bias gradient: b.grad += column sums of dy2 [M,N]; returns what autograd should get for b."""
    M, N = dy2.shape
    db, rb = _grad_target(b)
    ws = _ws(dy2)
    call("mvk_colsum_acc", ptr(dy2), ptr(y_out), y_act, ptr(db), M, N, ptr(ws), ws.numel(), stream_ptr())
    return rb


def pack_conv(wref, want_down=True, want_up=True):
    """wref [Cv][Cu][4][4] -> (Wdown [16*Cu, Cv], Wup [4, 4*Cv, Cu])."""
    (wd, wu), = pack_weights([(wref, want_down, want_up)])
    return wd, wu


def _frag(like, Cu, Cv):
    """Buffer for the bf16-piece fragment pack of a 4x4/stride-2 layer the register-stationary kernels cover, else None."""
    nb = _lib.load().mvk_imgconv_frag_bytes(Cu, Cv)
    return torch.empty(nb, dtype=torch.uint8, device=like.device) if nb else None


def wfrag(w):
    """The fragment pack that travels with a packed weight (attached by pack_weights), or None."""
    return getattr(w, "mvk_frag", None)


PACK_MAX = 16  # MVK_PACK_MAX descriptors per launch
# MVK_PREPACK=0: every network packs its own weights when it runs (one launch per network and forward pass)
PREPACK = _lib.tune("MVK_PREPACK", "1") != "0"
# MVK_PACK_SIDE=1: the pack launch on the first branch stream, beside the image-consuming first convolution (which reads the
# reference weight layout).  MEASURED (three same-box pairs): 1.074 / 1.079 / 1.074 ms without vs 1.137 / 1.130 / 1.131 ms with —
# one more fork / join edge at the head of the captured graph costs three times the 20 us it takes off the chain (the fourth result
# of this kind: DESIGN.md section 9).  Off.
PACK_SIDE = _lib.tune("MVK_PACK_SIDE", "0") == "1"
_PACK_SCOPE = None


class pack_scope:
    """with pack_scope(model): ONE weight-pack launch for the whole forward pass.  The first `pack_weights` call inside the
    scope also packs what every other sub-module of `model` announces through `pack_jobs()` (the decoder's pack used to sit on
    the critical chain between the posterior and the first decoder layer); later calls find their packs in the scope's cache.
    The cache dies with the scope: nothing packed here outlives the forward pass (the weights change at the next optimizer step)."""

    def __init__(self, model):
        self.model, self.own = model, False

    def __enter__(self):
        global _PACK_SCOPE
        if PREPACK and _PACK_SCOPE is None:
            _PACK_SCOPE = {"model": self.model, "cache": {}, "done": False}
            self.own = True
            # the zeroed arena of the amax protocol (AmaxPool, the pack launches): filled HERE, on the caller's stream and in front
            # of every fork of run_branches — a slot must be zero before ANY stream publishes into it, and the branch streams
            # are ordered behind the fork only, not behind what the first branch enqueues
            dev = next((p.device for p in self.model.parameters()), None)
            if (C3_F16 or IMG_F16) and dev is not None and dev.type == "cuda":
                _PACK_SCOPE["amax"], _PACK_SCOPE["amax_pos"] = torch.zeros(512, dtype=torch.float32, device=dev), 0
            if PACK_SIDE and BRANCH_STREAMS and dev is not None and dev.type == "cuda":
                self._launch_beside(dev)
        if _ROTATE_ARMED:  # a rotated step's head branch forks HERE, behind the first launch of the main chain
            for r in list(_ROTATE_ARMED.values()):
                r.begin_if_armed()
        return self

    def _launch_beside(self, dev):
        """The step's ONE weight-pack launch on the first branch stream, beside the first kernels of the forward pass (the image-
        consuming first convolution reads the reference layout: `mvk_conv4s2_small_down_fwd_wref`), instead of in front of
        them on the caller's stream: 20 us of the launch-latency-bound head of the step.  A stream that takes a pack out of the
        cache orders itself behind the launch once (`pack_weights`)."""
        sc = _PACK_SCOPE
        jobs, keys = [], []
        for mod in self.model.modules():
            announce = getattr(mod, "pack_jobs", None)
            if announce is None:
                continue
            for j in announce():
                k = _job_key(j)
                if k not in keys and j[0].device == dev:
                    jobs.append(j)
                    keys.append(k)
        sc["done"] = True
        if not jobs:
            return
        main = torch.cuda.current_stream(dev)
        st = _side_stream(dev, 1)
        st.wait_event(main.record_event())  # behind the optimizer step that wrote the weights, and the amax arena's fill
        with torch.cuda.stream(st):
            outs = []
            for i0 in range(0, len(jobs), PACK_MAX):
                outs += _pack_launch(jobs[i0:i0 + PACK_MAX])
            sc["event"] = st.record_event()
        sc["pack_stream"], sc["waited"] = st, {st.cuda_stream}
        for k, o in zip(keys, outs):
            sc["cache"][k] = o

    def __exit__(self, *exc):
        global _PACK_SCOPE
        if self.own:
            _PACK_SCOPE = None
        return False


def _job_key(job):
    return (job[0].data_ptr(), tuple(job[0].shape)) + tuple(job[1:])


def pack_weights(jobs):
    """`_pack_launch(jobs)`, or inside a pack_scope: from the scope's cache / together with every announced job of the model."""
    sc = _PACK_SCOPE
    keys = [_job_key(j) for j in jobs]
    rot = rotation(jobs[0][0].device)
    if rot is not None and rot.cache:  # some (or all) of them packed by the head branch of a rotated step, behind their update
        hit = [k in rot.cache for k in keys]
        if any(hit):
            rot.wait()
            if all(hit):
                return [rot.cache[k] for k in keys]
            rest = iter(pack_weights([j for j, h in zip(jobs, hit) if not h]))
            return [rot.cache[k] if h else next(rest) for k, h in zip(keys, hit)]
    if sc is None:
        return _pack_launch(jobs)
    cache = sc["cache"]
    if all(k in cache for k in keys):
        ev = sc.get("event")
        if ev is not None:  # packed on another stream (pack_scope._launch_beside): this stream waits for it once
            cur = torch.cuda.current_stream(jobs[0][0].device)
            if cur.cuda_stream not in sc["waited"]:
                cur.wait_event(ev)
                sc["waited"].add(cur.cuda_stream)
                for t in _tensors_of([cache[k] for k in cache if not (isinstance(k, tuple) and k and k[0] in ("dense16", "dense16_xamax"))]):
                    t.record_stream(cur)
                    if getattr(t, "mvk_frag", None) is not None:
                        t.mvk_frag.record_stream(cur)
        return [cache[k] for k in keys]
    todo, tkeys = list(jobs), list(keys)
    if not sc["done"]:
        sc["done"] = True
        for mod in sc["model"].modules():
            announce = getattr(mod, "pack_jobs", None)
            if announce is None:
                continue
            for j in announce():
                k = _job_key(j)
                if rot is not None and k in rot.cache:
                    continue  # behind the rotated update of these weights, on the head branch
                if k not in tkeys and k not in cache and j[0].device == jobs[0][0].device and len(todo) < PACK_MAX:
                    todo.append(j)
                    tkeys.append(k)
    outs = []
    for i0 in range(0, len(todo), PACK_MAX):
        outs += _pack_launch(todo[i0:i0 + PACK_MAX])
    for k, o in zip(tkeys, outs):
        cache[k] = o
    return outs[:len(jobs)]


def _pack_launch(jobs, am=None):
    """All weight packs of a network in ONE launch.  jobs: list of (wref, want_down, want_up) for 4x4/stride-2
    layers ([Cv][Cu][4][4] -> (Wdown [16*Cu, Cv], Wup [4, 4*Cv, Cu])) or (wref, "unflatten") for the 1x1-spatial
    transposed convolution ([Cin][Cout][4][4] -> [Cin, 16*Cout]).  Returns the packed tensors in job order."""
    descs = (PackDesc * len(jobs))()
    outs = []
    # max |W| per convolution weight: the operand scale of the scaled-fp16 launches (mvk_conv3x3_s, mvk_conv4s2_down_s / _up_s)
    # (am given: zeroed slots of the caller's — the head branch of a rotated step packs outside the scope's arena)
    if am is None and ((C3_F16 and any(job[1] == "c3" for job in jobs))
                       or (IMG_F16 and any(job[1] not in ("c3", "unflatten") for job in jobs))):
        am = _amax_slots(jobs[0][0], len(jobs), create=True)
    for i, job in enumerate(jobs):
        wref = job[0]
        Cv, Cu = wref.shape[0], wref.shape[1]
        d = descs[i]
        d.Wref, d.Cv, d.Cu, d.ld_down, d.col_off = wref.data_ptr(), Cv, Cu, Cv, 0
        if job[1] == "unflatten":
            wp = _new((Cv, 16 * Cu), wref)
            d.kind, d.Wdown, d.Wup = 1, None, wp.data_ptr()
            outs.append(wp)
        elif job[1] == "c3":  # 3x3 convolution [Cout][Cin][3][3] -> (forward [9*Cin, Cout], backward-data [9*Cout, Cin])
            wf = _new((9 * Cu, Cv), wref) if job[2] else None
            wb = _new((9 * Cv, Cu), wref) if job[3] else None
            d.kind = 2
            d.Wdown = wf.data_ptr() if wf is not None else None
            d.Wup = wb.data_ptr() if wb is not None else None
            if am is not None:
                d.amax = am[i:i + 1].data_ptr()
                for t in (wf, wb):
                    if t is not None:
                        t.mvk_amax = am[i:i + 1]
            outs.append((wf, wb))
        else:
            wd = _new((16 * Cu, Cv), wref) if job[1] else None
            wu = _new((4, 4 * Cv, Cu), wref) if job[2] else None
            d.kind = 0
            d.Wdown = wd.data_ptr() if wd is not None else None
            d.Wup = wu.data_ptr() if wu is not None else None
            # bf16-piece MFMA fragments for the register-stationary kernels travel with the packs (same launch)
            if wd is not None:
                wd.mvk_frag = _frag(wref, Cu, Cv)
                d.Fdown = wd.mvk_frag.data_ptr() if wd.mvk_frag is not None else None
            if wu is not None:
                wu.mvk_frag = _frag(wref, Cu, Cv)
                d.Fup = wu.mvk_frag.data_ptr() if wu.mvk_frag is not None else None
            if am is not None:
                d.amax = am[i:i + 1].data_ptr()
                for t in (wd, wu):
                    if t is not None:
                        t.mvk_amax = am[i:i + 1]
            outs.append((wd, wu))
    call("mvk_pack_weights", descs, len(jobs), stream_ptr())
    return outs


def conv4s2_scaled_ok(n, h, w, Cu, Cv):
    """True when the `amax=` forms of conv_down / conv_up (mvk_conv4s2_down_s / _up_s) take this layer at this batch."""
    return bool(_lib.load().mvk_conv4s2_scaled_ok(n, h, w, Cu, Cv))


def conv_down(U, wdown, bias, n, h, w, Cu, Cv, act=NONE, u_nchw=False, u_act_src=None, u_act=NONE,
              v_act_src=None, v_act=NONE, out_bias=None, in_bf3=False, frag=None, amax=None, out=None):
    """out_bias: bias parameter whose gradient is the per-channel sum of the result (backward-data use): fused into
    the launch; returns (V, grad for autograd) then.  amax = (x_amax, w_amax, y_amax): the amax protocol (mvk_conv4s2_down_s;
    x and w bounds given = scaled fp16 pairs, y_amax = zeroed slot that receives max |V|).  out: the buffer V is written to."""
    V = out if out is not None else torch.empty((n, h, w, Cv), dtype=torch.float32, device=U.device)
    ws = _ws(V)
    tb, rb = _bias_target(out_bias)
    if amax is not None:
        if u_nchw or u_act_src is not None or in_bf3:
            raise _lib.MvkError("conv_down: the amax form takes NHWC fp32 input without a fused input activation")
        call("mvk_conv4s2_down_s", ptr(U), ptr(wdown), ptr(bias), ptr(V), n, h, w, Cu, Cv, act, ptr(v_act_src), v_act, ptr(tb),
             ptr(amax[0]), ptr(amax[1]), ptr(amax[2]), ptr(ws), ws.numel(), ptr(frag if frag is not None else wfrag(wdown)),
             stream_ptr())
        return V if out_bias is None else (V, rb)
    call("mvk_conv4s2_down", ptr(U), ptr(wdown), ptr(bias), ptr(V), n, h, w, Cu, Cv, act, int(u_nchw),
         ptr(u_act_src), u_act, ptr(v_act_src), v_act, ptr(tb), ptr(ws), ws.numel(), FMT_IN_BF3 if in_bf3 else 0,
         ptr(frag if frag is not None else wfrag(wdown)), stream_ptr())
    return V if out_bias is None else (V, rb)


FMT_TILED = 2  # mvk.h MVK_FMT_TILED


def conv_up(V, wup, bias, n, h, w, Cu, Cv, act=NONE, u_nchw=False, u_act_src=None, u_act=NONE, out_bias=None,
            in_bf3=False, frag=None, amax=None, out=None, tiled=False):
    U = out if out is not None else torch.empty((n, Cu, 2 * h, 2 * w) if u_nchw else (n, 2 * h, 2 * w, Cu), dtype=torch.float32,
                                                device=V.device)
    ws = _ws(U)
    tb, rb = _bias_target(out_bias)
    if amax is not None:  # (x_amax, w_amax, y_amax): see conv_down
        if u_nchw or in_bf3:
            raise _lib.MvkError("conv_up: the amax form writes NHWC and reads fp32")
        call("mvk_conv4s2_up_s", ptr(V), ptr(wup), ptr(bias), ptr(U), n, h, w, Cu, Cv, act, ptr(u_act_src), u_act, ptr(tb),
             ptr(amax[0]), ptr(amax[1]), ptr(amax[2]), ptr(ws), ws.numel(), ptr(frag if frag is not None else wfrag(wup)),
             stream_ptr())
        return U if out_bias is None else (U, rb)
    call("mvk_conv4s2_up", ptr(V), ptr(wup), ptr(bias), ptr(U), n, h, w, Cu, Cv, act, int(u_nchw),
         ptr(u_act_src), u_act, ptr(tb), ptr(ws), ws.numel(), (FMT_IN_BF3 if in_bf3 else 0) | (FMT_TILED if tiled else 0),
         ptr(frag if frag is not None else wfrag(wup)), stream_ptr())
    return U if out_bias is None else (U, rb)


LEAKY = ACT["leaky_relu_0.2"]


# MVK_C3_Y_AMAX=0: the image-side 3x3 launches do not publish max |Y| (an mvk_amax pass computes it where needed; A/B)
C3_Y_AMAX = _lib.tune("MVK_C3_Y_AMAX", "1") != "0"


def conv3x3(X, wpack, bias, n, H, W, Cin, Cout, act=NONE, y_act_src=None, y_src_act=NONE, out_bias=None, res=None,
            res_alpha=1.0, y_amax=None):
    """3x3/1/1 convolution on NHWC (forward with the forward pack; backward data with the backward pack and Cin/Cout
    swapped, y_act_src = the activation whose derivative multiplies the result, out_bias = the bias whose gradient is
    the channel sum of the result, res: the result becomes res + res_alpha * result in the same pass)."""
    Y = _new((n, H, W, Cout), X)
    ws = _ws(X)
    if res is not None:
        if out_bias is not None:
            raise _lib.MvkError("conv3x3: a residual and a fused bias gradient are exclusive")
        call("mvk_conv3x3_res", ptr(X), ptr(wpack), ptr(bias), ptr(Y), n, H, W, Cin, Cout, act, ptr(y_act_src), y_src_act,
             ptr(res), float(res_alpha), ptr(ws), ws.numel(), stream_ptr())
        return Y
    tb, rb = _bias_target(out_bias)
    if y_amax is not None:  # image on the input side + published maximum (mvk_conv3x3_y; MvkError when the shape is not covered)
        call("mvk_conv3x3_y", ptr(X), ptr(wpack), ptr(bias), ptr(Y), n, H, W, Cin, Cout, act, ptr(y_act_src), y_src_act,
             ptr(tb), ptr(y_amax), ptr(ws), ws.numel(), stream_ptr())
        return Y if out_bias is None else (Y, rb)
    call("mvk_conv3x3", ptr(X), ptr(wpack), ptr(bias), ptr(Y), n, H, W, Cin, Cout, act, ptr(y_act_src), y_src_act,
         ptr(tb), ptr(ws), ws.numel(), stream_ptr())
    return Y if out_bias is None else (Y, rb)


def conv3x3_wgrad(X, dY, wparam, n, H, W, Cin, Cout):
    dw, rw = _grad_target(wparam)
    ws = _ws(X)
    call("mvk_conv3x3_wgrad", ptr(X), ptr(dY), ptr(dw), n, H, W, Cin, Cout, ptr(ws), ws.numel(), stream_ptr())
    return rw


def conv3x3_fused_ok(n, H, W, Cin, Cout):
    """True when mvk_conv3x3_f / mvk_conv3x3_wgrad_f (the register-stationary kernels) take this problem: the elementwise
    passes of a ResnetBlock are then folded into the convolutions (ResnetStackFn)."""
    return bool(_lib.load().mvk_conv3x3_fused_ok(n, H, W, Cin, Cout))


def conv3x3_f(X, wpack, bias, n, H, W, Cin, Cout, act=NONE, y_act_src=None, y_src_act=NONE, res=None, res_alpha=1.0,
              out_bias=None, x_act=NONE, pre_scale=1.0):
    """Y = [res + res_alpha *] (act(pre_scale * conv(x_act(X)) + bias) * y_src_act'(y_act_src)); out_bias as in conv3x3."""
    Y = _new((n, H, W, Cout), X)
    ws = _ws(X)
    tb, rb = _bias_target(out_bias)
    call("mvk_conv3x3_f", ptr(X), ptr(wpack), ptr(bias), ptr(Y), n, H, W, Cin, Cout, act, ptr(y_act_src), y_src_act, ptr(res),
         float(res_alpha), ptr(tb), x_act, float(pre_scale), ptr(ws), ws.numel(), stream_ptr())
    return Y if out_bias is None else (Y, rb)


# ---- scaled-fp16 form of the register-stationary 3x3 kernels (csrc/bf3.hpp: 3 MFMAs per product instead of 6) --------------
# MVK_C3_F16=0 (under MVK_TUNE=1) keeps every 3x3 convolution on the bf16-piece kernels.
C3_F16 = _lib.tune("MVK_C3_F16", "1") != "0"
# the same product form for the 4x4 / stride-2 layers of the SVHN decoder (csrc/imgconv.hip NP = 2); MVK_IMG_F16=0: bf16 pieces
IMG_F16 = _lib.tune("MVK_IMG_F16", "1") != "0"
# decoder rows from which SVHNDecoderFn takes that form (below, the in-kernel conversion of the weights costs what the cheaper
# product saves); tests set 1 (with debug flag 0x200) so the small goldens meet these kernels too
IMG_F16_MIN_ROWS = 1024


def _amax_slots(like, n, create=False):
    """n zeroed device scalars.  Inside a pack_scope they are carved from the ONE arena the scope filled when it was entered
    (in front of every stream fork of the forward pass): one fill per forward pass instead of one per pool, all of them on
    the step's critical chain.  Outside a scope (backward passes, PREPACK off): a fill on the current stream."""
    sc = _PACK_SCOPE
    if sc is not None:
        ar = sc.get("amax")
        if ar is not None and ar.device == like.device and sc["amax_pos"] + n <= ar.numel():
            sc["amax_pos"] += n
            return ar[sc["amax_pos"] - n:sc["amax_pos"]]
    return torch.zeros(n, dtype=torch.float32, device=like.device)


class AmaxPool:
    """Zeroed device scalars for the `amax` protocol (one fill launch per pool): a launch publishes max |Y| into a slot by
    atomic max, the launch that consumes Y takes its operand scale from it.  A slot is used by ONE producer."""

    def __init__(self, like, slots):
        self.t = _amax_slots(like, slots)
        self.i = 0

    def take(self):
        if self.i >= self.t.numel():
            self.t, self.i = torch.zeros_like(self.t), 0
        self.i += 1
        return self.t[self.i - 1:self.i]


def amax_of(x, slot):
    """slot <- max(slot, max |x|) (mvk_amax); returns the slot."""
    call("mvk_amax", ptr(x), x.numel(), ptr(slot), stream_ptr())
    return slot


def conv3x3_scaled_ok(n, H, W, Cin, Cout):
    """True when mvk_conv3x3_s takes this problem."""
    return C3_F16 and bool(_lib.load().mvk_conv3x3_scaled_ok(n, H, W, Cin, Cout))


def conv3x3_s(X, wpack, bias, n, H, W, Cin, Cout, x_amax, w_amax, y_amax=None, act=NONE, y_act_src=None, y_src_act=NONE,
              res=None, res_alpha=1.0, out_bias=None, x_act=NONE, pre_scale=1.0):
    """conv3x3_f on scaled fp16 pairs: x_amax / w_amax = device scalars bounding max |X| / max |wpack|, y_amax (optional,
    zeroed) receives max |Y|."""
    Y = _new((n, H, W, Cout), X)
    ws = _ws(X)
    tb, rb = _bias_target(out_bias)
    call("mvk_conv3x3_s", ptr(X), ptr(wpack), ptr(bias), ptr(Y), n, H, W, Cin, Cout, act, ptr(y_act_src), y_src_act, ptr(res),
         float(res_alpha), ptr(tb), x_act, float(pre_scale), ptr(x_amax), ptr(w_amax), ptr(y_amax), ptr(ws), ws.numel(),
         stream_ptr())
    return Y if out_bias is None else (Y, rb)


def conv3x3_s2(X, wpack, bias, n, H, W, Cin, Cout, x_amax, w_amax, y_amax, act, res, res_alpha):
    """conv3x3_s in its residual form with a second store -> (res + res_alpha * a, a), a = act(conv + bias)."""
    Y, A = _new((n, H, W, Cout), X), _new((n, H, W, Cout), X)
    ws = _ws(X)
    call("mvk_conv3x3_s2", ptr(X), ptr(wpack), ptr(bias), ptr(Y), ptr(A), n, H, W, Cin, Cout, act, ptr(res), float(res_alpha),
         ptr(x_amax), ptr(w_amax), ptr(y_amax), ptr(ws), ws.numel(), stream_ptr())
    return Y, A


def conv3x3_s_split(X, wpack, bias, n, H, W, Cin, Cout, x_amax, w_amax, y_amax=None, act=NONE, y_act_src=None, y_src_act=NONE,
                    out_bias=None, x_act=NONE, pre_scale=1.0):
    """conv3x3_s for a layer with 256 input channels: two launches over 128-channel slices (mvk_conv3x3_s_part) — the first leaves
    its partial sum, the second adds it in front of its bias / activation / mask and publishes max |Y|.  out_bias: the column
    sums of Y (the bias gradient of the layer below) come from a pass of their own."""
    half = Cin // 2
    part, Y = _new((n, H, W, Cout), X), _new((n, H, W, Cout), X)
    call("mvk_conv3x3_s_part", ptr(X), Cin, 0, ptr(wpack), None, ptr(part), n, H, W, half, Cout, NONE, None, NONE, None, 0, x_act,
         float(pre_scale), ptr(x_amax), ptr(w_amax), None, stream_ptr())
    call("mvk_conv3x3_s_part", ptr(X), Cin, half, ptr(wpack), ptr(bias), ptr(Y), n, H, W, half, Cout, act, ptr(y_act_src), y_src_act,
         ptr(part), 1, x_act, float(pre_scale), ptr(x_amax), ptr(w_amax), ptr(y_amax), stream_ptr())
    return Y if out_bias is None else (Y, colsum(Y.view(-1, Cout), out_bias))


def conv3x3_wgrad_scaled_ok(n, H, W, Cin, Cout):
    """True when mvk_conv3x3_wgrad_s takes this problem."""
    return C3_F16 and bool(_lib.load().mvk_conv3x3_wgrad_scaled_ok(n, H, W, Cin, Cout))


def conv3x3_wgrad_s(X, dY, wparam, bparam, n, H, W, Cin, Cout, x_amax, dy_amax, x_act=NONE, dy_scale=1.0):
    """conv3x3_wgrad_f on scaled fp16 pairs; x_amax / dy_amax: device scalars bounding max |X| / max |dY|."""
    dw, rw = _grad_target(wparam)
    tb, rb = _bias_target(bparam)
    ws = _ws(X)
    call("mvk_conv3x3_wgrad_s", ptr(X), ptr(dY), ptr(dw), ptr(tb), n, H, W, Cin, Cout, x_act, float(dy_scale), ptr(x_amax),
         ptr(dy_amax), ptr(ws), ws.numel(), stream_ptr())
    return rw, rb


def conv3x3_wgrad_f(X, dY, wparam, bparam, n, H, W, Cin, Cout, x_act=NONE, dy_scale=1.0):
    """dW += dy_scale * sum x_act(X) (x) dY and (bparam given) db += dy_scale * sum dY, one launch.  -> (dW ref, db ref)"""
    dw, rw = _grad_target(wparam)
    tb, rb = _bias_target(bparam)
    ws = _ws(X)
    call("mvk_conv3x3_wgrad_f", ptr(X), ptr(dY), ptr(dw), ptr(tb), n, H, W, Cin, Cout, x_act, float(dy_scale), ptr(ws),
         ws.numel(), stream_ptr())
    return rw, rb


def avgpool(x, n, H, W, C):
    y = _new((n, (H + 1) // 2, (W + 1) // 2, C), x)
    call("mvk_avgpool3s2_fwd", ptr(x), ptr(y), n, H, W, C, stream_ptr())
    return y


def avgpool_bwd(dy, n, H, W, C):
    dx = _new((n, H, W, C), dy)
    call("mvk_avgpool3s2_bwd", ptr(dy), ptr(dx), n, H, W, C, stream_ptr())
    return dx


def upsample2(x, n, H, W, C):
    y = _new((n, 2 * H, 2 * W, C), x)
    call("mvk_upsample2_fwd", ptr(x), ptr(y), n, H, W, C, stream_ptr())
    return y


def upsample2_bwd(dy, n, H, W, C):
    dx = _new((n, H, W, C), dy)
    call("mvk_upsample2_bwd", ptr(dy), ptr(dx), n, H, W, C, stream_ptr())
    return dx


# ---- noise with the generator state in device memory (mvk_device_rng) -------------------------------------------------
# MVK_DEVICE_RNG=0: draw from torch's generator instead (two extra host-issued fill launches per hipGraph replay).
DEVICE_RNG = _lib.tune("MVK_DEVICE_RNG", "1") != "0"
_RNG = {}  # device -> [state tensor (3 x int64: seed, offset, ticket), torch seed it was built from, breadcrumb offset]


def _rng_state(device):
    """The device-resident generator state, tied to torch's CUDA generator of that device: re-seeded from its seed whenever
    torch was re-seeded since the last draw.  A re-seed is seen as a changed seed or an offset that went backwards — every
    draw leaves a breadcrumb by advancing torch's (host-side) offset, so `torch.manual_seed(s)` twice with the same s and
    nothing drawn from torch in between is still noticed.  Not consulted while a stream is capturing (the launches inside
    a graph advance the state they find).

    The state tensor is allocated ONCE per device and re-seeded IN PLACE (a stream-ordered copy): a captured GraphedStep has
    its raw pointer baked into its mvk_device_rng launches, so the tensor must never be replaced — and a re-seed then also
    reaches every existing graph.  Draws must be stream-ordered with respect to each other (the {offset, ticket} protocol of
    device_rng_kernel serialises nothing): every caller in the package draws on the step's main stream or on a branch
    stream between that stream's fork and join."""
    ent = _RNG.get(device)
    if ent is not None and torch.cuda.is_current_stream_capturing():
        return ent[0]
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    seed, off = gen.initial_seed(), gen.get_offset()
    if ent is None or ent[1] != seed or off < ent[2]:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.MvkError("the device generator state must exist before a graph capture starts (run one eager step)")
        fresh = torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, off // 4, 0], dtype=torch.int64)
        if ent is None:
            ent = [torch.empty(3, dtype=torch.int64, device=device), seed, off]
            _RNG[device] = ent
        ent[0].copy_(fresh)  # pageable host source: the copy has read `fresh` when it returns; ordered on the current stream
        ent[1], ent[2] = seed, off
    ent[2] = off + 4
    gen.set_offset(ent[2])
    return ent[0]


def device_randn(shape, device, uniform=False, lo=0.0, hi=1.0):
    """N(0, 1) (or U[lo, hi)) noise of `shape` from the device-resident generator."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    call("mvk_device_rng", ptr(out), out.numel(), ptr(_rng_state(device)), 1 if uniform else 0, float(lo), float(hi), stream_ptr())
    return out


# Test hook (tests/relu_sites.py): a list that receives (node, first weight, activation tensors) from the forward of every
# network node, so a parity test can read which side of zero the HIP path put each ReLU / LeakyReLU unit on.  None = off (the default).
TAPS = None


def _tap(node, w, *acts):
    if TAPS is not None:
        TAPS.append((node, w, acts))


class TransposeActFn(Function):
    """y[b, c, r] = act(x[b, r, c]): NHWC <-> NCHW-flattened in ONE pass with the activation in front of the flatten
    (mvk_transpose_act); backward = the transposed gradient times act'(y)."""

    @staticmethod
    def forward(ctx, x, act):
        x = _c(x)
        b, r, c = x.shape
        y = _new((b, c, r), x)
        call("mvk_transpose_act", ptr(x), ptr(y), b, r, c, act, None, NONE, stream_ptr())
        ctx.act = act
        if act != NONE:
            ctx.save_for_backward(y)
            _tap("transpose_act", None, y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = _c(dy)
        b, c, r = dy.shape
        dx = _new((b, r, c), dy)
        y = ctx.saved_tensors[0] if ctx.act != NONE else None
        call("mvk_transpose_act", ptr(dy), ptr(dx), b, c, r, NONE, ptr(y), ctx.act, stream_ptr())
        return dx, None


def nhwc_to_flat_nchw(h, act=NONE):
    """[n, H, W, C] (NHWC) -> [n, C * H * W] in the reference's NCHW flatten order, with `act` applied on the way."""
    n, H, W, C = h.shape
    return TransposeActFn.apply(h.reshape(n, H * W, C), act).reshape(n, C * H * W)


def flat_nchw_to_nhwc(flat, C, H, W):
    """[n, C * H * W] (NCHW order) -> [n, H, W, C]."""
    n = flat.shape[0]
    return TransposeActFn.apply(flat.reshape(n, C, H * W), NONE).reshape(n, H, W, C)


def axpby(x, a, y, b, act=NONE, out=None):
    ref = x if x is not None else y
    out = torch.empty_like(ref) if out is None else out
    call("mvk_axpby", ptr(x), float(a), ptr(y), float(b), ref.numel(), act, ptr(out), stream_ptr())
    return out


def conv_wgrad(U, V, wparam, n, h, w, Cu, Cv, u_nchw=False, u_act_src=None, u_act=NONE, amax=None):
    """amax = (u_amax, v_amax): the scaled-fp16 launch (mvk_conv4s2_wgrad_s) where mvk_conv4s2_wgrad_scaled_ok says so."""
    dw, rw = _grad_target(wparam)
    ws = _ws(V)
    if (amax is not None and amax[0] is not None and amax[1] is not None and not u_nchw and u_act_src is None
            and _lib.load().mvk_conv4s2_wgrad_scaled_ok(n, h, w, Cu, Cv)):
        call("mvk_conv4s2_wgrad_s", ptr(U), ptr(V), ptr(dw), n, h, w, Cu, Cv, ptr(amax[0]), ptr(amax[1]), ptr(ws), ws.numel(),
             stream_ptr())
        return rw
    call("mvk_conv4s2_wgrad", ptr(U), ptr(V), ptr(dw), n, h, w, Cu, Cv, int(u_nchw), ptr(u_act_src), u_act,
         ptr(ws), ws.numel(), stream_ptr())
    return rw


# =====================================================================================================
# Encoder_VAE_MLP / Decoder_AE_MLP  (models/nn/default_architectures.py:21-72, 225-258)
# =====================================================================================================
class MLPEncoderFn(Function):
    """x -> [Linear+ReLU]*n -> (embedding, log_covariance).  params = (W0,b0, W1,b1, ..., We,be, Wl,bl)."""

    @staticmethod
    def forward(ctx, x, *params):
        n_layers = (len(params) - 4) // 2
        K0 = params[0].shape[1]
        x2 = _c(x.reshape(-1, K0))
        acts = [x2]
        h = x2
        for i in range(n_layers):
            h = linear_fwd(h, params[2 * i], params[2 * i + 1], RELU)
            acts.append(h)
        we, be, wl, bl = params[-4:]
        heads = heads_fwd(h, we, be, wl, bl, we.shape[0], 1, we.shape[1]) if we.shape == wl.shape else None
        if heads is not None:
            mu, lv = heads
        else:
            mu = linear_fwd(h, we, be, NONE)
            lv = linear_fwd(h, wl, bl, NONE)
        ctx.save_for_backward(*acts, *params)
        _tap("mlp_encoder", params[0], *acts[1:])
        ctx.n_layers = n_layers
        ctx.x_shape = x.shape
        return mu, lv

    @staticmethod
    @once_differentiable
    def backward(ctx, dmu, dlv):
        n = ctx.n_layers
        saved = ctx.saved_tensors
        acts, params = saved[: n + 1], saved[n + 1 :]
        we, be, wl, bl = params[-4:]
        dmu, dlv = _c(dmu), _c(dlv)
        h = acts[-1]
        grads = [None] * len(params)
        prev_src, prev_act = (h, RELU) if n > 0 else (None, NONE)
        need_dx = ctx.needs_input_grad[0]
        dh = None
        bias_done = False  # bias gradient of layer i already produced by the backward-data launch of layer i+1
        # both heads in one launch: weight / bias gradients, the gradient w.r.t. the last hidden layer's PRE-activation
        # (ReLU' applied) and, with hidden layers, that layer's bias gradient
        fused = heads_bwd(h, prev_act, [dmu, dlv], [we, wl], [be, bl], 1, h.shape[1], want_dx=n > 0 or need_dx,
                          prev_bias=params[2 * n - 1] if n > 0 else None) if we.shape == wl.shape and heads_bwd_mlp(h.device) else None
        if fused is not None:
            dh, (grads[-4], grads[-2]), (grads[-3], grads[-1]), gprev = fused
            if n > 0:
                grads[2 * n - 1], bias_done = gprev, True
        else:
            grads[-4], grads[-3] = linear_bwd_weight(dmu, h, we, be)
            grads[-2], grads[-1] = linear_bwd_weight(dlv, h, wl, bl)
            if n > 0 or need_dx:
                dh = linear_bwd_data(dmu, we, prev_out=prev_src, prev_act=prev_act)
                linear_bwd_data(dlv, wl, prev_out=prev_src, prev_act=prev_act, out=dh, accumulate=True)
        for i in range(n - 1, -1, -1):
            w, inp = params[2 * i], acts[i]
            gb = grads[2 * i + 1]
            grads[2 * i], gb2 = linear_bwd_weight(dh, inp, w, None if bias_done else params[2 * i + 1])
            grads[2 * i + 1] = gb if bias_done else gb2
            bias_done = False
            if i > 0:
                dh, grads[2 * i - 1] = linear_bwd_data(dh, w, prev_out=acts[i], prev_act=RELU, prev_bias=params[2 * i - 1])
                bias_done = True
            elif need_dx:
                dh = linear_bwd_data(dh, w)
        dx = dh.reshape(ctx.x_shape) if need_dx else None
        return (dx, *grads)


class MLPHeadsFn(Function):
    """x -> [Linear+ReLU]*n -> n_heads linear heads (Encoder_VAE_MLP_Style: embedding, log_var, style_embedding,
    style_log_var; default_architectures.py:75-141).  params = (W0,b0, ..., Wh1,bh1, ..., Whn,bhn)."""

    @staticmethod
    def forward(ctx, x, n_heads, *params):
        n_layers = (len(params) - 2 * n_heads) // 2
        K0 = params[0].shape[1]
        x2 = _c(x.reshape(-1, K0))
        acts = [x2]
        h = x2
        for i in range(n_layers):
            h = linear_fwd(h, params[2 * i], params[2 * i + 1], RELU)
            acts.append(h)
        outs = tuple(linear_fwd(h, params[2 * (n_layers + j)], params[2 * (n_layers + j) + 1], NONE)
                     for j in range(n_heads))
        ctx.save_for_backward(*acts, *params)
        _tap("mlp_heads", params[0], *acts[1:])
        ctx.n_layers, ctx.n_heads = n_layers, n_heads
        ctx.x_shape = x.shape
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, *douts):
        n, nh = ctx.n_layers, ctx.n_heads
        saved = ctx.saved_tensors
        acts, params = saved[: n + 1], saved[n + 1 :]
        h = acts[-1]
        grads = [None] * len(params)
        prev_src, prev_act = (h, RELU) if n > 0 else (None, NONE)
        need_dx = ctx.needs_input_grad[0]
        dh = None
        fused = None
        if heads_bwd_mlp(h.device) and nh <= 2 and (n > 0 or need_dx):
            # every head in one launch; with hidden layers it also emits the last hidden layer's bias gradient
            hw = [params[2 * (n + j)] for j in range(nh)]
            fused = heads_bwd(h, prev_act, [_c(d) for d in douts[:nh]], hw, [params[2 * (n + j) + 1] for j in range(nh)],
                              1, h.shape[1], prev_bias=params[2 * n - 1] if n > 0 else None)
        if fused is not None:
            dh, gw, gb, gprev = fused
            for j in range(nh):
                grads[2 * (n + j)], grads[2 * (n + j) + 1] = gw[j], gb[j]
        for j in range(nh if fused is None else 0):
            w, b = params[2 * (n + j)], params[2 * (n + j) + 1]
            dy = _c(douts[j])
            grads[2 * (n + j)], grads[2 * (n + j) + 1] = linear_bwd_weight(dy, h, w, b)
            if n > 0 or need_dx:  # pre-activation gradient of the last hidden layer: all heads, ReLU' in the epilogue
                if dh is None:
                    dh = linear_bwd_data(dy, w, prev_out=prev_src, prev_act=prev_act)
                else:
                    linear_bwd_data(dy, w, prev_out=prev_src, prev_act=prev_act, out=dh, accumulate=True)
        for i in range(n - 1, -1, -1):
            w, inp = params[2 * i], acts[i]
            if fused is not None and i == n - 1:  # its bias gradient came with the heads' launch
                grads[2 * i], _ = linear_bwd_weight(dh, inp, w, None)
                grads[2 * i + 1] = gprev
            else:
                grads[2 * i], grads[2 * i + 1] = linear_bwd_weight(dh, inp, w, params[2 * i + 1])
            if i > 0:
                dh = linear_bwd_data(dh, w, prev_out=acts[i], prev_act=RELU)
            elif need_dx:
                dh = linear_bwd_data(dh, w)
        dx = dh.reshape(ctx.x_shape) if need_dx else None
        return (dx, None, *grads)


# The MLP decoder at the decoder batch on pre-split fp16 pair planes (csrc/dense16.hip): MVK_DENSE16=0 keeps the tiled engine
DENSE16 = _lib.tune("MVK_DENSE16", "1") != "0"
DENSE16_MIN_ROWS = 1024  # below, the tiled engine's latency-sized tiles win (tests set 1 to meet the small goldens)


def mlp_fused_tail_ok(n, L, H, D):
    """MLPDecoderFn's fused tail (mvk_dense16_*) takes z [n, L] -> Linear(L, H) + ReLU -> Linear(H, D) + Sigmoid."""
    return (DENSE16 and n >= DENSE16_MIN_ROWS and 4 <= L <= 32 and L % 4 == 0 and 16 <= H <= 1024 and H % 16 == 0
            and 256 % (H // 4) == 0 and D % 8 == 0 and bool(_lib.load().mvk_dense16_ok(n, D, H)))


def _planes(rows, cols, like):
    """An fp16 pair-planes buffer [2][rows][cols] (hi, lo): the bytes of the fp32 tensor it replaces."""
    t = torch.empty((2, rows, cols), dtype=torch.float16, device=like.device)
    return t, t[0], t[1]


def dense16_pack(w, raw=False, defer=None):
    """Planes of a Linear weight [N][K] in both orientations + the per-row inverse scales; once per forward pass inside a
    pack_scope (the cache dies with the scope: the weights change at the next optimizer step).  raw: launch here, no cache."""
    sc = _PACK_SCOPE
    key = ("dense16", w.data_ptr(), tuple(w.shape))
    rot = rotation(w.device)
    if not raw and rot is not None and key in rot.cache:  # packed by the head branch of a rotated step, behind w's update
        rot.wait()
        return rot.cache[key]
    if not raw and sc is not None and key in sc["cache"]:
        return sc["cache"][key]
    N, K = w.shape
    nk, kn = _planes(N, K, w), _planes(K, N, w)
    nk_inv, kn_inv = _new((N,), w), _new((K,), w)

    def go():
        call("mvk_dense16_pack", ptr(w), N, K, ptr(nk[1]), ptr(nk[2]), ptr(nk_inv), ptr(kn[1]), ptr(kn[2]), ptr(kn_inv), stream_ptr())

    if defer is not None:
        defer.append(go)  # (the caller launches it: MLPDecoderFn with MVK_FWD_DEFER=4)
    else:
        go()
    out = (nk, nk_inv, kn, kn_inv)
    if sc is not None and not raw:
        sc["cache"][key] = out
    return out


def dense16_xamax(x, defer=None):
    """Device scalar >= max |x| of a target batch, once per forward pass inside a pack_scope."""
    sc = _PACK_SCOPE
    key = ("dense16_xamax", x.data_ptr(), tuple(x.shape))
    if sc is not None and key in sc["cache"]:
        return sc["cache"][key]
    slot = _amax_slots(x, 1)
    if defer is not None:
        defer.append(lambda: call("mvk_amax", ptr(x), x.numel(), ptr(slot), stream_ptr()))
    else:
        call("mvk_amax", ptr(x), x.numel(), ptr(slot), stream_ptr())
    if sc is not None:
        sc["cache"][key] = slot
    return slot


_CONST_GRADS = {}  # data_ptr -> (weakref of the tensor, numel, value): gradient buffers known to hold ONE constant


def register_const_grad(t, value):
    """ReconLossFn.backward: `t` was filled with `value` by the forward launch (the unit-seed path)."""
    import weakref

    _CONST_GRADS[t.data_ptr()] = (weakref.ref(t), t.numel(), float(value))


def const_grad(t):
    """The constant every element of the gradient tensor `t` holds, when a producer registered it (None otherwise: read it on
    the device).  The registration dies with the tensor: a recycled address never matches."""
    e = _CONST_GRADS.get(t.data_ptr())
    if e is None:
        return None
    ref, numel, value = e
    src = ref()
    if src is None or numel != t.numel() or src.data_ptr() != t.data_ptr():
        _CONST_GRADS.pop(t.data_ptr(), None)
        return None
    return value


def _check_nll_weight(nll_x, nll_weight):
    """The fused tails store nll_weight * d NLL / d pre-activation and divide by it on their general backward path: a zero or
    non-finite weight has no such form (the decoders' `reconstruction_nll` return None for it and the caller takes the generic
    likelihood kernel)."""
    if nll_x is not None and not (math.isfinite(float(nll_weight)) and float(nll_weight) != 0.0):
        raise _lib.MvkError(f"fused decoder tail: nll_weight must be finite and non-zero, got {nll_weight}")


class MLPDecoderFn(Function):
    """z[...,L] -> Linear+ReLU -> Linear+Sigmoid -> reshape(*z.shape[:-1], *input_dim).

    nll_x [B, D] given: the FUSED TAIL on fp16 pair planes (csrc/dense16.hip; the caller checks `mlp_fused_tail_ok`) — the
    output layer scores its rows against nll_x[row % B] by a Normal(nll_scale) likelihood in its epilogue and the node returns
    PARTIAL NLL row sums [P, *z.shape[:-1]] (their sum over P is -log p(x | decoder(z)) per row) instead of the reconstruction;
    what is kept for the backward pass is nll_weight * d NLL / d pre-activation as planes.  nll_weight = the weight the rows
    are expected to enter the loss with (d loss / d rows): when the backward pass receives exactly that constant (const_grad)
    nothing is rescaled; any other upstream gradient takes the general path (planes -> fp32 times the row factor)."""

    @staticmethod
    def forward(ctx, z, w0, b0, w1, b1, input_dim, nll_x=None, nll_scale=1.0, nll_weight=1.0):
        _check_nll_weight(nll_x, nll_weight)
        L = w0.shape[1]
        z2 = _c(z.reshape(-1, L))
        ctx.fused = nll_x is not None
        ctx.z_shape = z.shape
        if ctx.fused:
            n, H, D = z2.shape[0], w0.shape[0], w1.shape[0]
            # rotated step (Rotation): the weight gradients of w1 / b1 / w0 run at the head of the NEXT step; what they read (z,
            # the planes of h and of the stored gradient, their bounds, the column-sum partials; dh in backward) is persistent
            rot = rotation(z2.device)
            if rot is not None:
                rot.wait()  # w0 is read as it is, w1 through planes packed behind its update
            if not (ROT_MLP and rot is not None and LATE_LEAVES and z2.device in _DEFER_ACTIVE
                    and all(_is_direct(t) for t in (w0, w1, b1))):
                rot = None
            ctx.rot, ctx.rot_mode = rot, ROT_MLP
            if rot is not None and ROT_MLP == 2:
                # mode 2 rotates the first layer's gradient only: z at a fixed address (copied HERE, on this branch's stream: the
                # late-leaf stream must never wait for a branch stream — hip::Stream::EndCapture recursion, see deferred_reductions)
                z2 = rot.buf(("mlp_dec", w0.data_ptr(), z2.shape[0], "z"), z2.shape).copy_(z2)
            if ROT_MLP != 1:
                rot = None
            rkey = ("mlp_dec", w0.data_ptr(), n)
            # MVK_FWD_DEFER=4: this branch's z-independent preparation joins the postponed launches (flushed behind the large
            # decoder's FIRST layer, the latency-sized launch on the step's chain these 30 us of launches otherwise run beside)
            prep = [] if (FWD_DEFER_AT == 4 and z2.device in _FWD_DEFER_OPEN and TAPS is None) else None
            nk, nk_inv, kn, kn_inv = dense16_pack(w1, defer=prep)  # both usually done already: Decoder_AE_MLP.early_work
            xam = dense16_xamax(nll_x, defer=prep)
            zam = _amax_slots(z2, 1)
            lib = _lib.load()
            P, CR = lib.mvk_dense16_fwd_nll_rows(D), lib.mvk_dense16_colsum_rows(n)
            if rot is None:
                hp = _planes(n, H, z2)
                gp = _planes(n, D, z2)
                bounds = _new((2,), z2)  # [bound of h, bound of G]
                cs = _new((CR, D), z2)
            else:
                z2 = rot.buf(rkey + ("z",), z2.shape).copy_(z2)
                hp = rot.buf(rkey + ("hp",), (2, n, H), torch.float16)
                hp = (hp, hp[0], hp[1])
                gp = rot.buf(rkey + ("gp",), (2, n, D), torch.float16)
                gp = (gp, gp[0], gp[1])
                bounds = rot.buf(rkey + ("bounds",), (2,))
                cs = rot.buf(rkey + ("cs",), (CR, D))
            rows = _new((P, n), z2)

            def launch():  # the three launches of the forward pass (possibly postponed: defer_forward)
                for f in prep or ():
                    f()
                call("mvk_amax", ptr(z2), z2.numel(), ptr(zam), stream_ptr())
                call("mvk_dense16_first", ptr(z2), ptr(w0), ptr(b0), ptr(zam), ptr(hp[1]), ptr(hp[2]), ptr(bounds[0:1]), n, H, L,
                     RELU, stream_ptr())
                call("mvk_dense16_fwd_nll", ptr(hp[1]), ptr(hp[2]), ptr(bounds[0:1]), ptr(nk[1]), ptr(nk[2]), ptr(nk_inv), ptr(b1),
                     ptr(nll_x), nll_x.shape[0], ptr(xam), float(nll_scale), float(nll_weight), ptr(gp[1]), ptr(gp[2]),
                     ptr(bounds[1:2]), ptr(rows), ptr(cs), n, D, H, stream_ptr())

            if not defer_forward(z2.device, launch):
                launch()
            ctx.save_for_backward(z2, hp[0], gp[0], bounds, kn[0], kn_inv, cs, w0, b0, w1, b1)
            ctx.nll_weight = float(nll_weight)
            if TAPS is not None:  # tests: the hidden activation as an fp32 tensor (the sign of h is the sign of its hi plane)
                h = _new((n, H), z2)
                call("mvk_dense16_unsplit", ptr(hp[1]), ptr(hp[2]), ptr(bounds[0:1]), None, 1.0, 0, n, H, ptr(h), stream_ptr())
                _tap("mlp_decoder", w0, h)
            return rows.view(P, *z.shape[:-1])
        h = linear_fwd(z2, w0, b0, RELU)
        out = linear_fwd(h, w1, b1, SIGMOID)
        ctx.save_for_backward(z2, h, out, w0, b0, w1, b1)
        _tap("mlp_decoder", w0, h)
        return out.view(*z.shape[:-1], *input_dim)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        if ctx.fused:
            return MLPDecoderFn._backward_fused(ctx, dout)
        z2, h, out, w0, b0, w1, b1 = ctx.saved_tensors
        dout = _c(dout).view(out.shape)
        # pre-activation gradient of the output layer + its bias gradient in one pass; the two GEMMs then read it
        # plainly (instead of re-deriving it from dout and out on every operand load, and a third time for the bias)
        dpre = torch.empty_like(dout)
        tb1, db1 = _grad_target(b1)
        ws = _ws(dout)
        call("mvk_act_bwd_colsum", ptr(dout), ptr(out), SIGMOID, dout.shape[0], dout.shape[1], ptr(dpre), ptr(tb1),
             ptr(ws), ws.numel(), stream_ptr())
        dw1, _ = linear_bwd_weight(dpre, h, w1, None)
        dh, db0 = linear_bwd_data(dpre, w1, prev_out=h, prev_act=RELU, prev_bias=b0)
        dw0, _ = linear_bwd_weight(dh, z2, w0, None)
        dz = None
        if ctx.needs_input_grad[0]:
            dz = linear_bwd_data(dh, w0).view(ctx.z_shape)
        return dz, dw0, db0, dw1, db1, None, None, None, None

    @staticmethod
    def _backward_fused(ctx, drows):
        z2, hp, gp, bounds, kn, kn_inv, cs, w0, b0, w1, b1 = ctx.saved_tensors
        wait_tail_bwd(z2.device)
        n, H, D = z2.shape[0], w0.shape[0], w1.shape[0]
        hb, gb = bounds[0:1], bounds[1:2]
        c = const_grad(drows)
        ws = _ws(z2)
        if c is not None and c == ctx.nll_weight:
            # the planes already hold d loss / d pre-activation: two GEMMs on planes, bias gradients out of their launches
            tw1, dw1 = _grad_target(w1)
            tb1, db1 = _grad_target(b1)
            tb0, db0 = _grad_target(b0)
            rot = ctx.rot if (ctx.rot is not None and rotation(z2.device) is ctx.rot and dw1 is None and db1 is None) else None
            rot0 = rot if (rot is not None and ctx.rot_mode == 2) else None
            dh = _new((n, H), z2) if rot is None else rot.buf(("mlp_dec", w0.data_ptr(), n, "dh"), (n, H))
            if rot0 is not None:
                rot = None
            call("mvk_dense16_bwd_data", ptr(gp[0]), ptr(gp[1]), ptr(gb), ptr(kn[0]), ptr(kn[1]), ptr(kn_inv), ptr(hp[0]), ptr(dh),
                 ptr(tb0), ptr(ws), ws.numel(), n, H, D, stream_ptr())
            # the gradient of z first (the posterior's backward waits for it), the weight gradients — leaves — behind it
            dz = None
            if ctx.needs_input_grad[0]:
                dz = narrow_linear(dh, w0, n, w0.shape[1], H, w0.shape[1], 1)  # W(k = h, l) = w0[h][l]
                dz = (dz if dz is not None else linear_bwd_data(dh, w0)).view(ctx.z_shape)

            def wgrad1():
                wsl = _ws(z2)
                call("mvk_dense16_wgrad", ptr(gp[0]), ptr(gp[1]), ptr(gb), ptr(hp[0]), ptr(hp[1]), ptr(hb), ptr(cs), cs.shape[0],
                     ptr(tw1), ptr(tb1), ptr(wsl), wsl.numel(), n, D, H, stream_ptr())

            if rot is not None:  # rotated step: both weight gradients are the head of the next step (Rotation.begin_step)
                def leaves():
                    wgrad1()
                    linear_bwd_weight(dh, z2, w0, None)

                def prepare(r):  # behind the update of w1: its planes for the next forward pass
                    r.cache[("dense16", w1.data_ptr(), tuple(w1.shape))] = dense16_pack(w1, raw=True)

                rot.push(leaves, (w1, b1, w0), prepare)
                return dz, None, db0, None, None, None, None, None, None

            # a leaf: postponed into the tail of the step when its targets are views of the flat gradient buffer
            if dw1 is not None or db1 is not None or not run_last(z2.device, wgrad1, gp, hp, cs, bounds, params=(w1, b1)):
                wgrad1()
            dw0 = None
            if rot0 is not None:  # mode 2: the first layer's weight gradient is the head of the next step (z2 is the fixed copy)
                rot0.push(lambda: linear_bwd_weight(dh, z2, w0, None), (w0,), None)
            elif not _is_direct(w0) or not run_last(z2.device, lambda: linear_bwd_weight(dh, z2, w0, None), dh, z2, force=LATE_DW0, params=(w0,)):
                dw0, _ = linear_bwd_weight(dh, z2, w0, None)
            return dz, dw0, db0, dw1, db1, None, None, None, None
        else:  # general upstream gradient: d pre = G * drows[column tile, row] / nll_weight as an fp32 tensor, then the tiled engine
            wait_loss(z2.device)
            drows = _c(drows).reshape(-1, n)
            dpre = _new((n, D), z2)
            h = _new((n, H), z2)
            call("mvk_dense16_unsplit", ptr(gp[0]), ptr(gp[1]), ptr(gb), ptr(drows), 1.0 / ctx.nll_weight, 128, n, D, ptr(dpre),
                 stream_ptr())
            call("mvk_dense16_unsplit", ptr(hp[0]), ptr(hp[1]), ptr(hb), None, 1.0, 0, n, H, ptr(h), stream_ptr())
            db1 = colsum(dpre, b1)
            dw1, _ = linear_bwd_weight(dpre, h, w1, None)
            dh, db0 = linear_bwd_data(dpre, w1, prev_out=h, prev_act=RELU, prev_bias=b0)
        dz = None
        if ctx.needs_input_grad[0]:
            dz = linear_bwd_data(dh, w0).view(ctx.z_shape)
        dw0 = None
        if not _is_direct(w0) or not run_last(z2.device, lambda: linear_bwd_weight(dh, z2, w0, None), dh, z2, params=(w0,)):
            dw0, _ = linear_bwd_weight(dh, z2, w0, None)
        return dz, dw0, db0, dw1, db1, None, None, None, None


# =====================================================================================================
# Encoder_VAE_SVHN / Decoder_VAE_SVHN  (models/nn/svhn.py:7-70), NHWC activations inside
# =====================================================================================================
class SVHNEncoderFn(Function):
    """x[B,C,32,32] NCHW -> 3x(Conv 4/2/1 + ReLU) -> two Conv(4,2,0) heads -> (mu, lv) [B,L]."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, w2, b2, wc1, bc1, wc2, bc2):
        x = _c(x)
        B, C0, H, W = x.shape
        chans = [C0, w0.shape[0], w1.shape[0], w2.shape[0]]
        L = wc1.shape[0]
        if _lib.load().mvk_conv4s2_small_up_supported(H // 2, W // 2, chans[0], chans[1]):
            # the image-consuming layer reads the reference weight layout: no pack in front of the encoder's first kernel
            h1 = _new((B, H // 2, W // 2, chans[1]), x)
            call("mvk_conv4s2_small_down_fwd_wref", ptr(x), ptr(w0), ptr(b0), ptr(h1), B, H // 2, W // 2, chans[0], chans[1], RELU,
                 stream_ptr())
            (wd1, wu1), (wd2, wu2), (wdc1, _), (wdc2, _) = pack_weights(
                [(w1, True, True), (w2, True, True), (wc1, True, False), (wc2, True, False)])
        else:
            (wd0, _), (wd1, wu1), (wd2, wu2), (wdc1, _), (wdc2, _) = pack_weights(
                [(w0, True, False), (w1, True, True), (w2, True, True), (wc1, True, False), (wc2, True, False)])
            h1 = conv_down(x, wd0, b0, B, H // 2, W // 2, chans[0], chans[1], RELU, u_nchw=True)
        h2 = conv_down(h1, wd1, b1, B, H // 4, W // 4, chans[1], chans[2], RELU)
        h3 = conv_down(h2, wd2, b2, B, H // 8, W // 8, chans[2], chans[3], RELU)
        if (H // 8, W // 8) != (4, 4):
            raise _lib.MvkError("Encoder_VAE_SVHN expects 32x32 inputs (4x4 feature map before the heads)")
        Kf = 16 * chans[3]
        h3f = h3.view(B, Kf)
        heads = heads_fwd(h3f, wdc1, bc1, wdc2, bc2, L, L, 1)
        if heads is not None:
            mu, lv = heads
        else:
            mu = gemm(h3f, wdc1, B, L, Kf, bias=bc1, bias_mod=L)
            lv = gemm(h3f, wdc2, B, L, Kf, bias=bc2, bias_mod=L)
        ctx.save_for_backward(x, h1, h2, h3, wu1, wu2, wdc1, wdc2, w0, b0, w1, b1, w2, b2, wc1, bc1, wc2, bc2)
        _tap("svhn_encoder", w0, h1, h2, h3)  # NHWC
        ctx.frags = (wfrag(wu1), wfrag(wu2))  # saved tensors come back without Python attributes
        ctx.dims = (B, H, W, chans, L)
        return mu, lv

    @staticmethod
    @once_differentiable
    def backward(ctx, dmu, dlv):
        x, h1, h2, h3, wu1, wu2, wdc1, wdc2, w0, b0, w1, b1, w2, b2, wc1, bc1, wc2, bc2 = ctx.saved_tensors
        B, H, W, ch, L = ctx.dims
        dmu, dlv = _c(dmu).view(B, L), _c(dlv).view(B, L)
        Kf = 16 * ch[3]
        h3f = h3.view(B, Kf)
        defer_flush_sibling(x.device, node_params=(w0, b0, w1, b1, w2, b2, wc1, bc1, wc2, bc2))
        # heads: both weight gradients (in the Conv2d layout), both bias gradients, d h3 (pre-activation: dmu Wd1^T + dlv Wd2^T
        # with ReLU'(h3)) and its channel sums (the bias gradient of the layer below) in one launch
        lf = LeafStream(x.device)  # the weight / bias gradients run beside the backward-data chain
        fused = heads_bwd(h3f, RELU, [dmu, dlv], [wdc1, wdc2], [bc1, bc2], L, 1, flat_c=ch[3], prev_bias=b2,
                          dw_params=[wc1, wc2]) if (H // 8, W // 8) == (4, 4) else None
        if fused is not None:
            dh3, (dwc1, dwc2), (dbc1, dbc2), db2 = fused
        else:
            tw1, dwc1 = _grad_target(wc1)
            tw2, dwc2 = _grad_target(wc2)
            with lf:
                ws = _ws(x)
                call("mvk_flatten_wgrad", ptr(h3f), ptr(dmu), ptr(tw1), B, ch[3], L, ptr(ws), ws.numel(), stream_ptr())
                call("mvk_flatten_wgrad", ptr(h3f), ptr(dlv), ptr(tw2), B, ch[3], L, ptr(ws), ws.numel(), stream_ptr())
                dbc1, dbc2 = colsum(dmu, bc1), colsum(dlv, bc2)
            dh3 = gemm(dmu, wdc1, B, Kf, L, tb=True, c_act_src=h3f, c_act=RELU)
            gemm(dlv, wdc2, B, Kf, L, tb=True, c_act_src=h3f, c_act=RELU, out=dh3, accumulate=True)
        dh3 = dh3.view(B, H // 8, W // 8, ch[3])
        # The two inner layers' weight gradients as ONE launch behind both backward-data launches (mvk_conv4s2_wgrad_pair): at the
        # training batch each is a split-K GEMM of 256 workgroups and ~35 us, and the four launches were one dependent chain
        # (the step's last one).  Only where both targets are views of the flat gradient buffer inside deferred_reductions (the
        # pair's slabs live in the arena) and the leaf stream is off.
        pair = (WGRAD_PAIR and not lf.on and x.device in _DEFER_ACTIVE and _is_direct(w1) and _is_direct(w2)
                and x.device.type == "cuda")
        if not pair:
            with lf:
                dw2 = conv_wgrad(h2, dh3, w2, B, H // 8, W // 8, ch[2], ch[3])
        if fused is None:
            with lf:
                db2 = colsum(dh3.view(-1, ch[3]), b2)
        # each backward-data launch also emits the bias gradient of the layer it lands in (column sums of its output)
        dh2, db1 = conv_up(dh3, wu2, None, B, H // 8, W // 8, ch[2], ch[3], u_act_src=h2, u_act=RELU, out_bias=b1,
                           frag=ctx.frags[1], tiled=bool(ENC_BWD_TILED & 1) and x.device in _BIG_LATE)
        if not pair:
            with lf:
                dw1 = conv_wgrad(h1, dh2, w1, B, H // 4, W // 4, ch[1], ch[2])
        dh1, db0 = conv_up(dh2, wu1, None, B, H // 4, W // 4, ch[1], ch[2], u_act_src=h1, u_act=RELU, out_bias=b0,
                           frag=ctx.frags[0], tiled=bool(ENC_BWD_TILED & 2) and x.device in _BIG_LATE)
        late_gate(x.device)  # the decoder's gated late leaves (MVK_LATE_GATE) may start: no 512-register launch of this chain is left
        if pair:
            ws = _ws(x)
            call("mvk_conv4s2_wgrad_pair", ptr(h2), ptr(dh3), ptr(_grad_target(w2)[0]), H // 8, W // 8, ch[2], ch[3],
                 ptr(h1), ptr(dh2), ptr(_grad_target(w1)[0]), H // 4, W // 4, ch[1], ch[2], B, ptr(ws), ws.numel(), stream_ptr())
            dw2 = dw1 = None
        with lf:
            dw0 = conv_wgrad(x, dh1, w0, B, H // 2, W // 2, ch[0], ch[1], u_nchw=True)
        lf.join()
        dx = None
        if ctx.needs_input_grad[0]:
            raise _lib.MvkError("gradient w.r.t. the encoder input image is not implemented")
        return dx, dw0, db0, dw1, db1, dw2, db2, dwc1, dbc1, dwc2, dbc2


# MVK_TAIL_F16=0: the fused SVHN tail on bf16 pieces (small_up_fwd_bf_kernel) also where the scaled-fp16 chain runs
TAIL_F16 = _lib.tune("MVK_TAIL_F16", "1") != "0"
# MVK_C3_SPLIT256=0: 3x3 layers with 256 input channels stay on the tiled engine (A/B of mvk_conv3x3_s_part)
C3_SPLIT256 = _lib.tune("MVK_C3_SPLIT256", "1") != "0"
# MVK_C3_DUAL=0: a post-activation ResNet block forms its sum in an elementwise pass behind conv2 (A/B of mvk_conv3x3_s2)
C3_DUAL = _lib.tune("MVK_C3_DUAL", "1") != "0"
# MVK_TAIL_BWD_F16=0: the image layer's backward stays on bf16 pieces (small_up_bwd_bf_kernel) where its forward runs the scaled form
TAIL_BWD_F16 = _lib.tune("MVK_TAIL_BWD_F16", "1") != "0"


class SVHNDecoderFn(Function):
    """z[...,L] -> ConvT(4,1,0)+ReLU -> 2x ConvT(4,2,1)+ReLU -> ConvT(4,2,1)+Sigmoid -> [...,C,32,32] NCHW."""

    @staticmethod
    def forward(ctx, z, w0, b0, w1, b1, w2, b2, w3, b3, nll_x=None, nll_scale=1.0, nll_weight=1.0):
        """nll_x [B, C, 32, 32] given: the FUSED TAIL — the last layer scores its image against nll_x[row % B] by a
        Normal(nll_scale) likelihood in its epilogue (mvk_conv4s2_small_up_fwd_nll) and the node returns the NLL row sums
        [*z.shape[:-1]] instead of the images; the buffer that would hold the images holds d rows / d pre-activation for the
        backward pass.  The caller checks `svhn_fused_tail_ok` first."""
        _check_nll_weight(nll_x, nll_weight)
        L = w0.shape[0]
        z2 = _c(z.reshape(-1, L))
        n = z2.shape[0]
        C1, C2, C3, C4 = w0.shape[1], w1.shape[1], w2.shape[1], w3.shape[1]
        # [L][C1] unflatten pack; [Cv=C1][Cu=C2]; [Cv=C2][Cu=C3]; [Cv=C3][Cu=C4] — one launch
        wp0, (wd1, wu1), (wd2, wu2), (wd3, _) = pack_weights(
            [(w0, "unflatten"), (w1, True, True), (w2, True, True), (w3, True, False)])
        # scaled-fp16 form of the two 4x4/stride-2 layers (csrc/imgconv.hip NP = 2): every producer on the chain publishes the
        # maximum of what it writes, the consumer scales by it (amax protocol; no pass over a tensor)
        # (from 1024 images: below, the in-kernel conversion of the weights costs what the cheaper product saves — cfg2's decoder
        # batch of 512 measured 0.581 ms per step on bf16 pieces, 0.591 on fp16 pairs)
        f16 = (IMG_F16 and n >= IMG_F16_MIN_ROWS and conv4s2_scaled_ok(n, 4, 4, C2, C1) and conv4s2_scaled_ok(n, 8, 8, C3, C2)
               and all(getattr(t, "mvk_amax", None) is not None for t in (wu1, wu2, wd1, wd2)))
        ctx.f16 = f16
        # rotated step (Rotation): the three weight gradients of w0 / w1 / w2 run at the head of the NEXT step — what they read
        # (z, g1, g2 here; the three gradients in backward; the amax slots) lives in the rotation's persistent buffers
        rot = rotation(z2.device)
        if rot is not None:
            rot.wait()  # w0 / w1 / w2 (and their packs above) are final behind the head branch
        if not (ROT_SVHN and rot is not None and f16 and nll_x is not None and LATE_LEAVES and z2.device in _DEFER_ACTIVE
                and all(_is_direct(t) for t in (w0, w1, w2))):
            rot = None
        ctx.rot, ctx.rot_mode = rot, ROT_SVHN
        if ROT_SVHN != 1:
            rot = None  # mode 2 rotates the first layer's gradient only: nothing of the forward pass needs a fixed address
        rkey = ("svhn_dec", w0.data_ptr(), n)
        if rot is not None:
            z2 = rot.buf(rkey + ("z",), z2.shape).copy_(z2)
        if f16:
            pool = AmaxPool(z2, 6) if rot is None else AmaxPool.__new__(AmaxPool)
            if rot is not None:
                pool.t, pool.i = rot.slots(6), 0
            a1, a2 = pool.take(), pool.take()
            a3 = pool.take() if (TAIL_F16 and nll_x is not None) else None  # bound of g3 for the image layer's scaled form
            # bound of the fused tail's stored gradient: the image layer's backward scales it (small_up_bwd_h_kernel)
            ctx.a_dpre = pool.take() if (a3 is not None and TAIL_BWD_F16) else None
            ctx.a_g3 = a3
            ctx.bslots = (pool.take(), pool.take())  # the backward pass's two slots: no fill launch there
            g1 = _new((n, 16 * C1), z2) if rot is None else rot.buf(rkey + ("g1",), (n, 16 * C1))
            if L <= 32 and (16 * C1) % 4 == 0:
                call("mvk_gemm_smallk_amax", ptr(z2), ptr(wp0), ptr(g1), n, 16 * C1, L, 0, ptr(b0), C1, RELU, ptr(a1),
                     stream_ptr())
            else:
                gemm(z2, wp0, n, 16 * C1, L, bias=b0, bias_mod=C1, act=RELU, out=g1)
                amax_of(g1, a1)
            if FWD_DEFER_AT == 4:
                flush_deferred_forward(z2.device)  # a side branch's postponed launches: behind the first (latency-sized) layer
            g2 = conv_up(g1, wu1, b1, n, 4, 4, C2, C1, RELU, amax=(a1, wu1.mvk_amax, a2),
                         out=None if rot is None else rot.buf(rkey + ("g2",), (n, 8, 8, C2)))  # [n,8,8,C2]
            if FWD_DEFER_AT <= 1:
                flush_deferred_forward(z2.device)  # a side branch's postponed launches: beside the 64 -> 32 layer, not the 128 -> 64 one
            g3 = conv_up(g2, wu2, b2, n, 8, 8, C3, C2, RELU, amax=(a2, wu2.mvk_amax, a3))  # [n,16,16,C3]
            if FWD_DEFER_AT == 2:
                flush_deferred_forward(z2.device)  # ... beside the fused tail (HBM-bound: the other resource)
            ctx.wamax = (wd1.mvk_amax, wd2.mvk_amax)
            ctx.gamax = (a1, a2)  # bounds of g1, g2: the V operands of the two weight gradients
        else:
            a3 = None
            g1 = gemm(z2, wp0, n, 16 * C1, L, bias=b0, bias_mod=C1, act=RELU)  # [n,4,4,C1]
            g2 = conv_up(g1, wu1, b1, n, 4, 4, C2, C1, RELU)  # [n,8,8,C2]
            g3 = conv_up(g2, wu2, b2, n, 8, 8, C3, C2, RELU)  # [n,16,16,C3]
        out = _new((n, C4, 32, 32), z2)
        small = bool(_lib.load().mvk_conv4s2_small_up_supported(16, 16, C4, C3))
        ctx.fused = nll_x is not None
        if ctx.fused:
            rows = _new((n,), z2)
            # the stored gradient is pre-multiplied by nll_weight, the weight the rows are expected to enter the loss with
            ctx.nll_weight = float(nll_weight)
            if a3 is not None and ctx.a_dpre is not None:  # ... and the maximum of the stored gradient published for the backward
                call("mvk_conv4s2_small_up_fwd_nll_sy", ptr(g3), ptr(w3), ptr(b3), ptr(nll_x), nll_x.shape[0], float(nll_scale),
                     float(nll_weight), ptr(out), ptr(rows), n, 16, 16, C4, C3, SIGMOID, ptr(a3), ptr(ctx.a_dpre), stream_ptr())
            elif a3 is not None:  # scaled fp16 pairs under the bound the 64 -> 32 launch published (small_up_fwd_h_kernel)
                call("mvk_conv4s2_small_up_fwd_nll_s", ptr(g3), ptr(w3), ptr(b3), ptr(nll_x), nll_x.shape[0], float(nll_scale),
                     float(nll_weight), ptr(out), ptr(rows), n, 16, 16, C4, C3, SIGMOID, ptr(a3), stream_ptr())
            else:
                call("mvk_conv4s2_small_up_fwd_nll_w", ptr(g3), ptr(w3), ptr(b3), ptr(nll_x), nll_x.shape[0], float(nll_scale),
                     float(nll_weight), ptr(out), ptr(rows), n, 16, 16, C4, C3, SIGMOID, stream_ptr())
        elif small:  # per-image MFMA column-matrix kernel (smallconv.hip)
            call("mvk_conv4s2_small_up_fwd", ptr(g3), ptr(w3), ptr(b3), ptr(out), n, 16, 16, C4, C3, SIGMOID,
                 stream_ptr())
        else:
            call("mvk_conv4s2_up_nchw_small", ptr(g3), ptr(w3), ptr(b3), ptr(out), n, 16, 16, C4, C3, SIGMOID,
                 stream_ptr())
        ctx.small = small
        ctx.frags = (wfrag(wd1), wfrag(wd2))
        ctx.save_for_backward(z2, g1, g2, g3, out, wp0, wd1, wd2, wd3, w0, b0, w1, b1, w2, b2, w3, b3)
        _tap("svhn_decoder", w0, g1.view(n, 4, 4, C1), g2, g3)  # NHWC
        ctx.dims = (n, L, C1, C2, C3, C4)
        ctx.z_shape = z.shape
        if ctx.fused:
            return rows.view(*z.shape[:-1])
        return out.view(*z.shape[:-1], C4, 32, 32)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        z2, g1, g2, g3, out, wp0, wd1, wd2, wd3, w0, b0, w1, b1, w2, b2, w3, b3 = ctx.saved_tensors
        n, L, C1, C2, C3, C4 = ctx.dims
        if ctx.fused:  # dout = d loss / d rows [n]; `out` holds nll_weight * d rows / d pre-activation
            c = const_grad(dout)
            if c is not None and c == ctx.nll_weight:
                drows = None  # exactly the weight folded into `out`: no row gradient is read (nor waited for)
            else:
                wait_loss(dout.device)
                drows = _c(dout).reshape(-1)
                if ctx.nll_weight != 1.0:
                    drows = drows * (1.0 / ctx.nll_weight)
            tw3, dw3 = _grad_target(w3)
            tb3, db3 = _grad_target(b3)
            rot = ctx.rot if (ctx.rot is not None and ctx.rot_mode == 1 and rotation(z2.device) is ctx.rot) else None
            rkey = ("svhn_dec", w0.data_ptr(), n)
            dg3 = _new((n, 16, 16, C3), z2) if rot is None else rot.buf(rkey + ("dg3",), (n, 16, 16, C3))
            ws = _ws(z2)
            tb2, db2 = _grad_target(b2)
            if ctx.f16:
                a_dg3 = ctx.bslots[0]
                if ctx.a_dpre is not None:  # scaled fp16 pairs: the gradient under the published bound x max |drows|, g3 under its own
                    call("mvk_conv4s2_small_up_bwd_pre_s", ptr(out), ptr(drows), ptr(g3), RELU, ptr(w3), ptr(dg3), ptr(tw3), ptr(tb3),
                         ptr(tb2), ptr(ws), ws.numel(), n, 16, 16, C4, C3, ptr(ctx.a_dpre), ptr(ctx.a_g3), ptr(a_dg3), stream_ptr())
                else:
                    call("mvk_conv4s2_small_up_bwd_pre_y", ptr(out), ptr(drows), ptr(g3), RELU, ptr(w3), ptr(dg3), ptr(tw3), ptr(tb3),
                         ptr(tb2), ptr(ws), ws.numel(), n, 16, 16, C4, C3, ptr(a_dg3), stream_ptr())
                tail_bwd_done(z2.device)
            else:
                call("mvk_conv4s2_small_up_bwd_pre", ptr(out), ptr(drows), ptr(g3), RELU, ptr(w3), ptr(dg3), ptr(tw3), ptr(tb3),
                     ptr(tb2), ptr(ws), ws.numel(), n, 16, 16, C4, C3, stream_ptr())
        else:
            dout = _c(dout).view(out.shape)
        # last layer: dpre = dout * out(1-out) applied while loading
        if ctx.fused:
            pass
        elif ctx.small:  # backward-data + backward-weight + bias gradient in ONE pass over dout / out / g3
            tw3, dw3 = _grad_target(w3)
            tb3, db3 = _grad_target(b3)
            dg3 = _new((n, 16, 16, C3), z2)
            ws = _ws(z2)
            tb2, db2 = _grad_target(b2)  # bias gradient of the layer below = channel sums of dg3: fused
            call("mvk_conv4s2_small_up_bwd", ptr(dout), ptr(out), SIGMOID, ptr(g3), RELU, ptr(w3), ptr(dg3), ptr(tw3),
                 ptr(tb3), ptr(tb2), ptr(ws), ws.numel(), n, 16, 16, C4, C3, stream_ptr())
        else:
            dw3 = conv_wgrad(dout, g3, w3, n, 16, 16, C4, C3, u_nchw=True, u_act_src=out, u_act=SIGMOID)
            tb3, db3 = _grad_target(b3)
            ws = _ws(dout)
            call("mvk_nchw_channel_sum_acc", ptr(dout), ptr(out), SIGMOID, ptr(tb3), n, C4, 32 * 32, ptr(ws), ws.numel(),
                 stream_ptr())
            dg3 = conv_down(dout, wd3, None, n, 16, 16, C4, C3, NONE, u_nchw=True, u_act_src=out, u_act=SIGMOID,
                            v_act_src=g3, v_act=RELU)
        am2 = am1 = wam2 = wam1 = None
        if ctx.f16:  # the bound of dg3 came with the fused-tail backward; without it the first launch stays on bf16 pieces
            if not ctx.fused:
                a_dg3 = None
            a_dg2 = ctx.bslots[1]
            am2 = (a_dg3, ctx.wamax[1] if a_dg3 is not None else None, a_dg2)
            am1 = (a_dg2, ctx.wamax[0], None)
            wam2, wam1 = (a_dg3, ctx.gamax[1]), (a_dg2, ctx.gamax[0])  # (max |U|, max |V|) of the two weight gradients
        if ctx.f16 and wam1 is not None and LATE_LEAVES and z2.device in _DEFER_ACTIVE:
            # register-stationary weight gradients are this step's late leaves (what an MLP encoder's backward asks: heads_bwd_mlp;
            # a rotated step decides the same way, so that it launches exactly the kernels of the unrotated one)
            big_late_leaves(z2.device)
        rot = ctx.rot if (ctx.fused and ctx.rot is not None and rotation(z2.device) is ctx.rot) else None
        rot0 = rot if (rot is not None and ctx.rot_mode == 2) else None  # mode 2: only the first layer's weight gradient
        if rot0 is not None:
            rot = None
        late = late_leaves(z2.device, dg3, g2, g1, z2)
        if not late.on:
            rot = rot0 = None
            dw2 = conv_wgrad(dg3, g2, w2, n, 8, 8, C3, C2, amax=wam2)
        if not ctx.small and not ctx.fused:
            db2 = colsum(dg3.view(-1, C3), b2)
        dg2, db1 = conv_down(dg3, wd2, None, n, 8, 8, C3, C2, NONE, v_act_src=g2, v_act=RELU, out_bias=b1,
                             frag=ctx.frags[1], amax=am2, out=None if rot is None else rot.buf(rkey + ("dg2",), (n, 8, 8, C2)))
        if not late.on:
            dw1 = conv_wgrad(dg2, g1, w1, n, 4, 4, C2, C1, amax=wam1)
        rany = rot if rot is not None else rot0
        dg1, db0 = conv_down(dg2, wd1, None, n, 4, 4, C2, C1, NONE, v_act_src=g1, v_act=RELU, out_bias=b0, frag=ctx.frags[0],
                             amax=am1, out=None if rany is None else rany.buf(("svhn_dec", w0.data_ptr(), n, "dg1"), (n, 4, 4, C1)))
        dg1f = dg1.view(n, 16 * C1)
        tw0, dw0 = _grad_target(w0)
        dz = None
        if rot is not None:  # rotated step: the three weight gradients are the head of the next step (Rotation.begin_step)
            if ctx.needs_input_grad[0]:
                dz = _svhn_dz(dg1f, wp0, n, L, C1).view(ctx.z_shape)

            def leaves():
                conv_wgrad(dg3, g2, w2, n, 8, 8, C3, C2, amax=wam2)
                conv_wgrad(dg2, g1, w1, n, 4, 4, C2, C1, amax=wam1)
                wsl = _ws(z2)
                call("mvk_unflatten_wgrad", ptr(z2), ptr(dg1f), ptr(_grad_target(w0)[0]), n, L, C1, ptr(wsl), wsl.numel(),
                     stream_ptr())

            def prepare(r):  # behind the update of w0 / w1 / w2: the packs the next forward pass asks for
                jobs = [(w0, "unflatten"), (w1, True, True), (w2, True, True), (w3, True, False)]
                for j, o in zip(jobs, _pack_launch(jobs, am=r.slots(len(jobs)))):
                    r.cache[_job_key(j)] = o

            rot.push(leaves, (w0, w1, w2), prepare)
            return dz, None, db0, None, db1, None, db2, dw3, db3, None, None, None
        if late.on:  # the backward-data chain first, then the weight gradients beside whatever follows it
            if ctx.needs_input_grad[0]:
                dz = _svhn_dz(dg1f, wp0, n, L, C1).view(ctx.z_shape)
            dg2.record_stream(_side_stream(z2.device, 30))
            dg1.record_stream(_side_stream(z2.device, 30))
            def big_leaves():
                conv_wgrad(dg3, g2, w2, n, 8, 8, C3, C2, amax=wam2)
                conv_wgrad(dg2, g1, w1, n, 4, 4, C2, C1, amax=wam1)

            gated_ok = (LATE_GATE and ctx.f16 and wam1 is not None and not SKIP_LATE and _is_direct(w1) and _is_direct(w2)
                        and run_gated(z2.device, big_leaves, dg3, g2, dg2, g1))
            with late:
                dw2 = dw1 = None
                if not SKIP_LATE and not gated_ok:
                    dw2 = conv_wgrad(dg3, g2, w2, n, 8, 8, C3, C2, amax=wam2)
                    dw1 = conv_wgrad(dg2, g1, w1, n, 4, 4, C2, C1, amax=wam1)
                if rot0 is not None:
                    # mode 2: the first layer's weight gradient (a split-K GEMM over z and dg1) is the head of the next step; z is
                    # copied to a buffer with a fixed address HERE, on the late-leaf stream (off the backward chain)
                    zp = rot0.buf(("svhn_dec", w0.data_ptr(), n, "z"), z2.shape).copy_(z2)

                    def leaf0():
                        wsl = _ws(zp)
                        call("mvk_unflatten_wgrad", ptr(zp), ptr(dg1f), ptr(_grad_target(w0)[0]), n, L, C1, ptr(wsl), wsl.numel(),
                             stream_ptr())

                    def prepare0(r):  # behind the update of w0: its pack for the next forward pass
                        job = (w0, "unflatten")
                        r.cache[_job_key(job)] = _pack_launch([job])[0]

                    rot0.push(leaf0, (w0,), prepare0)
                else:
                    ws = _ws(z2)
                    call("mvk_unflatten_wgrad", ptr(z2), ptr(dg1f), ptr(tw0), n, L, C1, ptr(ws), ws.numel(), stream_ptr())
            if dw0 is not None or dw1 is not None or dw2 is not None:  # a gradient autograd itself accumulates: join now
                torch.cuda.current_stream(z2.device).wait_stream(_side_stream(z2.device, 30))
            return dz, dw0, db0, dw1, db1, dw2, db2, dw3, db3, None, None, None
        ws = _ws(z2)
        call("mvk_unflatten_wgrad", ptr(z2), ptr(dg1f), ptr(tw0), n, L, C1, ptr(ws), ws.numel(), stream_ptr())
        if ctx.needs_input_grad[0]:
            dz = _svhn_dz(dg1f, wp0, n, L, C1).view(ctx.z_shape)
        return dz, dw0, db0, dw1, db1, dw2, db2, dw3, db3, None, None, None


def _svhn_dz(dg1f, wp0, n, L, C1):
    """d z = d g1 W0^T of the SVHN decoder's first layer (wp0: the [L][16 C1] unflatten pack)."""
    y = narrow_linear(dg1f, wp0, n, L, 16 * C1, 1, 16 * C1)
    return y if y is not None else gemm(dg1f, wp0, n, L, 16 * C1, tb=True)


def svhn_fused_tail_ok(C4, C3):
    """The fused decoder tail of SVHNDecoderFn exists for this image layer (mvk_conv4s2_small_up_nll_supported)."""
    return bool(_lib.load().mvk_conv4s2_small_up_nll_supported(16, 16, C4, C3))


# =====================================================================================================
# ResNet stacks (models/nn/mmnist.py:214-366, models/nn/cub.py:144-293), NHWC activations
# =====================================================================================================
def _rs_conv(pool, X, xam, wpack, bias, n, H, W, Cin, Cout, **kw):
    """One 3x3 convolution of a ResNet stack -> (result of conv3x3 / conv3x3_f, amax slot of Y or None).  With a pool (the
    scaled-fp16 form is on) and a covered shape the launch is mvk_conv3x3_s: xam = the slot that bounds max |X| (None: it is
    computed here), the weight's bound came with its pack, and max |Y| lands in a fresh slot for the consumer of Y."""
    w_am = getattr(wpack, "mvk_amax", None)
    if (pool is not None and w_am is not None and kw.get("act", NONE) != SIGMOID and kw.get("y_src_act", NONE) != SIGMOID
            and conv3x3_scaled_ok(n, H, W, Cin, Cout)):
        xam = xam if xam is not None else amax_of(X, pool.take())
        yam = pool.take()
        return conv3x3_s(X, wpack, bias, n, H, W, Cin, Cout, xam, w_am, yam, **kw), yam
    if (C3_SPLIT256 and pool is not None and w_am is not None and Cin == 256 and kw.get("res") is None
            and kw.get("act", NONE) != SIGMOID and kw.get("y_src_act", NONE) != SIGMOID and conv3x3_scaled_ok(n, H, W, 128, Cout)):
        # 256 input channels: more weights than a register-stationary wave holds — two launches over 128-channel slices
        xam = xam if xam is not None else amax_of(X, pool.take())
        yam = pool.take()
        kw2 = {k: v for k, v in kw.items() if k not in ("res", "res_alpha")}
        return conv3x3_s_split(X, wpack, bias, n, H, W, Cin, Cout, xam, w_am, yam, **kw2), yam
    if kw.get("x_act", NONE) != NONE or kw.get("pre_scale", 1.0) != 1.0:
        return conv3x3_f(X, wpack, bias, n, H, W, Cin, Cout, **kw), None
    kw.pop("x_act", None)
    kw.pop("pre_scale", None)
    if pool is not None and Cin <= 4 and Cout % 4 == 0 and kw.get("res") is None and C3_Y_AMAX:
        # an image on the input side (conv_img of an encoder, the backward-data pass of a decoder's conv_img): the direct
        # kernel publishes max |Y| itself — the stack behind it takes the scaled-fp16 form without an amax pass over Y
        yam = pool.take()
        try:
            return conv3x3(X, wpack, bias, n, H, W, Cin, Cout, y_amax=yam, **kw), yam
        except _lib.MvkError:
            pass  # shape not covered by the direct kernel (MVK_EINVAL): the plain launch below
    return conv3x3(X, wpack, bias, n, H, W, Cin, Cout, **kw), None


def _rs_wgrad(pool, X, xam, dY, dyam, wparam, bparam, n, H, W, Cin, Cout, x_act=NONE, dy_scale=1.0, fused=False):
    """Weight (and bias) gradient of one 3x3 convolution of a ResNet stack -> (dW ref, db ref or None).  With a pool and a
    covered shape: mvk_conv3x3_wgrad_s (xam / dyam = slots bounding max |X| / max |dY|, computed here when unknown); else the
    bf16-piece launch (fused: bias gradient, x_act and dy_scale in it) or the plain one + a column-sum pass."""
    if pool is not None and conv3x3_wgrad_scaled_ok(n, H, W, Cin, Cout):
        xam = xam if xam is not None else amax_of(X, pool.take())
        dyam = dyam if dyam is not None else amax_of(dY, pool.take())
        return conv3x3_wgrad_s(X, dY, wparam, bparam, n, H, W, Cin, Cout, xam, dyam, x_act=x_act, dy_scale=dy_scale)
    if fused:
        return conv3x3_wgrad_f(X, dY, wparam, bparam, n, H, W, Cin, Cout, x_act=x_act, dy_scale=dy_scale)
    if x_act != NONE or dy_scale != 1.0:
        raise _lib.MvkError("_rs_wgrad: x_act / dy_scale need the fused launch")
    rw = conv3x3_wgrad(X, dY, wparam, n, H, W, Cin, Cout)
    return rw, (colsum(dY.view(-1, Cout), bparam) if bparam is not None else None)


class ResnetStackFn(Function):
    """x [n,H,W,C] NHWC -> a static program of layers -> y NHWC, one autograd node.

    program: list of tuples (static Python data; parameter references are indices into `params`)
      ("conv", iw, ib, act)                      3x3/1/1 convolution (+ bias, + activation)
      ("block", order, iw1, ib1, iw2, ib2, isc)  ResnetBlock; order "post": xs + 0.1*lrelu(conv2(lrelu(conv1(x))))
                                                 (mmnist.py:229-246), "pre": xs + 0.1*conv2(lrelu(conv1(lrelu(x))))
                                                 (cub.py:274-280); isc = 1x1 shortcut weight index or None
      ("pool",)                                  AvgPool2d(3, 2, 1)
      ("up",)                                    Upsample(scale_factor=2), nearest
      ("act", code)                              elementwise activation (cub.py: fc(actvn(out)), conv_img(actvn(out)))
    """

    @staticmethod
    def forward(ctx, x, program, *params):
        x = _c(x)
        n, H, W, C = x.shape
        tape = []  # per layer: what backward needs
        packs = {}
        jobs, order = [], []
        for op in program:
            idxs = [op[1]] if op[0] == "conv" else ([op[2], op[4]] if op[0] == "block" else [])
            for i in idxs:
                if i not in packs:
                    packs[i] = None
                    jobs.append((params[i], "c3", True, True))
                    order.append(i)
        for i0 in range(0, len(jobs), PACK_MAX):
            for i, pk in zip(order[i0:i0 + PACK_MAX], pack_weights(jobs[i0:i0 + PACK_MAX])):
                packs[i] = pk
        h = x
        sites = []  # sign sources of the activation sites in the reference's evaluation order (test hook, see TAPS)
        # scaled-fp16 form: ham = the slot that bounds max |h| (None: unknown, computed on demand)
        pool = AmaxPool(x, 4 * len(program) + 4) if C3_F16 else None
        ham = None
        for op in program:
            if op[0] == "conv":
                _, iw, ib, act = op
                Cout = params[iw].shape[0]
                y, yam = _rs_conv(pool, h, ham, packs[iw][0], params[ib] if ib is not None else None, n, H, W, C, Cout, act=act)
                tape.append((h, y, (H, W, C, Cout), ham))
                if act != NONE:
                    sites.append(y)
                h, C, ham = y, Cout, yam
            elif op[0] == "block":
                _, order_, iw1, ib1, iw2, ib2, isc = op
                Chid, Cout = params[iw1].shape[0], params[iw2].shape[0]
                b1 = params[ib1] if ib1 is not None else None
                b2 = params[ib2] if ib2 is not None else None
                if pool is not None and ham is None and conv3x3_scaled_ok(n, H, W, C, Chid):
                    ham = amax_of(h, pool.take())  # once: conv1 now, the weight gradient of conv1 in the backward pass
                # fused forms (register-stationary kernels): the leading LeakyReLU of a pre-activation block is applied while
                # conv1 stages its input (a0 is never written) and the backward pass folds the 0.1 and the bias gradients
                fused = (conv3x3_fused_ok(n, H, W, C, Chid) and conv3x3_fused_ok(n, H, W, Chid, Cout)
                         and conv3x3_fused_ok(n, H, W, Cout, Chid))  # conv1, conv2 and conv2's backward-data launch
                oam = y2am = None
                out = None
                if order_ == "post":
                    a0 = h
                    a1, a1am = _rs_conv(pool, a0, ham, packs[iw1][0], b1, n, H, W, C, Chid, act=LEAKY)
                    w2am = getattr(packs[iw2][0], "mvk_amax", None)
                    if (C3_DUAL and pool is not None and a1am is not None and w2am is not None
                            and conv3x3_scaled_ok(n, H, W, Chid, Cout)):
                        # conv2, its LeakyReLU and the block's sum in one launch; y2 (the backward pass needs its signs) is the
                        # launch's second store, max |out| its published bound
                        xs = h if isc is None else linear_fwd(h.view(-1, C), params[isc].view(Cout, C), None, NONE).view(n, H, W, Cout)
                        oam = pool.take()
                        out, y2 = conv3x3_s2(a1, packs[iw2][0], b2, n, H, W, Chid, Cout, a1am, w2am, oam, LEAKY, xs, 0.1)
                    else:
                        y2, y2am = _rs_conv(pool, a1, a1am, packs[iw2][0], b2, n, H, W, Chid, Cout, act=LEAKY)
                elif fused:
                    a0 = None
                    a1, a1am = _rs_conv(pool, h, ham, packs[iw1][0], b1, n, H, W, C, Chid, act=LEAKY, x_act=LEAKY)
                    y2 = None
                else:
                    a0 = axpby(h, 1.0, None, 0.0, act=LEAKY)
                    a1, a1am = _rs_conv(pool, a0, ham, packs[iw1][0], b1, n, H, W, C, Chid, act=LEAKY)
                    y2 = None  # the block's sum is formed in conv2's epilogue; backward does not need conv2's output
                if out is None:
                    xs = h if isc is None else linear_fwd(h.view(-1, C), params[isc].view(Cout, C), None, NONE).view(n, H, W, Cout)
                if out is not None:
                    pass
                elif y2 is None:
                    out, oam = _rs_conv(pool, a1, a1am, packs[iw2][0], b2, n, H, W, Chid, Cout, act=NONE, res=xs, res_alpha=0.1)
                else:
                    out = axpby(xs, 1.0, y2, 0.1)
                    if isc is None and ham is not None and y2am is not None:
                        oam = torch.add(ham, y2am, alpha=0.1)  # |xs + 0.1 y2| <= max |xs| + 0.1 max |y2|: no pass over `out`
                tape.append((h, a0, a1, y2, (H, W, C, Chid, Cout), fused, ham, a1am))
                sites += [a1, y2] if order_ == "post" else [h, a1]
                h, C, ham = out, Cout, oam
            elif op[0] == "pool":
                y = avgpool(h, n, H, W, C)  # an average, a copy, a (leaky) ReLU: max |h| still bounds the result
                tape.append((H, W, C))
                h, H, W = y, (H + 1) // 2, (W + 1) // 2
            elif op[0] == "up":
                y = upsample2(h, n, H, W, C)
                tape.append((H, W, C))
                h, H, W = y, 2 * H, 2 * W
            elif op[0] == "act":
                y = axpby(h, 1.0, None, 0.0, act=op[1])
                tape.append((y,))
                sites.append(y)
                h = y
                ham = ham if op[1] in (LEAKY, RELU) else None
            else:
                raise _lib.MvkError(f"unknown ResNet op {op[0]!r}")
        ctx.tape, ctx.packs, ctx.program, ctx.n = tape, packs, program, n
        ctx.save_for_backward(*params)
        _tap("resnet_stack", params[0] if params else None, *sites)
        return h

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        params = ctx.saved_tensors
        n, packs = ctx.n, ctx.packs
        grads = [None] * len(params)
        g = _c(dout)
        need_dx = ctx.needs_input_grad[0]
        pool = AmaxPool(g, 4 * len(ctx.program) + 4) if C3_F16 else None
        gam = None  # the slot that bounds max |g| (None: unknown)
        for li in range(len(ctx.program) - 1, -1, -1):
            op, rec = ctx.program[li], ctx.tape[li]
            first = li == 0
            if op[0] == "conv":
                _, iw, ib, act = op
                xin, y, (H, W, C, Cout), xam = rec
                dpre = g
                if act != NONE:
                    dpre = g.clone() if g is dout else g
                    call("mvk_act_bwd", ptr(dpre), ptr(y), dpre.numel(), act, stream_ptr())
                gam = gam if act in (NONE, LEAKY, RELU) else None  # |dpre| <= |g|: these derivatives are at most 1
                grads[iw], gb = _rs_wgrad(pool, xin, xam, dpre, gam, params[iw], params[ib] if ib is not None else None, n, H, W,
                                          C, Cout)
                if ib is not None:
                    grads[ib] = gb
                if not first or need_dx:
                    g, gam = _rs_conv(pool, dpre, gam, packs[iw][1], None, n, H, W, Cout, C)
            elif op[0] == "block":
                _, order_, iw1, ib1, iw2, ib2, isc = op
                xin, a0, a1, y2, (H, W, C, Chid, Cout), fused, xam, a1am = rec
                gout = g
                if gam is None and pool is not None:
                    gam = amax_of(gout, pool.take())  # three launches below take it
                if fused and order_ == "pre":
                    # d2 = 0.1 * gout is never formed: the 0.1 rides on the weight / bias gradient and on the backward-data
                    # sum; a0 = lrelu(x) is re-applied while the weight-gradient kernel stages x
                    grads[iw2], gb2 = _rs_wgrad(pool, a1, a1am, gout, gam, params[iw2], params[ib2] if ib2 is not None else None,
                                                n, H, W, Chid, Cout, dy_scale=0.1, fused=True)
                    if ib2 is not None:
                        grads[ib2] = gb2
                    if ib1 is not None:
                        (d1, grads[ib1]), d1am = _rs_conv(pool, gout, gam, packs[iw2][1], None, n, H, W, Cout, Chid, y_act_src=a1,
                                                          y_src_act=LEAKY, out_bias=params[ib1], pre_scale=0.1)
                    else:
                        d1, d1am = _rs_conv(pool, gout, gam, packs[iw2][1], None, n, H, W, Cout, Chid, y_act_src=a1,
                                            y_src_act=LEAKY, pre_scale=0.1)
                    grads[iw1], _ = _rs_wgrad(pool, xin, xam, d1, d1am, params[iw1], None, n, H, W, C, Chid, x_act=LEAKY, fused=True)
                    a0 = xin  # sign(lrelu(x)) == sign(x): the mask source of dx below
                else:
                    if order_ == "post":  # gradient w.r.t. conv2's pre-activation: 0.1 * gout * lrelu'(y2), one pass
                        d2 = _new(gout.shape, gout)
                        call("mvk_act_bwd_scaled", ptr(gout), 0.1, ptr(y2), LEAKY, ptr(d2), d2.numel(), stream_ptr())
                    else:
                        d2 = axpby(gout, 0.1, None, 0.0)  # gradient w.r.t. y2 (max |gout| bounds it)
                    grads[iw2], gb2 = _rs_wgrad(pool, a1, a1am, d2, gam, params[iw2], params[ib2] if ib2 is not None else None, n, H,
                                                W, Chid, Cout, fused=fused)  # fused: bias gradient with the weight gradient
                    if ib2 is not None:
                        grads[ib2] = gb2
                    # backward data of conv2 with lrelu'(a1) fused; its channel sums are conv1's bias gradient
                    if ib1 is not None:
                        (d1, grads[ib1]), d1am = _rs_conv(pool, d2, gam, packs[iw2][1], None, n, H, W, Cout, Chid, y_act_src=a1,
                                                          y_src_act=LEAKY, out_bias=params[ib1])
                    else:
                        d1, d1am = _rs_conv(pool, d2, gam, packs[iw2][1], None, n, H, W, Cout, Chid, y_act_src=a1, y_src_act=LEAKY)
                    grads[iw1], _ = _rs_wgrad(pool, a0, xam, d1, d1am, params[iw1], None, n, H, W, C, Chid)
                if isc is not None:
                    grads[isc], _ = linear_bwd_weight(gout.view(-1, Cout), xin.view(-1, C), params[isc].view(Cout, C), None)
                    if grads[isc] is not None:
                        grads[isc] = grads[isc].view(params[isc].shape)
                if not first or need_dx:
                    # the shortcut's gradient first, the convolution path adds itself to it in its epilogue
                    gsc = gout if isc is None else linear_bwd_data(gout.view(-1, Cout), params[isc].view(Cout, C)).view(n, H, W, C)
                    if order_ == "post":
                        dx, gam = _rs_conv(pool, d1, d1am, packs[iw1][1], None, n, H, W, Chid, C, res=gsc)
                    else:  # through the leading LeakyReLU: multiply by lrelu'(x) = lrelu'(a0)
                        dx, gam = _rs_conv(pool, d1, d1am, packs[iw1][1], None, n, H, W, Chid, C, y_act_src=a0, y_src_act=LEAKY,
                                           res=gsc)
                    g = dx
            elif op[0] == "pool":
                H, W, C = rec
                g = avgpool_bwd(g, n, H, W, C)  # every input position collects at most 4 window shares of 1/9: the bound holds
            elif op[0] == "act":
                g = g.clone() if g is dout else g
                call("mvk_act_bwd", ptr(g), ptr(rec[0]), g.numel(), op[1], stream_ptr())
            else:
                H, W, C = rec
                g = upsample2_bwd(g, n, H, W, C)
                gam = gam * 4.0 if gam is not None else None  # sums of 4 entries: a looser bound costs range, not precision
        return (g if need_dx else None, None, *grads)


# =====================================================================================================
# Fused posterior kernels
# =====================================================================================================
class MoPoEPosteriorFn(Function):
    """(mu_m, lv_m)_m -> z[K,B,L], kld_rows[B] (+ subset statistics); see mvk_mopoe_posterior_fwd.

    Inputs are given in PoE summation order (sorted modality names); subset_masks refer to that order.
    """

    @staticmethod
    def forward(ctx, eps, subset_masks, sel, weights, want_stats, *mus_lvs):
        M = len(mus_lvs) // 2
        mus = [_c(t) for t in mus_lvs[:M]]
        lvs = [_c(t) for t in mus_lvs[M:]]
        K, B, L = eps.shape
        S = subset_masks.numel()
        z = _new((K, B, L), eps)
        kld_rows = _new((B,), eps)
        mus_out = _new((S, B, L), eps) if want_stats else None
        lvs_out = _new((S, B, L), eps) if want_stats else None
        jmu = _new((B, L), eps) if want_stats else None
        jlv = _new((B, L), eps) if want_stats else None
        call("mvk_mopoe_posterior_fwd", ptr_array(mus), ptr_array(lvs), M, ptr(subset_masks), S, ptr(sel),
             ptr(weights), ptr(eps), K, B, L, ptr(z), ptr(kld_rows), ptr(mus_out), ptr(lvs_out), ptr(jmu), ptr(jlv),
             stream_ptr())
        ctx.save_for_backward(eps, subset_masks, sel, *mus, *lvs)
        ctx.weights = weights
        ctx.M = M
        if want_stats:
            ctx.mark_non_differentiable(mus_out, lvs_out, jmu, jlv)
            return z, kld_rows, mus_out, lvs_out, jmu, jlv
        return z, kld_rows

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dkld_rows, *unused):
        dev = ctx.saved_tensors[0].device
        # the gradient of the KL rows: ONE constant the host knows when the loss node registered it (const_grad: the unit-seed
        # path) — read from a kept buffer of that constant, and the assembly launch has no reader inside the step; else it comes
        # out of the assembly launch, which this node then waits for (or runs, when it was postponed)
        c = const_grad(dkld_rows) if dkld_rows is not None else None
        gk = const_rows(dev, dkld_rows.numel(), c) if c is not None else None
        if gk is None:
            wait_loss(dev)
            gk = _c(dkld_rows) if dkld_rows is not None else None
        late_ready(dev)
        saved = ctx.saved_tensors
        eps, subset_masks, sel = saved[:3]
        M = ctx.M
        mus, lvs = saved[3 : 3 + M], saved[3 + M :]
        K, B, L = eps.shape
        dz = _c(dz) if dz is not None else torch.zeros_like(eps)
        dmus = [_new((B, L), eps) for _ in range(M)]
        dlvs = [_new((B, L), eps) for _ in range(M)]
        defer_flush_side(eps.device)  # the decoders are done: finish their gradients beside the encoder backward
        call("mvk_mopoe_posterior_bwd", ptr_array(mus), ptr_array(lvs), M, ptr(subset_masks), subset_masks.numel(),
             ptr(sel), ptr(ctx.weights), ptr(eps), ptr(dz), K, B, L, ptr(gk), ptr_array(dmus), ptr_array(dlvs),
             stream_ptr())
        return (None, None, None, None, None, *dmus, *dlvs)


class MVTCAEPosteriorFn(Function):
    """(mu_m, lv_m)_m -> z[K,B,L], joint_kl_rows[B], cond_kl_rows[M,B], joint_mu, joint_lv."""

    @staticmethod
    def forward(ctx, eps, masks, *mus_lvs):
        M = len(mus_lvs) // 2
        mus = [_c(t) for t in mus_lvs[:M]]
        lvs = [_c(t) for t in mus_lvs[M:]]
        K, B, L = eps.shape
        z = _new((K, B, L), eps)
        jkl = _new((B,), eps)
        ckl = _new((M, B), eps)
        jmu, jlv = _new((B, L), eps), _new((B, L), eps)
        marr = ptr_array(masks) if masks is not None else None
        call("mvk_mvtcae_posterior_fwd", ptr_array(mus), ptr_array(lvs), marr, M, ptr(eps), K, B, L, ptr(z),
             ptr(jkl), ptr(ckl), ptr(jmu), ptr(jlv), stream_ptr())
        ctx.save_for_backward(eps, *mus, *lvs)
        ctx.masks = masks
        ctx.M = M
        ctx.mark_non_differentiable(jmu, jlv)
        return z, jkl, ckl, jmu, jlv

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, djkl, dckl, *unused):
        wait_loss(ctx.saved_tensors[0].device)  # the gradients of the KL rows come out of the loss assembly launch
        late_ready(ctx.saved_tensors[0].device)
        saved = ctx.saved_tensors
        eps = saved[0]
        M = ctx.M
        mus, lvs = saved[1 : 1 + M], saved[1 + M :]
        K, B, L = eps.shape
        dz = _c(dz) if dz is not None else torch.zeros_like(eps)
        gj = _c(djkl) if djkl is not None else None
        gc = _c(dckl) if dckl is not None else None
        dmus = [_new((B, L), eps) for _ in range(M)]
        dlvs = [_new((B, L), eps) for _ in range(M)]
        marr = ptr_array(ctx.masks) if ctx.masks is not None else None
        call("mvk_mvtcae_posterior_bwd", ptr_array(mus), ptr_array(lvs), marr, M, ptr(eps), ptr(dz), K, B, L,
             ptr(gj), ptr(gc), ptr_array(dmus), ptr_array(dlvs), stream_ptr())
        return (None, None, *dmus, *dlvs)


class JMVAEPosteriorFn(Function):
    """joint (mu, lv), unimodal (mu_m, lv_m)_m -> z[K,B,L], kld_rows[B], ljm_rows[B]  (jmvae_model.py:133-174)."""

    @staticmethod
    def forward(ctx, eps, jmu, jlv, *mus_lvs):
        M = len(mus_lvs) // 2
        jmu, jlv = _c(jmu), _c(jlv)
        mus = [_c(t) for t in mus_lvs[:M]]
        lvs = [_c(t) for t in mus_lvs[M:]]
        K, B, L = eps.shape
        z = _new((K, B, L), eps)
        kld, ljm = _new((B,), eps), _new((B,), eps)
        call("mvk_jmvae_posterior_fwd", ptr(jmu), ptr(jlv), ptr_array(mus), ptr_array(lvs), M, ptr(eps), K, B, L,
             ptr(z), ptr(kld), ptr(ljm), stream_ptr())
        ctx.save_for_backward(eps, jmu, jlv, *mus, *lvs)
        ctx.M = M
        return z, kld, ljm

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dkld, dljm):
        wait_loss(ctx.saved_tensors[0].device)  # the gradients of the KL rows come out of the loss assembly launch
        late_ready(ctx.saved_tensors[0].device)
        saved = ctx.saved_tensors
        eps, jmu, jlv = saved[:3]
        M = ctx.M
        mus, lvs = saved[3 : 3 + M], saved[3 + M :]
        K, B, L = eps.shape
        dz = _c(dz) if dz is not None else None
        gk = _c(dkld) if dkld is not None else None
        gj = _c(dljm) if dljm is not None else None
        djmu, djlv = _new((B, L), eps), _new((B, L), eps)
        dmus = [_new((B, L), eps) for _ in range(M)]
        dlvs = [_new((B, L), eps) for _ in range(M)]
        call("mvk_jmvae_posterior_bwd", ptr(jmu), ptr(jlv), ptr_array(mus), ptr_array(lvs), M, ptr(eps), ptr(dz), K, B,
             L, ptr(gk), ptr(gj), ptr(djmu), ptr(djlv), ptr_array(dmus), ptr_array(dlvs), stream_ptr())
        return (None, djmu, djlv, *dmus, *dlvs)


# =====================================================================================================
# Fused reconstruction NLL + scalar assembly (single autograd node producing the loss)
# =====================================================================================================
_UNIT_SEEDS = {}  # data_ptr -> tensor (kept alive): backward seeds known to hold exactly 1


def unit_seed(like):
    """A ones tensor shaped like the loss `like`, registered as THE unit backward seed of its device: pass it as
    `loss.backward(gradient=...)` and ReconLossFn.backward needs no launch (autograd's own default seed is a fresh tensor
    whose value the host cannot know without a synchronisation)."""
    key = (like.device, tuple(like.shape))
    t = _UNIT_SEEDS.get(key)
    if t is None:
        t = torch.ones_like(like)
        _UNIT_SEEDS[key] = t
        _UNIT_SEEDS[t.data_ptr()] = t
    return t


UNIT_SEED = _lib.tune("MVK_UNIT_SEED", "1") != "0"  # 0: always launch the seed kernel (A/B)


def is_unit_seed(g):
    if not UNIT_SEED:
        return False
    t = _UNIT_SEEDS.get(g.data_ptr())
    return t is not None and t.shape == g.shape and t.device == g.device


TERMS_MULTI_WG = _lib.tune("MVK_TERMS_WS", "1") != "0"  # 0: the one-workgroup launch (A/B)
_TERMS_WS = {}  # (device, stream) -> the arrival counter + partials of mvk_reduce_terms_ws (zero between launches)


def _reduce_terms(terms, n_terms, loss_sum_scale, out, loss):
    """The scalar assembly on the current stream, on several workgroups where a term is long (mvk_reduce_terms_ws)."""
    if not TERMS_MULTI_WG:
        call("mvk_reduce_terms", terms, n_terms, loss_sum_scale, ptr(out), ptr(loss), stream_ptr())
        return
    key = (out.device, torch.cuda.current_stream(out.device).cuda_stream)
    ws = _TERMS_WS.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            # a workspace made HERE would live in the capturing graph's private pool, zero-filled by a node of that
            # graph only, and then be handed to every later capture through this cache.  The warm-up passes of GraphedStep run on
            # another stream than the capture: inside a capture the launch takes the one-workgroup form, which needs no workspace
            # (same sums in the same order per term; bit-identical results are tested for both forms)
            call("mvk_reduce_terms", terms, n_terms, loss_sum_scale, ptr(out), ptr(loss), stream_ptr())
            return
        ws = torch.zeros(1 + 32 * 64, dtype=torch.float32, device=out.device)  # MVK_REDUCE_TERMS_WS_FLOATS
        _TERMS_WS[key] = ws
    call("mvk_reduce_terms_ws", terms, n_terms, loss_sum_scale, ptr(out), ptr(loss), ptr(ws), ws.numel(), stream_ptr())


def terms_workspace(device, stream):
    """Make the assembly's workspace of (device, stream) now, outside any capture (trainers/graph.py: before torch.cuda.graph, for
    the capture stream and the late-leaf stream the assembly may run on).  Eager memory, zeroed once, left zero by every launch."""
    key = (device, stream.cuda_stream)
    if key not in _TERMS_WS and TERMS_MULTI_WG:
        _TERMS_WS[key] = torch.zeros(1 + 32 * 64, dtype=torch.float32, device=device)


class ReconLossFn(Function):
    """loss = sum_i lossw_i * coef_i * sum_{k,b} mask_i[b] rows_i[k,b] + sum_j lossw_j * coef_j * sum(extra_j).

    forward: ONE mvk_recon_nll_fwd launch for all modalities (row NLLs + d loss / d recon assuming an upstream
             gradient of 1) and ONE mvk_reduce_terms launch.  Returns (loss, terms[n_terms+2]) where terms holds
             every individual term, the loss and loss * loss_sum_scale (all detached: metrics).
    spec:    dict(K, B, x[], masks[], dist[], scale[], rescale[], coef[], lossw[], extra_coef[], extra_lossw[],
             extra_split[], loss_sum_scale).
    extras:  tensors (KL rows) whose sums enter the loss with a constant weight; with extra_split[j] = n the tensor is
             n equal chunks, one term each, and extra_coef[j] may be a list of n per-chunk coefficients.
    pairs:   optional spec["pairs"] = [(tensor index, slab)]: reconstruction term i scores ONE [B, D] slab of the
             decoder output tensors[tensor index] ([K_t, B, D]) against x[i] with its own mask / coefficient (MVAE: a
             decoder runs once over the samples of all its subsets, every (modality, subset) pair is a term).  The
             per-entry lists of spec are then indexed by pair; gradients are still one buffer per decoder output.
    """

    @staticmethod
    def forward(ctx, spec, n_mod, *tensors):
        recons = [_c(t) for t in tensors[:n_mod]]
        extras = [_c(t) for t in tensors[n_mod:]]
        xs, masks = spec["x"], spec["masks"]
        K, B = spec["K"], spec["B"]
        pairs = spec.get("pairs")
        ref = recons[0] if recons else extras[0]
        drecons = [torch.empty_like(r) if ctx.needs_input_grad[2 + i] else None for i, r in enumerate(recons)]
        if pairs is not None:  # every term is one slab: K = 1 launches; untouched slabs of a buffer get no gradient
            used = {}
            for j, k in pairs:
                used.setdefault(j, set()).add(k)
            for j, g in enumerate(drecons):
                if g is not None and len(used.get(j, ())) != g.shape[0]:
                    g.zero_()
        n_rec = len(pairs) if pairs is not None else n_mod
        descs = (ReconDesc * max(n_rec, 1))()
        rows = []
        for i in range(n_rec):
            j, k = pairs[i] if pairs is not None else (i, None)
            Kt = recons[j].shape[0] if pairs is not None else K
            D = recons[j].numel() // (Kt * B)
            off = 0 if k is None else 4 * k * B * D
            r = _new((1 if pairs is not None else K, B), ref)
            rows.append(r)
            d = descs[i]
            d.recon, d.x = recons[j].data_ptr() + off, xs[i].data_ptr()
            d.mask = masks[i].data_ptr() if masks[i] is not None else None
            d.rows = r.data_ptr()
            d.drecon = drecons[j].data_ptr() + off if drecons[j] is not None else None
            d.rowcoef = None
            d.D, d.dist, d.n_classes = D, spec["dist"][i], recons[j].shape[-1]
            d.scale, d.rescale = spec["scale"][i], spec["rescale"][i]
            d.coef = spec["coef"][i] * spec["lossw"][i]
        if n_rec:
            prof = PROFILE.get("recon_nll")
            if prof is not None:
                # measurement only: a short spin kernel ahead of the first event lets the host finish enqueueing the
                # launch while the GPU is busy, so the event pair brackets the kernel and not the launch latency
                if PROFILE.get("presleep_cycles"):
                    torch.cuda._sleep(int(PROFILE["presleep_cycles"]))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            Kl = 1 if pairs is not None else K
            for i0 in range(0, n_rec, _lib.MAX_MODALITIES):  # MVK_MAX_MODALITIES descriptors per launch
                n = min(_lib.MAX_MODALITIES, n_rec - i0)
                call("mvk_recon_nll_fwd", C.cast(C.byref(descs[i0]), C.POINTER(ReconDesc)), n, Kl, B, stream_ptr())
            if prof is not None:
                e1.record()
                prof.append((e0, e1))
        splits = spec.get("extra_split") or [1] * len(extras)
        n_terms = n_rec + sum(splits)
        terms = (TermDesc * n_terms)()
        for i in range(n_rec):
            t = terms[i]
            t.v, t.n = rows[i].data_ptr(), rows[i].numel()
            t.mask = masks[i].data_ptr() if masks[i] is not None else None
            t.period, t.coef, t.lossw = B, spec["coef"][i], spec["lossw"][i]
        ti = n_rec
        extra_grad = []  # (byte offset, numel, coefficient) per chunk of every extra tensor
        # the gradients of the extra tensors for an upstream gradient of 1 are constants: mvk_reduce_terms writes them
        # (backward then launches nothing when its upstream gradient is the unit seed)
        dextras = [torch.empty_like(e) if ctx.needs_input_grad[2 + n_mod + j] else None for j, e in enumerate(extras)]
        for j, e in enumerate(extras):
            ns = splits[j]
            chunk = e.numel() // ns
            cj = spec["extra_coef"][j]
            for c in range(ns):  # one term per chunk (e.g. per-modality KL rows), one gradient per tensor
                t = terms[ti]
                cc = cj[c] if isinstance(cj, (list, tuple)) else cj
                t.v, t.n, t.mask, t.period = e.data_ptr() + 4 * c * chunk, chunk, None, 1
                t.coef, t.lossw = cc, spec["extra_lossw"][j]
                t.gfill = dextras[j].data_ptr() + 4 * c * chunk if dextras[j] is not None else None
                extra_grad.append((j, 4 * c * chunk, chunk, cc * spec["extra_lossw"][j]))
                ti += 1
        out = _new((n_terms + 2,), ref)
        loss = _new((), ref)
        # Every reconstruction term came from a fused decoder tail (n_rec == 0) and the model vouches for its backward nodes
        # (spec["async_ok"]: they call wait_loss before reading a gradient this launch fills): the assembly is a leaf of the
        # step — nothing on the backward chain needs the loss VALUE — so it runs on the late-leaf stream, joined where the
        # deferred finishes run, instead of between the last forward and the first backward launch of the critical chain.
        late = late_leaves(ref.device, *extras) if (ASYNC_LOSS and n_rec == 0 and spec.get("async_ok")) else None
        dev = ref.device
        if late is not None and late.on and ASSEMBLY_LAST and spec.get("assembly_last") and dev in _DEFER_ACTIVE:
            # The model vouches that NO backward node reads what this launch fills when the backward seed is the unit seed (its
            # posterior node takes the KL rows' constant gradient from const_grad): the launch is enqueued with the postponed
            # leaves at the end of the scope (run_last: the late-leaf stream, behind the large decoder's weight gradients) instead
            # of at the head of the backward pass, where its 18 workgroups wait 50-80 us for slots beside the image layer's 512
            # persistent workgroups — and delay them.  A node that does read (general seed, style latents) runs it first: wait_loss.
            keep = [out, loss] + [d for d in dextras if d is not None]

            def assemble():
                _LOSS_POSTPONED.pop(dev, None)
                _reduce_terms(terms, n_terms, spec["loss_sum_scale"], out, loss)
                st = torch.cuda.current_stream(dev)
                for t in keep:
                    t.record_stream(st)

            if run_last(dev, assemble, *extras, force=True):
                _LOSS_POSTPONED[dev] = assemble
            else:
                assemble()
        elif late is not None and late.on:
            # (postponing this launch behind the image layer's backward kernel — it competes with that kernel's persistent
            # workgroups for CU slots at the head of the backward pass — was measured: +60 us per step, the ninth cross-stream
            # edge inside the captured step that lost; profiles/NOTES_r05.md section 9)
            with late:
                _reduce_terms(terms, n_terms, spec["loss_sum_scale"], out, loss)
                st = torch.cuda.current_stream(ref.device)
                _LOSS_EVENT[ref.device] = st.record_event()
                for t in [out, loss] + [d for d in dextras if d is not None]:
                    t.record_stream(st)
        else:
            _reduce_terms(terms, n_terms, spec["loss_sum_scale"], out, loss)
        ctx.drecons = drecons
        ctx.dextras = dextras
        ctx.extra_shapes = [e.shape for e in extras]
        ctx.extra_grad = extra_grad
        ctx.rows = rows  # keep alive: metrics / debugging
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)  # no zero-fill launch for the gradient of `out` (never used)
        return loss, out

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss, gout):
        if gloss is None:  # only the (non-differentiable) terms were used
            return (None, None) + (None,) * (len(ctx.drecons) + len(ctx.extra_shapes))
        grads = list(ctx.drecons)
        if is_unit_seed(gloss):  # d loss = 1 exactly: every gradient below was written by the forward launches
            extras = list(ctx.dextras)
            # a single-term extra tensor's gradient is ONE constant the host knows: a fused decoder tail whose stored gradient
            # was pre-multiplied by it (MLPDecoderFn / SVHNDecoderFn `nll_weight`) then reads no row gradient at all
            per = {}
            for j, _off, _n, c in ctx.extra_grad:
                per.setdefault(j, []).append(c)
            for j, cs in per.items():
                if extras[j] is not None and len(set(cs)) == 1:
                    register_const_grad(extras[j], cs[0])
            ctx.drecons = ctx.dextras = None
            return (None, None, *grads, *extras)
        wait_loss(gloss.device)  # the seed launch below rewrites what the assembly launch filled
        gloss = _c(gloss.reshape(1))
        extras = [d if d is not None else _new(shape, gloss) for d, shape in zip(ctx.dextras, ctx.extra_shapes)]
        jobs = [(g.data_ptr(), g.numel(), 1.0, 0) for g in ctx.drecons if g is not None]
        jobs += [(extras[j].data_ptr() + off, n, c, 1) for j, off, n, c in ctx.extra_grad]
        for i0 in range(0, len(jobs), 12):  # MVK_SEED_MAX buffers per launch: normally ONE launch
            chunk = jobs[i0:i0 + 12]
            descs = (SeedDesc * len(chunk))()
            for d, (p, n, c, fill) in zip(descs, chunk):
                d.buf, d.n, d.coef, d.fill = p, n, float(c), fill
            call("mvk_loss_backward_seed", descs, len(chunk), ptr(gloss), stream_ptr())
        ctx.drecons = ctx.dextras = None
        return (None, None, *grads, *extras)


class GaussSampleKLFn(Function):
    """(mu, lv) [B,S], eps [K,B,S] -> w [K,B,S], kl_rows [B]  (mvk_gauss_sample_kl_fwd/bwd; MoPoE style latents)."""

    @staticmethod
    def forward(ctx, eps, mu, lv):
        eps, mu, lv = _c(eps), _c(mu), _c(lv)
        K, B, L = eps.shape
        w = torch.empty_like(eps)
        kl = _new((B,), eps)
        call("mvk_gauss_sample_kl_fwd", ptr(mu), ptr(lv), ptr(eps), K, B, L, ptr(w), ptr(kl), stream_ptr())
        ctx.save_for_backward(eps, mu, lv)
        return w, kl

    @staticmethod
    @once_differentiable
    def backward(ctx, dw, dkl):
        wait_loss(ctx.saved_tensors[0].device)  # the gradients of the KL rows come out of the loss assembly launch
        late_ready(ctx.saved_tensors[0].device)
        eps, mu, lv = ctx.saved_tensors
        K, B, L = eps.shape
        dw = _c(dw) if dw is not None else None
        dkl = _c(dkl) if dkl is not None else None
        dmu, dlv = torch.empty_like(mu), torch.empty_like(lv)
        call("mvk_gauss_sample_kl_bwd", ptr(mu), ptr(lv), ptr(eps), ptr(dw), ptr(dkl), K, B, L, ptr(dmu), ptr(dlv),
             stream_ptr())
        return None, dmu, dlv


class MVAEPosteriorFn(Function):
    """(mu_m, lv_m)_m, eps [S,B,L] -> z of every subset written into its modalities' decoder inputs zm[m] [K_m,B,L],
    kld_rows [S,B] (mvk_mvae_posterior_fwd/bwd; mvae_model.py:56-118)."""

    @staticmethod
    def forward(ctx, eps, masks, subset_bits, want_stats, *mus_lvs):
        M = len(mus_lvs) // 2
        mus = [_c(t) for t in mus_lvs[:M]]
        lvs = [_c(t) for t in mus_lvs[M:]]
        eps = _c(eps)
        S, B, L = eps.shape
        bits = (C.c_int32 * S)(*subset_bits)
        counts = [sum((b >> m) & 1 for b in subset_bits) for m in range(M)]
        zm = [_new((counts[m], B, L), eps) if counts[m] else None for m in range(M)]
        kld = _new((S, B), eps)
        smu = _new((S, B, L), eps) if want_stats else None
        slv = _new((S, B, L), eps) if want_stats else None
        marr = ptr_array(masks) if masks is not None else None
        call("mvk_mvae_posterior_fwd", ptr_array(mus), ptr_array(lvs), marr, M, bits, S, ptr(eps), B, L, ptr_array(zm),
             ptr(kld), ptr(smu) if want_stats else None, ptr(slv) if want_stats else None, stream_ptr())
        ctx.save_for_backward(eps, *mus, *lvs)
        ctx.masks, ctx.bits, ctx.M, ctx.n_z = masks, bits, M, sum(1 for t in zm if t is not None)
        ctx.has_z = [t is not None for t in zm]
        outs = [t for t in zm if t is not None] + [kld]
        if want_stats:
            outs += [smu, slv]
            ctx.mark_non_differentiable(smu, slv)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        saved = ctx.saved_tensors
        eps, M = saved[0], ctx.M
        mus, lvs = list(saved[1:1 + M]), list(saved[1 + M:1 + 2 * M])
        S, B, L = eps.shape
        gz = iter(grads[:ctx.n_z])
        dzm = [(_c(next(gz)) if has else None) for has in ctx.has_z]
        gk = grads[ctx.n_z]
        gk = _c(gk) if gk is not None else None
        dmu = [torch.empty_like(t) for t in mus]
        dlv = [torch.empty_like(t) for t in lvs]
        marr = ptr_array(ctx.masks) if ctx.masks is not None else None
        call("mvk_mvae_posterior_bwd", ptr_array(mus), ptr_array(lvs), marr, M, ctx.bits, S, ptr(eps), ptr_array(dzm), B,
             L, ptr(gk) if gk is not None else None, ptr_array(dmu), ptr_array(dlv), stream_ptr())
        return (None, None, None, None, *dmu, *dlv)


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0, vmax=None,
              zero_grad=False):
    """torch.optim.Adam on flat buffers, one launch; vmax = the amsgrad running maximum of exp_avg_sq (or None);
    zero_grad: g is cleared as it is consumed."""
    call("mvk_adam_step_fused", ptr(p), ptr(g), ptr(m), ptr(v), ptr(vmax), p.numel(), lr, beta1, beta2, eps,
         weight_decay, step, grad_scale, 1 if zero_grad else 0, stream_ptr())


# =====================================================================================================
# MMVAE: mixture-of-experts importance weights (IWAE / DReG)
# =====================================================================================================
class MMVAEStdFn(Function):
    """std = exp(lv/2) | softmax(lv)*L + 1e-6   (mmvae_model.py:66-74)."""

    @staticmethod
    def forward(ctx, lv, family):
        lv2 = _c(lv.reshape(-1, lv.shape[-1]))
        sd = torch.empty_like(lv2)
        call("mvk_mmvae_std_fwd", ptr(lv2), lv2.shape[0], lv2.shape[1], family, ptr(sd), stream_ptr())
        ctx.save_for_backward(lv2, sd)
        ctx.family = family
        ctx.shape = lv.shape
        return sd.view(lv.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dsd):
        lv2, sd = ctx.saved_tensors
        dsd = _c(dsd).view(lv2.shape)
        dlv = torch.empty_like(lv2)
        call("mvk_mmvae_std_bwd", ptr(lv2), ptr(sd), ptr(dsd), lv2.shape[0], lv2.shape[1], ctx.family, ptr(dlv),
             stream_ptr())
        return dlv.view(ctx.shape), None


class MMVAEState:
    """Tensors shared between the latent node and the objective node of one forward pass."""

    def __init__(self):
        self.z = self.lpz = self.lqz = self.lq_all = None
        self.w = self.rowcoef = self.lw = None
        self.gloss = None
        self.lqw = None          # MMVAE+: log q_c(w_c) rows
        self.shared_dims = None  # MMVAE+: leading latent dims that enter the mixture (None = all)
        self.beta = 1.0          # MMVAE+: weight of the latent terms of lw


class MMVAELatentFn(Function):
    """(mu_c, std_c)_c, prior_std -> z_c [K,B,L] for every conditioning modality; also fills state.lpz / lqz."""

    @staticmethod
    def forward(ctx, state, noises, masks, prior_mean, family, dreg, prior_std, *mus_stds):
        M = len(mus_stds) // 2
        mus = [_c(t) for t in mus_stds[:M]]
        sds = [_c(t) for t in mus_stds[M:]]
        prior_std = _c(prior_std.reshape(-1))
        prior_mean = _c(prior_mean.reshape(-1))
        K, B, L = noises[0].shape
        zs = [_new((K, B, L), mus[0]) for _ in range(M)]
        state.lpz = [_new((K, B), mus[0]) for _ in range(M)]
        state.lqz = [_new((K, B), mus[0]) for _ in range(M)]
        state.lq_all = [_new((M, K, B), mus[0]) for _ in range(M)]
        Ls = L if state.shared_dims is None else int(state.shared_dims)
        state.lqw = [_new((K, B), mus[0]) for _ in range(M)] if Ls < L else None
        marr = ptr_array(masks) if masks is not None else None
        call("mvk_mmvae_latent_fwd", ptr_array(mus), ptr_array(sds), ptr_array(noises), marr, ptr(prior_mean),
             ptr(prior_std), M, K, B, L, family, ptr_array(zs), ptr_array(state.lpz), ptr_array(state.lqz),
             ptr_array(state.lq_all), Ls, ptr_array(state.lqw) if state.lqw is not None else None, stream_ptr())
        state.z = zs
        ctx.save_for_backward(prior_mean, prior_std, *mus, *sds)
        ctx.state, ctx.noises, ctx.masks = state, noises, masks
        ctx.family, ctx.dreg, ctx.M = family, dreg, M
        return tuple(zs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dzs):
        saved = ctx.saved_tensors
        prior_mean, prior_std = saved[0], saved[1]
        M = ctx.M
        mus, sds = saved[2 : 2 + M], saved[2 + M :]
        st = ctx.state
        K, B, L = ctx.noises[0].shape
        dzs = [(_c(d) if d is not None else torch.zeros_like(st.z[i])) for i, d in enumerate(dzs)]
        dmus = [_new((B, L), mus[0]) for _ in range(M)]
        dsds = [_new((B, L), mus[0]) for _ in range(M)]
        dprior_rows = _new((B, L), mus[0])  # per-row terms of d loss / d prior_std, summed below in a fixed order
        marr = ptr_array(ctx.masks) if ctx.masks is not None else None
        call("mvk_mmvae_latent_bwd", ptr_array(mus), ptr_array(sds), ptr_array(ctx.noises), ptr_array(st.z), marr,
             ptr(prior_mean), ptr(prior_std), ptr_array(st.w), ptr_array(st.lq_all), ptr_array(st.lqz),
             ptr_array(dzs), M, K, B, L, ctx.family, ctx.dreg, ptr(st.gloss), ptr_array(dmus), ptr_array(dsds),
             ptr(dprior_rows), L if st.shared_dims is None else int(st.shared_dims), float(st.beta), stream_ptr())
        dprior = _zeros((L,), mus[0])
        ws = _ws(dprior_rows)
        call("mvk_colsum_acc", ptr(dprior_rows), None, NONE, ptr(dprior), B, L, ptr(ws), ws.numel(), stream_ptr())
        return (None, None, None, None, None, None, dprior.view(1, L), *dmus, *dsds)


class MMVAEPlusCrossLatentFn(Function):
    """z_c [K,B,D] (= [u_c, w_c]), the target modality's private prior std [S], noise [K,B,S] -> [u_c, w ~ prior_r]
    (mmvaePlus_model.py:152-172)."""

    @staticmethod
    def forward(ctx, z, prior_std, noise, Ls, family):
        z = _c(z)
        prior_std = _c(prior_std.reshape(-1))
        noise = _c(noise)
        K, B, D = z.shape
        zc = torch.empty_like(z)
        call("mvk_mmvaeplus_cross_latent_fwd", ptr(z), ptr(prior_std), ptr(noise), K * B, D, Ls, family, ptr(zc),
             stream_ptr())
        ctx.save_for_backward(noise)
        ctx.dims = (K, B, D, Ls, family)
        ctx.prior_shape = None
        return zc

    @staticmethod
    @once_differentiable
    def backward(ctx, dzc):
        (noise,) = ctx.saved_tensors
        K, B, D, Ls, family = ctx.dims
        dzc = _c(dzc)
        dz = torch.empty_like(dzc)
        dprior = _new((1, D - Ls), dzc)
        call("mvk_mmvaeplus_cross_latent_bwd", ptr(dzc), ptr(noise), K * B, D, Ls, family, ptr(dz), ptr(dprior),
             stream_ptr())
        return dz, dprior, None, None, None


class MMVAEObjectiveFn(Function):
    """recon[c][r] (M*M tensors, c-major) -> loss.  forward: row NLLs + IWAE/DReG objective; backward: the
    second reconstruction pass with the per-row weights d loss / d lw."""

    @staticmethod
    def forward(ctx, state, spec, M, dreg, *recons):
        recons = [_c(t) for t in recons]
        K, B = spec["K"], spec["B"]
        ref = recons[0]
        descs = (ReconDesc * M)()
        rows = []
        for c in range(M):
            for r in range(M):
                d = descs[r]
                rr = _new((K, B), ref)
                rows.append(rr)
                t = recons[c * M + r]
                d.recon, d.x = t.data_ptr(), spec["x"][r].data_ptr()
                d.mask, d.rows, d.drecon, d.rowcoef = None, rr.data_ptr(), None, None
                d.D, d.dist, d.n_classes = t.numel() // (K * B), spec["dist"][r], t.shape[-1]
                d.scale, d.rescale, d.coef = spec["scale"][r], spec["rescale"][r], 1.0
            call("mvk_recon_nll_fwd", descs, M, K, B, stream_ptr())
        state.lw = [_new((K, B), ref) for _ in range(M)]
        state.w = [_new((K, B), ref) for _ in range(M)]
        state.rowcoef = [_new((K, B), ref) for _ in range(M)]
        loss = _new((), ref)
        masks = spec["masks"]
        marr = ptr_array(masks) if masks[0] is not None else None
        call("mvk_mmvae_objective_fwd", ptr_array(rows), ptr_array(state.lpz), ptr_array(state.lqz), marr, M, K, B,
             int(dreg), ptr_array(state.lw), ptr_array(state.w), ptr_array(state.rowcoef), ptr(loss),
             ptr_array(state.lqw) if state.lqw is not None else None, float(state.beta), stream_ptr())
        ctx.save_for_backward(*recons)
        ctx.state, ctx.spec, ctx.M = state, spec, M
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss):
        recons = ctx.saved_tensors
        st, spec, M = ctx.state, ctx.spec, ctx.M
        K, B = spec["K"], spec["B"]
        st.gloss = _c(gloss.reshape(1))
        grads = []
        descs = (ReconDesc * M)()
        for c in range(M):
            gs = []
            for r in range(M):
                d = descs[r]
                t = recons[c * M + r]
                g = torch.empty_like(t)
                gs.append(g)
                mk = spec["masks"][r]
                d.recon, d.x = t.data_ptr(), spec["x"][r].data_ptr()
                d.mask = mk.data_ptr() if mk is not None else None
                d.rows, d.drecon, d.rowcoef = None, g.data_ptr(), st.rowcoef[c].data_ptr()
                d.D, d.dist, d.n_classes = t.numel() // (K * B), spec["dist"][r], t.shape[-1]
                # lw contains +log p = -NLL, so d loss / d recon = rowcoef * (-1) * rescale * dNLL/drecon
                d.scale, d.rescale, d.coef = spec["scale"][r], spec["rescale"][r], -1.0
            call("mvk_recon_nll_bwd", descs, M, K, B, stream_ptr())
            for g in gs:
                call("mvk_scale_by_device_scalar", ptr(g), g.numel(), ptr(st.gloss), stream_ptr())
            grads += gs
        return (None, None, None, None, *grads)


# =====================================================================================================
# Importance-sampled joint likelihood (compute_joint_nll)
# =====================================================================================================
IWAE_ROWS_BUDGET = 1 << 16  # decoder rows (K * data points) per pass of joint_nll


def std_from_logvar(lv, family=FAMILY["normal"]):
    """exp(lv/2) (or the MMVAE scale parametrisations) without an autograd node."""
    lv2 = _c(lv.reshape(-1, lv.shape[-1]))
    sd = torch.empty_like(lv2)
    call("mvk_mmvae_std_fwd", ptr(lv2), lv2.shape[0], lv2.shape[1], family, ptr(sd), stream_ptr())
    return sd.view(lv.shape)


def iwae_sample(loc, sd, noise, family=FAMILY["normal"]):
    """z [K,B,L] = loc + sd * t(noise)."""
    loc, sd, noise = _c(loc), _c(sd), _c(noise)
    K, B, L = noise.shape
    z = torch.empty_like(noise)
    call("mvk_iwae_sample", ptr(loc), ptr(sd), ptr(noise), K, B, L, family, ptr(z), stream_ptr())
    return z


def recon_nll_rows(recons, xs, dists, scales, K, B):
    """Unrescaled, unmasked NLL rows [K,B] of every modality in one mvk_recon_nll_fwd launch (no gradient)."""
    n = len(recons)
    recons = [_c(r) for r in recons]
    descs = (ReconDesc * n)()
    rows = []
    for i in range(n):
        r = _new((K, B), recons[i])
        rows.append(r)
        d = descs[i]
        d.recon, d.x, d.mask, d.rows, d.drecon, d.rowcoef = recons[i].data_ptr(), xs[i].data_ptr(), None, r.data_ptr(), None, None
        d.D, d.dist, d.scale, d.rescale, d.coef = recons[i].numel() // (K * B), dists[i], scales[i], 1.0, 1.0
        d.n_classes = recons[i].shape[-1]
    call("mvk_recon_nll_fwd", descs, n, K, B, stream_ptr())
    return rows


def iwae_logw(z, rows, locs, sds, family=FAMILY["normal"], prior_loc=None, prior_sd=None):
    """lw [K,B] = -sum rows + log p(z) - log mean_e q_e(z)   (mvk_iwae_logw)."""
    z = _c(z)
    K, B, L = z.shape
    locs, sds = [_c(t) for t in locs], [_c(t) for t in sds]
    lw = _new((K, B), z)
    pl = _c(prior_loc.reshape(-1)) if prior_loc is not None else None
    ps = _c(prior_sd.reshape(-1)) if prior_sd is not None else None
    call("mvk_iwae_logw", ptr(z), ptr_array(rows) if rows else None, len(rows), ptr_array(locs), ptr_array(sds),
         len(locs), ptr(pl) if pl is not None else None, ptr(ps) if ps is not None else None, K, B, L, family,
         ptr(lw), stream_ptr())
    return lw


def iwae_reduce(lws, out=None):
    """ll [B] = log-mean-exp over the K rows of every array in lws (mvk_iwae_reduce)."""
    lws = [_c(t) for t in lws]
    K, B = lws[0].shape
    ll = _new((B,), lws[0]) if out is None else out
    call("mvk_iwae_reduce", ptr_array(lws), len(lws), K, B, ptr(ll), stream_ptr())
    return ll


def joint_nll(decode_rows, z, locs, sds, family=FAMILY["normal"], prior_loc=None, prior_sd=None):
    """-sum_b log-mean-exp_k [ log p(x_b | z_kb) + log p(z_kb) - log mean_e q_e(z_kb | x_b) ].

    decode_rows(z_chunk [K,b,L], b0, b1) -> list of [K,b] NLL rows (decoders + recon_nll_rows on data rows b0:b1).
    The data axis is processed in chunks of IWAE_ROWS_BUDGET // K points so that the decoder activations of
    K = 1000 samples stay bounded; every chunk is independent (the estimate is per data point).
    """
    K, B, L = z.shape
    ll = _new((B,), z)
    step = max(1, IWAE_ROWS_BUDGET // max(K, 1))
    for b0 in range(0, B, step):
        b1 = min(B, b0 + step)
        zc = z[:, b0:b1].contiguous()
        rows = decode_rows(zc, b0, b1)
        lw = iwae_logw(zc, rows, [t[b0:b1] for t in locs], [t[b0:b1] for t in sds], family, prior_loc, prior_sd)
        iwae_reduce([lw], out=ll[b0:b1])
    return -ll.sum()
