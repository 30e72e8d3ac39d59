"""Flat fp32 parameter / gradient buffers and the fused HIP Adam.

All trainable parameters of a model become views into ONE contiguous buffer (and their .grad views into
another) so that (a) the optimizer is one kernel launch over the whole model (mvk_adam_step), and (b) the
data-parallel gradient exchange is ONE RCCL all-reduce over xGMI per step — the
MI355X-native equivalent of the reference's DDP bucket reducer (`base_trainer.py:116-117`).
"""
import torch

from .. import _lib, kernels


class FlatParams:
    ALIGN = 64  # floats

    def __init__(self, model: torch.nn.Module):
        self.all_params = list(model.parameters())  # torch.optim indexes its state by position in this list
        self.params = [p for p in self.all_params if p.requires_grad]
        if not self.params:
            raise ValueError("model has no trainable parameters")
        # Parameters a module announces as `late_leaf_params()` (the decoders' weights whose gradients a rotated step produces
        # at the head of the NEXT step: kernels.Rotation) sit together at the END of the buffers, so that the step's two
        # optimizer launches (everything else / the rotated range) and the two halves of a data-parallel collective are one
        # contiguous range each.  The layout inside the buffer is private: state dicts and `dense()` go by parameter.
        late_first, late_rest = [], []  # per announcing module: its FIRST entry (by convention the first layer's weight, the
        # one a rotation of the small leaves only moves) / the others; the first entries sit at the very end, together
        for mod in model.modules():
            announce = getattr(mod, "late_leaf_params", None)
            if announce is not None:
                ps = [p for p in announce() if p.requires_grad]
                late_first += ps[:1]
                late_rest += ps[1:]
        group = {id(p): 1 for p in late_rest}
        group.update({id(p): 2 for p in late_first})
        late_ids = set(group)
        dev = self.params[0].device
        # Every parameter starts on a 256-byte boundary of the buffer.  Densely packed, ONE 3-element bias (the image layer
        # of the SVHN decoder) leaves every parameter behind it 12 bytes off a 16-byte boundary, and the kernels' 16-byte
        # weight loads (`mvk_aligned16` checks in the C layer) silently fall back to their scalar paths for all of them
        # (in `model.parameters()` order that was every encoder of the MnistSvhn models).  The padding stays zero: zero
        # gradient, zero Adam update; the all-reduce carries it along (< 0.1 % of the buffer).
        # (`params` / `offsets` stay in model.parameters() order; only the offsets of the late-leaf parameters are at the end)
        where, off = {}, 0
        for g in (0, 1, 2):
            for p in self.params:
                if group.get(id(p), 0) == g:
                    where[id(p)] = off
                    off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.offsets = [where[id(p)] for p in self.params]
        self.late_start = min([where[i] for i in late_ids], default=off)  # first float of the late-leaf parameters
        n = off
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            self.flat[off : off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off : off + k].view(p.shape)
            p.grad = self.grad[off : off + k].view(p.shape)
        self.numel = n
        # set by FusedAdam(zero_grad_in_step=True) after a step that cleared the buffer while consuming it; the next
        # zero_grad() then costs nothing.  Whoever writes gradients in between WITHOUT calling zero_grad() first
        # (nobody in BaseTrainer / bench.py) must reset it.
        self.grads_zero = False

    def zero_grad(self):
        if self.grads_zero:
            self.grads_zero = False  # a backward pass follows
        else:
            self.grad.zero_()
        for p in self.params:  # re-attach views if something replaced .grad
            if p.grad is None or p.grad.data_ptr() < self.grad.data_ptr() or \
                    p.grad.data_ptr() >= self.grad.data_ptr() + 4 * self.numel:
                self._reattach()
                break

    def ensure_attached(self):
        """After a backward pass: fold any gradient that landed outside the flat buffer back into it."""
        lo, hi = self.grad.data_ptr(), self.grad.data_ptr() + 4 * self.numel
        for p in self.params:
            if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                self._reattach()
                return

    def _reattach(self):
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            view = self.grad[off : off + k].view(p.shape)
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                view.add_(p.grad)
            p.grad = view

    def dense(self, buf):
        """The parameter slices of a buffer laid out like `flat` / `grad`, concatenated without the alignment padding
        (the order of `model.parameters()`)."""
        return torch.cat([buf[off : off + p.numel()] for p, off in zip(self.params, self.offsets)])

    def all_reduce(self, group=None):
        """ONE collective over the whole gradient buffer (sum); the 1/world_size is folded into Adam."""
        import torch.distributed as dist

        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)

    SEG_FLOATS = 1 << 19  # ~2 MB segments inside one RCCL group: 6.2 MB (MnistSvhn) = 3 segments, 96.6 MB (cfg5) = 46

    def all_reduce_mean(self, group=None):
        """ONE collective over the whole gradient buffer, the mean over the ranks — what the reference's DDP wrapper leaves in
        `.grad` (trainers/base/base_trainer.py:92-117).  Returns the factor the optimizer still has to apply (`grad_scale`):
        1.0 when the collective averaged (the C-ABI entry point mvk_allreduce_avg: RCCL over xGMI on the current stream, used
        whenever the default process group's backend is nccl), 1 / world_size after a plain sum (gloo: the CPU tests and the
        two-ranks-on-one-GPU test, where RCCL refuses a duplicate device).  MVK_RCCL=0 keeps torch.distributed's all_reduce."""
        import torch.distributed as dist

        world = dist.get_world_size(group)
        comm = self._rccl_comm(group)
        if comm is None:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
            return 1.0 / world
        # one call up to 8 MB (the MnistSvhn models' 6.2 MB): with one rank every segment is a launch of its own (RCCL's
        # oneRankReduce, 5 us each: profiles/r06_dp1_timeline.txt), and a ring over xGMI pipelines a call of this size internally
        nseg = 1 if self.grad.numel() <= 4 * self.SEG_FLOATS else max(1, min(64, (self.grad.numel() + self.SEG_FLOATS - 1) // self.SEG_FLOATS))
        _lib.call("mvk_allreduce_avg", _lib.ptr(self.grad), self.grad.numel(), nseg, comm, _lib.stream_ptr())
        return 1.0

    def all_reduce_mean_ranges(self, ranges, group=None):
        """The same mean over the ranks for a list of disjoint (offset, count) ranges of the gradient buffer, on the CURRENT
        stream (the overlapped step calls it twice: the ranges that are final early on the communication stream, the rest
        behind the backward pass).  Same return value as all_reduce_mean; every rank must call it with the same ranges."""
        import ctypes as C

        import torch.distributed as dist

        world = dist.get_world_size(group)
        ranges = [(int(o), int(n)) for o, n in ranges if n > 0]
        comm = self._rccl_comm(group)
        if comm is None:
            for o, n in ranges:
                dist.all_reduce(self.grad[o:o + n], op=dist.ReduceOp.SUM, group=group)
            return 1.0 / world
        off = (C.c_int64 * len(ranges))(*[o for o, _ in ranges])
        cnt = (C.c_int64 * len(ranges))(*[n for _, n in ranges])
        _lib.call("mvk_allreduce_avg_ranges", _lib.ptr(self.grad), off, cnt, len(ranges), self.SEG_FLOATS, comm, _lib.stream_ptr())
        return 1.0

    def complement(self, ranges):
        """The part of [0, numel) outside the given (offset, count) ranges, as sorted disjoint ranges."""
        out, pos = [], 0
        for o, n in sorted((int(o), int(n)) for o, n in ranges):
            if o > pos:
                out.append((pos, o - pos))
            pos = max(pos, o + n)
        if pos < self.numel:
            out.append((pos, self.numel - pos))
        return out

    def ranges_of(self, params, strict=False):
        """(offset, padded count) of the given parameters inside the flat buffers, merged where adjacent (a module's parameters
        are neighbours: one range).  Parameters that are not views of the buffer are ignored — or, with strict=True, make the
        result None (a LATE parameter that cannot be placed must not be classified as early by `complement`; the
        caller then falls back to one collective over the whole buffer)."""
        pos = {id(p): (off, (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN) for p, off in zip(self.params, self.offsets)}
        by_ptr = {p.data_ptr(): pos[id(p)] for p in self.params}
        got = set()
        for p in params:
            e = pos.get(id(p)) or by_ptr.get(p.data_ptr())  # a detached / re-wrapped view of the same storage still counts
            if e is None:
                if strict:
                    return None
                continue
            got.add(e)
        got = sorted(got)
        out = []
        for o, n in got:
            if out and out[-1][0] + out[-1][1] == o:
                out[-1] = (out[-1][0], out[-1][1] + n)
            else:
                out.append((o, n))
        return out

    def world_size(self, group=None):
        """The number of ranks the gradient collective runs over, as the COMMUNICATOR reports it (ncclCommCount) when the RCCL
        path is in use, else the process group's size; 1 without a process group."""
        import ctypes as C

        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return 1
        comm = self._rccl_comm(group)
        if comm is None:
            return dist.get_world_size(group)
        w = C.c_int(0)
        _lib.call("mvk_comm_size", comm, C.byref(w), None)
        return int(w.value)

    def _rccl_comm(self, group=None):
        """The RCCL communicator behind mvk_allreduce_avg (created on first use: rank 0 draws the rendezvous id, the default
        process group carries it to the other ranks), or None when the process group is not RCCL-backed.

        Collective-safe: nothing here raises on ONE rank while the others wait.  Every rank first reports whether
        RCCL resolves in its process (mvk_comm_available, no collective), rank 0 additionally whether it could draw the id; the
        ranks agree (all_reduce MIN over the process group) before anyone enters ncclCommInitRank, agree again on its outcome,
        and fall back to torch.distributed's all_reduce TOGETHER if any of them failed.  `_comm_checked` is set only behind
        that agreement; the communicator is destroyed by close() (BaseTrainer at the end of train(), bench.py at exit)."""
        import os

        import torch.distributed as dist

        if group is not None or not self.grad.is_cuda or os.environ.get("MVK_RCCL", "1") == "0":
            return None
        if getattr(self, "_comm_checked", False):
            return self._comm
        if dist.get_backend() != "nccl":
            self._comm_checked, self._comm = True, None
            return None
        import ctypes as C
        import warnings

        lib = _lib.load()
        ok = bool(lib.mvk_comm_available())
        nbytes = lib.mvk_comm_id_bytes()
        uid = (C.c_ubyte * nbytes)()
        if ok and dist.get_rank() == 0:
            ok = lib.mvk_comm_unique_id(uid) == _lib.MVK_OK
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0)  # carried unconditionally: every rank takes part whatever its own `ok`

        def agreed(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.grad.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        comm = C.c_void_p()
        if agreed(ok):
            uid = (C.c_ubyte * nbytes).from_buffer_copy(box[0])
            with torch.cuda.device(self.grad.device):
                ok = lib.mvk_comm_init(C.byref(comm), dist.get_world_size(), dist.get_rank(), uid) == _lib.MVK_OK
            w = C.c_int(0)
            ok = ok and lib.mvk_comm_size(comm, C.byref(w), None) == _lib.MVK_OK and w.value == dist.get_world_size()
            if not agreed(ok):
                if comm:
                    lib.mvk_comm_destroy(comm)
                comm = C.c_void_p()
        if not comm:
            warnings.warn("RCCL communicator for mvk_allreduce_avg could not be created on every rank: the gradient "
                          "collective falls back to torch.distributed.all_reduce on all ranks", RuntimeWarning)
            self._comm_checked, self._comm = True, None
            return None
        self._comm_checked, self._comm = True, comm
        return comm

    def close(self):
        """Destroy the RCCL communicator (the reference tears its process group down at the end of train(),
        trainers/base/base_trainer.py:613-614).  Idempotent; the next collective would create a new one."""
        comm = getattr(self, "_comm", None)
        self._comm, self._comm_checked = None, False
        if comm:
            if self.grad.is_cuda:
                torch.cuda.synchronize(self.grad.device)
            _lib.load().mvk_comm_destroy(comm)

    def __del__(self):
        try:
            if getattr(self, "_comm", None):
                _lib.load().mvk_comm_destroy(self._comm)
                self._comm = None
        except Exception:
            pass

    def broadcast(self, src=0, group=None):
        import torch.distributed as dist

        dist.broadcast(self.flat, src=src, group=group)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (amsgrad on or off) on FlatParams, one HIP launch per step (mvk_adam_step_amsgrad).

    A real `torch.optim.Optimizer` with ONE parameter group, so `torch.optim.lr_scheduler.*` drive it unchanged (the
    learning rate is a host scalar handed to the kernel at every step) and `state_dict()` / `load_state_dict()` speak
    the layout of `torch.optim.Adam(model.parameters())` — what the reference writes to / reads from `optimizer.pt`."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False,
                 zero_grad_in_step=False):
        """zero_grad_in_step: the update clears the gradient buffer as it reads it, and the `zero_grad()` of the next step
        becomes free (one launch and one pass over the buffer less per step).  For loops of the form zero_grad -> backward
        -> step (BaseTrainer, bench.py): a gradient written after `step()` and before the next `zero_grad()` would survive."""
        self.flat = flat
        self.zero_grad_in_step = bool(zero_grad_in_step)
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=bool(amsgrad))
        super().__init__(flat.all_params, defaults)
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.vmax = torch.zeros_like(flat.flat) if amsgrad else None
        self.step_count = 0
        self._dev_state = self._dev_scalars = self._dev_host = None  # the captured form (prepare_captured / step_captured)
        # the rotated form (trainers/graph.py GraphedStep(rotate=True)): ranges updated at the head of the NEXT replay
        self._rot_ranges = None
        self._rot_main = None
        self._rot_scalars = None
        self._rot_dirty = False

    # the hyper-parameters live in the parameter group (schedulers write `lr` there)
    def _g(self):
        return self.param_groups[0]

    lr = property(lambda self: self._g()["lr"], lambda self, v: self._g().__setitem__("lr", v))
    betas = property(lambda self: tuple(self._g()["betas"]), lambda self, v: self._g().__setitem__("betas", tuple(v)))
    eps = property(lambda self: self._g()["eps"], lambda self, v: self._g().__setitem__("eps", v))
    weight_decay = property(lambda self: self._g()["weight_decay"],
                            lambda self, v: self._g().__setitem__("weight_decay", v))
    amsgrad = property(lambda self: bool(self._g().get("amsgrad", False)))

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()  # gradients stay views of the flat buffer (never None)

    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        if self._rot_dirty and not self._rot_armed:
            raise RuntimeError("FusedAdam.step(): the rotated ranges of the previous replayed step are not updated yet — call "
                               "GraphedStep.drain() before an eager step")
        self.step_count += 1
        g = self._g()
        if self._rot_ranges is not None and self._rot_armed:
            # behind a rotated replay: everything but the rotated ranges now, with the scalars left on the device for the update
            # of the rotated ranges inside the next replay (or the drain)
            f = self.flat
            ranges = self._rot_main or [(0, 0)]
            for i, (o, n) in enumerate(ranges):
                _lib.call("mvk_adam_step_pub", _lib.ptr(f.flat[o:o + n]) if n else None, _lib.ptr(f.grad[o:o + n]) if n else None,
                          _lib.ptr(self.m[o:o + n]) if n else None, _lib.ptr(self.v[o:o + n]) if n else None,
                          _lib.ptr(self.vmax[o:o + n]) if (self.vmax is not None and n) else None, n, float(g["lr"]),
                          float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self.step_count,
                          float(grad_scale), 1 if self.zero_grad_in_step else 0,
                          _lib.ptr(self._rot_scalars) if i == 0 else None, _lib.stream_ptr())
            self._rot_dirty, self._rot_armed = True, False
            self.flat.grads_zero = self.zero_grad_in_step
            return loss
        kernels.adam_step(self.flat.flat, self.flat.grad, self.m, self.v, self.step_count, float(g["lr"]),
                          float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                          grad_scale, vmax=self.vmax, zero_grad=self.zero_grad_in_step)
        self.flat.grads_zero = self.zero_grad_in_step
        return loss

    # -- the rotated step (trainers/graph.py): the optimizer in two launches over disjoint ranges, same scalars ------------
    _rot_armed = False

    def set_rotation(self, ranges):
        """ranges: (offset, count) of the parameters a rotated GraphedStep updates at the head of its NEXT replay (None: off).
        step() behind a replay of that graph (`arm_rotation`) then covers the rest and publishes its scalars."""
        if self._rot_dirty:
            raise RuntimeError("FusedAdam.set_rotation(): drain the rotated step first")
        if not ranges:
            self._rot_ranges = self._rot_main = None
            return
        self._rot_ranges = [(int(o), int(n)) for o, n in ranges]
        self._rot_main = self.flat.complement(self._rot_ranges)
        if self._rot_scalars is None:
            self._rot_scalars = torch.zeros(8, dtype=torch.float32, device=self.flat.flat.device)
            _lib.call("mvk_adam_identity", _lib.ptr(self._rot_scalars), _lib.stream_ptr())

    def arm_rotation(self):
        """Called by the rotated GraphedStep behind every replay: the next step() is the main part of a rotated update."""
        self._rot_armed = True

    def rot_update(self):
        """mvk_adam_step_dev over the rotated ranges with the published scalars, on the current stream (captured at the head of
        the rotated graph behind the late gradients; eager in the drain).  Identity scalars (before the first step, after a
        drain) change nothing."""
        f = self.flat
        for o, n in self._rot_ranges:
            _lib.call("mvk_adam_step_dev", _lib.ptr(f.flat[o:o + n]), _lib.ptr(f.grad[o:o + n]), _lib.ptr(self.m[o:o + n]),
                      _lib.ptr(self.v[o:o + n]), _lib.ptr(self.vmax[o:o + n]) if self.vmax is not None else None, n,
                      _lib.ptr(self._rot_scalars), 1 if self.zero_grad_in_step else 0, _lib.stream_ptr())

    def rot_drained(self):
        """Behind an eager rot_update (the drain): the scalars become the identity again, every parameter has seen step_count updates."""
        _lib.call("mvk_adam_identity", _lib.ptr(self._rot_scalars), _lib.stream_ptr())
        self._rot_dirty = False

    # -- the optimizer inside a replayed hipGraph (trainers/graph.py): scalars in device memory -------------------------
    def sync_device_state(self, grad_scale=1.0):
        """Before a replay: the device copy of (lr, betas, eps, weight_decay, grad_scale, step) is brought up to date — one
        small copy when a scheduler changed the learning rate, an eager step() ran in between, or a state_dict was loaded;
        nothing otherwise (the captured mvk_adam_prepare advances the step itself)."""
        g = self._g()
        want = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                float(grad_scale), float(self.step_count))
        if self._dev_state is None:
            dev = self.flat.flat.device
            self._dev_state = torch.zeros(8, dtype=torch.float64, device=dev)
            self._dev_scalars = torch.zeros(8, dtype=torch.float32, device=dev)
        if want != self._dev_host:
            self._dev_state.copy_(torch.tensor(list(want) + [0.0], dtype=torch.float64))
            self._dev_host = want

    def prepare_captured(self):
        """mvk_adam_prepare on the current stream (inside a capture): step += 1, the update's scalars derived on the device."""
        _lib.call("mvk_adam_prepare", _lib.ptr(self._dev_state), _lib.ptr(self._dev_scalars), _lib.stream_ptr())

    def step_captured(self):
        """mvk_adam_step_dev on the current stream (inside a capture, behind prepare_captured and the finished gradients)."""
        f = self.flat
        _lib.call("mvk_adam_step_dev", _lib.ptr(f.flat), _lib.ptr(f.grad), _lib.ptr(self.m), _lib.ptr(self.v), _lib.ptr(self.vmax),
                  f.numel, _lib.ptr(self._dev_scalars), 1 if self.zero_grad_in_step else 0, _lib.stream_ptr())

    def note_replayed_step(self):
        """The host's mirror of what a replay of the captured pair did."""
        self.step_count += 1
        self._opt_called = True  # what torch's lr schedulers look for ("scheduler.step() before optimizer.step()")
        if self._dev_host is not None:
            self._dev_host = self._dev_host[:6] + (float(self.step_count),)
        self.flat.grads_zero = self.zero_grad_in_step

    def state_dict(self):
        """The layout of `torch.optim.Adam(model.parameters()).state_dict()` (what the reference writes to
        `optimizer.pt`, base_trainer.py:790-793, and reads back in `resume_training`, :413-419): per-parameter
        `step` / `exp_avg` / `exp_avg_sq` (/ `max_exp_avg_sq`) keyed by the parameter's position in
        `model.parameters()`, one param group."""
        if self._rot_dirty:
            raise RuntimeError("FusedAdam.state_dict(): the rotated ranges lag one update behind — call GraphedStep.drain() first")
        pos = {id(p): i for i, p in enumerate(self.flat.all_params)}
        state = {}
        for p, off in zip(self.flat.params, self.flat.offsets):
            k = p.numel()
            if self.step_count > 0:  # torch creates the state lazily at the first step
                st = {"step": torch.tensor(float(self.step_count)),
                      "exp_avg": self.m[off:off + k].view(p.shape).clone(),
                      "exp_avg_sq": self.v[off:off + k].view(p.shape).clone()}
                if self.vmax is not None:
                    st["max_exp_avg_sq"] = self.vmax[off:off + k].view(p.shape).clone()
                state[pos[id(p)]] = st
        g = self._g()
        group = dict(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"],
                     amsgrad=self.amsgrad, maximize=False, foreach=None, capturable=False, differentiable=False,
                     fused=None, decoupled_weight_decay=False, params=list(range(len(self.flat.all_params))))
        if "initial_lr" in g:  # written by lr schedulers
            group["initial_lr"] = g["initial_lr"]
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts the torch.optim.Adam layout above (also an `optimizer.pt` written by the reference)."""
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.flat.all_params):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        g = groups[0]
        if bool(g.get("amsgrad", False)) and self.vmax is None:
            self.vmax = torch.zeros_like(self.flat.flat)
        if not g.get("amsgrad", False):
            self.vmax = None
        mine = self._g()
        mine["lr"], mine["betas"] = float(g["lr"]), tuple(g["betas"])
        mine["eps"], mine["weight_decay"] = float(g["eps"]), float(g["weight_decay"])
        mine["amsgrad"] = bool(g.get("amsgrad", False))
        if "initial_lr" in g:
            mine["initial_lr"] = g["initial_lr"]
        pos = {id(p): i for i, p in enumerate(self.flat.all_params)}
        self.m.zero_()
        self.v.zero_()
        if self.vmax is not None:
            self.vmax.zero_()
        steps = set()
        for p, off in zip(self.flat.params, self.flat.offsets):
            k = p.numel()
            st = sd["state"].get(pos[id(p)])
            if st is not None:
                self.m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                if self.vmax is not None and "max_exp_avg_sq" in st:
                    self.vmax[off:off + k].copy_(st["max_exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"parameters with different step counts {sorted(steps)}: not representable by one fused step")
        self.step_count = steps.pop() if steps else 0
