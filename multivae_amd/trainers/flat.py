"""Flat fp32 parameter / gradient buffers and the fused HIP Adam.

All trainable parameters of a model become views into ONE contiguous buffer (and their .grad views into
another) so that (a) the optimizer is one kernel launch over the whole model (mvk_adam_step), and (b) the
data-parallel gradient exchange is ONE RCCL all-reduce over xGMI per step — the
MI355X-native equivalent of the reference's DDP bucket reducer (`base_trainer.py:116-117`).
"""
import torch

from .. import kernels


class FlatParams:
    def __init__(self, model: torch.nn.Module):
        self.all_params = list(model.parameters())  # torch.optim indexes its state by position in this list
        self.params = [p for p in self.all_params if p.requires_grad]
        if not self.params:
            raise ValueError("model has no trainable parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off : off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off : off + k].view(p.shape)
            p.grad = self.grad[off : off + k].view(p.shape)
            off += k
        self.numel = n

    def zero_grad(self):
        self.grad.zero_()
        for p in self.params:  # re-attach views if something replaced .grad
            if p.grad is None or p.grad.data_ptr() < self.grad.data_ptr() or \
                    p.grad.data_ptr() >= self.grad.data_ptr() + 4 * self.numel:
                self._reattach()
                break

    def _reattach(self):
        off = 0
        for p in self.params:
            k = p.numel()
            view = self.grad[off : off + k].view(p.shape)
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                view.add_(p.grad)
            p.grad = view
            off += k

    def all_reduce(self, group=None):
        """ONE collective over the whole gradient buffer (sum); the 1/world_size is folded into Adam."""
        import torch.distributed as dist

        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)

    def broadcast(self, src=0, group=None):
        import torch.distributed as dist

        dist.broadcast(self.flat, src=src, group=group)


class FusedAdam:
    """torch.optim.Adam (amsgrad=False) semantics on FlatParams, one HIP launch per step."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.step_count = 0

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self, grad_scale=1.0):
        self.step_count += 1
        kernels.adam_step(self.flat.flat, self.flat.grad, self.m, self.v, self.step_count, self.lr, self.betas[0],
                          self.betas[1], self.eps, self.weight_decay, grad_scale)

    def state_dict(self):
        """The layout of `torch.optim.Adam(model.parameters()).state_dict()` (what the reference writes to
        `optimizer.pt`, base_trainer.py:790-793, and reads back in `resume_training`, :413-419): per-parameter
        `step` / `exp_avg` / `exp_avg_sq` keyed by the parameter's position in `model.parameters()`, one param group."""
        pos = {id(p): i for i, p in enumerate(self.flat.all_params)}
        state, off = {}, 0
        for p in self.flat.params:
            k = p.numel()
            if self.step_count > 0:  # torch creates the state lazily at the first step
                state[pos[id(p)]] = {"step": torch.tensor(float(self.step_count)),
                                     "exp_avg": self.m[off:off + k].view(p.shape).clone(),
                                     "exp_avg_sq": self.v[off:off + k].view(p.shape).clone()}
            off += k
        group = dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.weight_decay, amsgrad=False,
                     maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     decoupled_weight_decay=False, params=list(range(len(self.flat.all_params))))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts the torch.optim.Adam layout above (also an `optimizer.pt` written by the reference)."""
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.flat.all_params):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        if groups[0].get("amsgrad", False):
            raise ValueError("the fused Adam has no amsgrad state: use use_fused_adam=False")
        g = groups[0]
        self.lr, self.betas = float(g["lr"]), tuple(g["betas"])
        self.eps, self.weight_decay = float(g["eps"]), float(g["weight_decay"])
        pos = {id(p): i for i, p in enumerate(self.flat.all_params)}
        self.m.zero_()
        self.v.zero_()
        steps, off = set(), 0
        for p in self.flat.params:
            k = p.numel()
            st = sd["state"].get(pos[id(p)])
            if st is not None:
                self.m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            off += k
        if len(steps) > 1:
            raise ValueError(f"parameters with different step counts {sorted(steps)}: not representable by one fused step")
        self.step_count = steps.pop() if steps else 0
