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
        self.params = [p for p in model.parameters() if p.requires_grad]
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
        return {"m": self.m, "v": self.v, "step": self.step_count, "lr": self.lr, "betas": self.betas,
                "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
