"""Oracle: one full training step (forward + ELBO + backward + Adam) on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates
  /root/reference/src/multivae/trainers/base/base_trainer.py:350-361 (_optimizers_step: zero_grad, backward, step)
  /root/reference/src/multivae/trainers/base/base_trainer.py:682-750 (train_step body for one batch)
  torch.optim.Adam (the reference's default optimizer_cls, base_trainer_config.py:58) single-tensor rule.
Used by tests (parity of a whole step) and by bench.py's `cpu_baseline` leg (timed on host cores).
"""
import math

import torch

from . import elbo, nets


def adam_update(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """In-place Adam update of p with state (m, v) at 1-based `step` (torch.optim.Adam, amsgrad=False)."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m.lerp_(g, 1 - beta1)  # torch.optim.Adam: exp_avg.lerp_(grad, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1**step
    bc2 = 1 - beta2**step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


class AdamState:
    def __init__(self, sd):
        self.m = {k: torch.zeros_like(v) for k, v in sd.items()}
        self.v = {k: torch.zeros_like(v) for k, v in sd.items()}
        self.step = 0


def mopoe_mnist_svhn_loss(sd, data, eps, *, beta=1.0, uses_likelihood_rescaling=False, masks=None, choice=None):
    """MoPoE on the MnistSvhn architecture; eps [B,L] (reference) or [K,B,L] (K-sample extension)."""
    names = ["mnist", "svhn"]
    enc, dec = nets.build_mnist_svhn(sd)
    rescale = elbo.rescale_factors({"mnist": (1, 28, 28), "svhn": (3, 32, 32)}, uses_likelihood_rescaling)
    e = {m: enc[m](data[m]) for m in names}
    return elbo.mopoe_forward(e, data, dec, eps, names=names, beta=beta, rescale=rescale, masks=masks, choice=choice)


def train_step(sd, state, loss_fn, lr=1e-3, grad_scale=1.0):
    """One optimizer step on the leaf tensors of `sd` (must require grad).  Returns the loss dict.

    grad_scale emulates DDP's gradient averaging (1/world_size applied after summing shard grads).
    """
    for p in sd.values():
        p.grad = None
    out = loss_fn(sd)
    out["loss"].backward()
    state.step += 1
    with torch.no_grad():
        for k, p in sd.items():
            if p.grad is None:
                continue
            adam_update(p, p.grad * grad_scale, state.m[k], state.v[k], state.step, lr=lr)
    return out
