"""`multivae.models.base.base_utils` on the HIP path: ModelOutput and the public helper functions a user's encoder /
model may import — `poe`, `stable_poe`, `kl_divergence`, `rsample_from_gaussian`, `set_decoder_dist`, `cross_entropy`
(`/root/reference/src/multivae/models/base/base_utils.py:28-172`), same names, arguments and results.

Each helper is a torch.autograd.Function over one HIP kernel of libmvk.so (`csrc/utils.hip`, `csrc/elbo.hip`); torch only
provides device memory, broadcasting views and the random draw.  The training path of the built-in models uses the fused
kernels of `csrc/elbo.hip` instead (one launch for PoE + sampling + KL, one for all reconstruction terms).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ... import _lib
from ..._lib import call, ptr, stream_ptr
from ..._output import ModelOutput  # noqa: F401  (re-exported: `multivae.models.base.base_utils.ModelOutput`)


def decoder_dist_code(dist_name):
    """'normal' | 'laplace' | 'bernoulli' | 'categorical' -> MVK_DIST_*; anything else is rejected like the reference
    (`set_decoder_dist` raises ValueError, base_utils.py:84-85).  'categorical' takes tensors: logits [..., n_classes]
    against one-hot (or probability) targets of the same trailing shape."""
    if dist_name in _lib.DIST:
        return _lib.DIST[dist_name]
    raise ValueError("The distribution type 'dist' is not supported")


def _f32(t):
    _lib.require_gpu_tensor(t)
    return t if t.is_contiguous() else t.contiguous()


def _experts(x):
    """list / tuple of [*] tensors or one [E, *] tensor -> contiguous [E, *]."""
    if isinstance(x, (list, tuple)):
        x = torch.stack(list(x))
    return _f32(x)


class _PoEFn(Function):
    @staticmethod
    def forward(ctx, mus, lvs, eps, stable):
        mus, lvs = _f32(mus), _f32(lvs)
        E = mus.shape[0]
        n = mus[0].numel()
        mu, lv = torch.empty_like(mus[0]), torch.empty_like(mus[0])
        call("mvk_poe_fwd", ptr(mus), ptr(lvs), E, n, float(eps), int(stable), ptr(mu), ptr(lv), stream_ptr())
        ctx.save_for_backward(mus, lvs)
        ctx.cfg = (float(eps), int(stable))
        return mu, lv

    @staticmethod
    @once_differentiable
    def backward(ctx, gmu, glv):
        mus, lvs = ctx.saved_tensors
        eps, stable = ctx.cfg
        gmu = _f32(gmu) if gmu is not None else None
        glv = _f32(glv) if glv is not None else None
        dmus, dlvs = torch.empty_like(mus), torch.empty_like(lvs)
        call("mvk_poe_bwd", ptr(mus), ptr(lvs), mus.shape[0], mus[0].numel(), eps, stable, ptr(gmu), ptr(glv), ptr(dmus),
             ptr(dlvs), stream_ptr())
        return dmus, dlvs, None, None


def poe(mus, logvars, eps=1e-8):
    """Product of Gaussian experts over dim 0 (base_utils.py:122-130): var = exp(lv) + eps, T = 1 / var,
    mu = sum(mu T) / sum(T), logvar = log(1 / sum(T)).  -> (pd_mu, pd_logvar)"""
    return _PoEFn.apply(_experts(mus), _experts(logvars), eps, 0)


def stable_poe(mus, logvars):
    """The log-sum-exp form (base_utils.py:133-147): no eps, a single expert is returned as is, an expert with
    logvar = +inf (a missing modality) has weight exactly 0.  -> (joint_mu, ln_var)"""
    return _PoEFn.apply(_experts(mus), _experts(logvars), 0.0, 1)


class _KLFn(Function):
    @staticmethod
    def forward(ctx, mean, log_var, prior_mean, prior_log_var, shape):
        ops = [_f32(t) for t in (mean, log_var, prior_mean, prior_log_var)]
        L = shape[-1]
        rows = 1
        for s in shape[:-1]:
            rows *= s
        kl = torch.empty(shape[:-1], dtype=torch.float32, device=ops[0].device)
        call("mvk_kl_gauss_fwd", ptr(ops[0]), ops[0].numel(), ptr(ops[1]), ops[1].numel(), ptr(ops[2]), ops[2].numel(),
             ptr(ops[3]), ops[3].numel(), rows, L, ptr(kl), stream_ptr())
        ctx.save_for_backward(*ops)
        ctx.geom = (tuple(shape), rows, L)
        return kl

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        ops = ctx.saved_tensors
        shape, rows, L = ctx.geom
        g = _f32(g)
        need = ctx.needs_input_grad[:4]
        full = [torch.empty(shape, dtype=torch.float32, device=g.device) if n else None for n in need]
        call("mvk_kl_gauss_bwd", ptr(ops[0]), ops[0].numel(), ptr(ops[1]), ops[1].numel(), ptr(ops[2]), ops[2].numel(),
             ptr(ops[3]), ops[3].numel(), rows, L, ptr(g), ptr(full[0]), ptr(full[1]), ptr(full[2]), ptr(full[3]),
             stream_ptr())
        outs = []
        for t, d in zip(ops, full):
            if d is None:
                outs.append(None)
            elif t.numel() == rows * L:
                outs.append(d.view(t.shape))
            else:  # broadcast operand (indexed modulo its size): ordered column sums over the leading block
                from ... import kernels

                acc = torch.zeros(t.numel(), dtype=torch.float32, device=g.device)
                ws = kernels._ws(g)
                call("mvk_colsum_acc", ptr(d), None, 0, ptr(acc), rows * L // t.numel(), t.numel(), ptr(ws), ws.numel(),
                     stream_ptr())
                outs.append(acc.view(t.shape))
        return (*outs, None)


def kl_divergence(mean, log_var, prior_mean, prior_log_var):
    """KL(N(mean, exp(log_var)) || N(prior_mean, exp(prior_log_var))) summed over the last dimension (base_utils.py:90-119).
    Operands broadcast over leading dimensions (e.g. a [1, L] prior)."""
    shape = torch.broadcast_shapes(mean.shape, log_var.shape, prior_mean.shape, prior_log_var.shape)
    for t in (mean, log_var, prior_mean, prior_log_var):  # trailing-dimension broadcasting only (modulo indexing)
        lead = len(shape) - t.dim()
        if tuple(t.shape) != tuple(shape[lead:]) and t.numel() != 1:
            stripped = tuple(t.shape)
            while stripped and stripped[0] == 1:
                stripped = stripped[1:]
            if stripped != tuple(shape[len(shape) - len(stripped):]):
                raise ValueError(f"kl_divergence: operand of shape {tuple(t.shape)} does not broadcast along leading "
                                 f"dimensions of {tuple(shape)}")
    return _KLFn.apply(mean, log_var, prior_mean, prior_log_var, tuple(shape))


def rsample_from_gaussian(mu, log_var, N=1, return_mean=False, flatten=False):
    """z = mu + exp(log_var / 2) * eps, eps ~ N(0, I) of shape [N, *mu.shape] ([*mu.shape] for N = 1), drawn from the
    global generator like `Normal(mu, sigma).rsample` (base_utils.py:150-172)."""
    if return_mean:
        z = torch.stack([mu] * N) if N > 1 else mu
    else:
        from ... import kernels

        mu2 = _f32(mu).reshape(-1, mu.shape[-1])
        lv2 = _f32(log_var).reshape(-1, mu.shape[-1])
        eps = torch.randn((N,) + tuple(mu2.shape), dtype=torch.float32, device=mu.device)
        z = kernels.GaussSampleKLFn.apply(eps, mu2, lv2)[0].reshape((N,) + tuple(mu.shape))
        if N == 1:
            z = z[0]
    if (N > 1) and flatten:
        if len(z.shape) == 2:
            z = z.unsqueeze(0)
        z = z.reshape(-1, *z.shape[2:])
    return z


class _LogProbFn(Function):
    @staticmethod
    def forward(ctx, recon, target, dist, scale, eps):
        recon, target = _f32(recon), _f32(target.to(torch.float32))
        C = recon.shape[-1]
        lp = torch.empty_like(recon)
        call("mvk_logprob_fwd", ptr(recon), ptr(target), recon.numel(), target.numel(), dist, float(scale), C, float(eps),
             ptr(lp), stream_ptr())
        ctx.save_for_backward(recon, target)
        ctx.cfg = (dist, float(scale), C, float(eps))
        return lp

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        recon, target = ctx.saved_tensors
        dist, scale, C, eps = ctx.cfg
        g = _f32(g)
        dr = torch.empty_like(recon)
        call("mvk_logprob_bwd", ptr(recon), ptr(target), recon.numel(), target.numel(), dist, scale, C, eps, ptr(g), ptr(dr),
             stream_ptr())
        return dr, None, None, None, None


def _check_target(recon, target):
    lead = recon.dim() - target.dim()
    if lead < 0 or tuple(recon.shape[lead:]) != tuple(target.shape):
        raise ValueError(f"target of shape {tuple(target.shape)} does not match the trailing dimensions of the "
                         f"reconstruction {tuple(recon.shape)}")


def cross_entropy(input, target, eps=1e-6):
    """`x * log_softmax(input + eps)` over the last dimension, element-wise (base_utils.py:28-57).  `target`: one-hot /
    probabilities of the input's trailing shape, or a dict with "one_hot" or "tokens" (class ids) like the reference's text
    modalities."""
    _input = input
    if isinstance(_input, dict):
        _input = _input["one_hot"]
    _target = target
    if isinstance(target, dict):
        if "one_hot" in target:
            _target = target["one_hot"]
        elif "tokens" in target:
            _target = torch.nn.functional.one_hot(target["tokens"], _input.shape[-1])
    _check_target(_input, _target)
    return _LogProbFn.apply(_input, _target, _lib.DIST["categorical"], 1.0, eps)


def set_decoder_dist(dist_name, dist_params):
    """Distribution name + parameters -> callable `log_prob(recon, target)` returning ELEMENT-WISE log-probabilities
    (base_utils.py:62-87): normal / laplace with `scale` (default 1), bernoulli with logits = recon, categorical =
    `cross_entropy`.  `scale` is popped from `dist_params` like in the reference."""
    if dist_name == "normal" or dist_name == "laplace":
        scale = dist_params.pop("scale", 1.0)
        code = _lib.DIST[dist_name]

        def log_prob(recon, target):
            _check_target(recon, target)
            return _LogProbFn.apply(recon, target, code, scale, 0.0)

    elif dist_name == "bernoulli":

        def log_prob(recon, target):
            _check_target(recon, target)
            return _LogProbFn.apply(recon, target, _lib.DIST["bernoulli"], 1.0, 0.0)

    elif dist_name == "categorical":
        log_prob = cross_entropy
    else:
        raise ValueError("The distribution type 'dist' is not supported")
    return log_prob
