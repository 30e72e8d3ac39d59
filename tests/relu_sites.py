"""Which side of zero every ReLU / LeakyReLU unit lands on: the oracle and the HIP path side by side.

A rectifier unit whose fp32 pre-activation is within the summation noise of zero (~2e-7 of the layer's largest value) is
switched by the ORDER of the additions, not by the arithmetic: the reference's own CPU and GPU runs disagree on such units.
Its forward value is ~0 either way, but the gradient of everything upstream of it changes by that sample's whole
contribution through the unit — up to ~1e-2 of a weight-gradient tensor's largest entry at BASELINE's full sizes, where
7e7 units leave ~300 within 1e-6 of zero (tools/fullsize_probe.py).  A gradient comparison at 1e-4 therefore has to say
what it does about those units.  This module says it explicitly:

  1. the oracle runs once with `torch.nn.functional.relu` / `leaky_relu` recorded: every site's pre-activations
     (`OracleSites.record`);
  2. the HIP path's activations come from `multivae_amd.kernels.TAPS` (a test hook of the network nodes: tensors whose sign is
     the sign of the pre-activation, in the reference's evaluation order);
  3. `reconcile` asserts that the two agree on EVERY unit outside the ambiguous set A = {|pre| <= tau * max|pre| of the
     site} (tau = 2e-6), reports how many units of A the HIP path decided the other way, and returns its decisions;
  4. the oracle runs again with exactly those decisions (`OracleSites.force`) and the test compares every gradient entry
     by entry — no tensor is exempted, no distributional statement is needed.
"""
import contextlib
import inspect

import torch
import torch.nn.functional as F

from oracle import nets

TAU = 2e-6
NET_FNS = ("mlp_encoder", "mlp_decoder", "svhn_encoder", "svhn_decoder", "mlp_style_encoder", "mmnist_resnet_encoder",
           "mmnist_resnet_decoder", "cub_resnet_encoder", "cub_resnet_decoder", "joint_mlp_encoder", "joint_encoder_generic")


class OracleSites:
    """Records (`record`) or dictates (`force`) the rectifier decisions of oracle.nets' networks, keyed by the network's
    state-dict prefix ("encoders.svhn.") and the index of the activation inside one call of the network."""

    def __init__(self):
        self.pre = {}  # (prefix, site) -> list of pre-activation tensors, one per call of the network
        self.slope = {}  # (prefix, site) -> slope on the negative side (0 for ReLU)

    @contextlib.contextmanager
    def _patched(self, act):
        saved = {n: getattr(nets, n) for n in NET_FNS}
        orig_relu, orig_lrelu = F.relu, F.leaky_relu
        state = dict(prefix=None, site=0)

        def wrap(fn):
            sig = inspect.signature(fn)
            pos = list(sig.parameters).index("prefix")
            default = sig.parameters["prefix"].default

            def inner(*a, **k):
                prefix = k["prefix"] if "prefix" in k else (a[pos] if len(a) > pos else default)
                outer = dict(state)
                state.update(prefix=prefix, site=0)
                try:
                    return fn(*a, **k)
                finally:
                    state.update(outer)
            return inner

        def hook(orig, slope_of):
            def patched(x, *a, **k):
                if state["prefix"] is None:
                    return orig(x, *a, **k)
                key = (state["prefix"], state["site"])
                state["site"] += 1
                return act(key, x, slope_of(*a, **k), lambda t: orig(t, *a, **k))
            return patched

        for n, fn in saved.items():
            setattr(nets, n, wrap(fn))
        F.relu = hook(orig_relu, lambda *a, **k: 0.0)
        F.leaky_relu = hook(orig_lrelu, lambda negative_slope=0.01, *a, **k: float(negative_slope))
        try:
            yield self
        finally:
            F.relu, F.leaky_relu = orig_relu, orig_lrelu
            for n, fn in saved.items():
                setattr(nets, n, fn)

    def record(self):
        def act(key, x, slope, orig):
            self.pre.setdefault(key, []).append(x.detach())
            self.slope[key] = slope
            return orig(x)
        return self._patched(act)

    def force(self, forced):
        """forced[(prefix, site)][call] = bool tensor: True = the unit passes with slope 1."""
        calls = {}

        def act(key, x, slope, orig):
            i = calls.get(key, 0)
            calls[key] = i + 1
            on = forced[key][i]
            return x * torch.where(on, 1.0, slope).to(x.dtype)
        return self._patched(act)


def hip_signs(model, taps, sites):
    """(prefix, site) -> bool tensor, all calls of the network concatenated along the rows; 4-D taps are NHWC -> the oracle's
    NCHW.  A tap without a weight (the flatten-with-activation pass) belongs to the network of the tap before it."""
    names = {p.data_ptr(): n for n, p in model.named_parameters()}
    prefixes = sorted({p for p, _ in sites.pre}, key=len, reverse=True)
    per_net = {p: 1 + max(s for q, s in sites.pre if q == p) for p in prefixes}
    count, out, last = {}, {}, None
    for node, w, acts in taps:
        if w is not None:
            name = names[w.data_ptr()]
            last = next(p for p in prefixes if name.startswith(p))
        for h in acts:
            h = h.detach()
            if h.dim() == 4:
                h = h.permute(0, 3, 1, 2)
            j = count.get(last, 0)
            count[last] = j + 1
            key = (last, j % per_net[last])
            m = (h > 0).reshape(h.shape[0], -1).cpu()
            out.setdefault(key, []).append(m)
    return out


LAST = {}  # what the last reconcile() saw: units, ambiguous, flipped, {network prefix: flipped units}


def record_counts(case, engine, path=None):
    """Append the counts of the last reconcile() to gpurun_out/flip_counts.jsonl (the GPU box's scratch directory, merged back
    by gpurun; tools/make_profiles.py turns it into the tracked profiles/rNN_flip_counts.json).  Returns the record."""
    import json
    import os

    rec = dict(case=case, engine=engine, units=LAST.get("units", 0), ambiguous=LAST.get("ambiguous", 0),
               flipped=LAST.get("flipped", 0), flipped_by_network=LAST.get("by_network", {}))
    path = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "flip_counts.jsonl")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    return rec


def unreachable_by_flips(param_names, by_network=None):
    """The parameters whose gradient NO flipped unit can reach.  A flipped unit changes the backward pass through itself: the
    gradient of everything UPSTREAM of it in its own network (conservatively: the whole network), and — when it sits in a
    decoder — of everything that produced the decoder's input (every encoder, the joint encoder, the learnable priors whose
    samples the decoders read).  Decoders of other modalities, and every network when only encoders flipped, keep the
    reference's gradients: those are compared with the fixture's samples (the REFERENCE's own decisions) whatever flipped."""
    by_network = LAST.get("by_network", {}) if by_network is None else by_network
    hit = [p for p, c in by_network.items() if c]
    if not hit:
        return list(param_names)
    dec_flip = any(p.startswith("decoders.") for p in hit)
    out = []
    for k in param_names:
        if any(k.startswith(p) for p in hit):
            continue
        if dec_flip and not k.startswith("decoders."):
            continue
        out.append(k)
    return out


def reconcile(model, taps, sites, tau=TAU):
    """-> (decisions for OracleSites.force, number of ambiguous units, number of them the HIP path decided the other way).
    Asserts agreement on every unit outside the ambiguous set."""
    hip = hip_signs(model, taps, sites)
    assert set(hip) == set(sites.pre), (sorted(set(hip) ^ set(sites.pre)))
    forced, n_amb, n_flip = {}, 0, 0
    LAST.clear()
    LAST.update(units=0, ambiguous=0, flipped=0, by_network={})
    for key, pres in sites.pre.items():
        flat = torch.cat([p.reshape(-1) for p in pres])
        gm = torch.cat([m.reshape(-1) for m in hip[key]])
        assert gm.numel() == flat.numel(), (key, gm.numel(), flat.numel())
        amb = flat.abs() <= tau * float(flat.abs().max())
        om = flat > 0
        clear_bad = int(((om != gm) & ~amb).sum())
        assert clear_bad == 0, f"{key}: {clear_bad} units away from zero rectified differently by the HIP path"
        n_amb += int(amb.sum())
        n_flip += int(((om != gm) & amb).sum())
        LAST["units"] += flat.numel()
        LAST["by_network"][key[0]] = LAST["by_network"].get(key[0], 0) + int(((om != gm) & amb).sum())
        full = torch.where(amb, gm, om)
        parts, at = [], 0
        for p in pres:
            parts.append(full[at:at + p.numel()].reshape(p.shape))
            at += p.numel()
        forced[key] = parts
    LAST.update(ambiguous=n_amb, flipped=n_flip)
    # the ambiguous band is what the module's header says it is: ~10 units per million within 1e-6 of zero (x 2 for tau = 2e-6,
    # x 5 of margin); a HIP path that pushed many more units into it has a forward-pass problem, not a summation-order one
    assert n_amb <= 100e-6 * LAST["units"] + 5, (n_amb, LAST["units"])
    return forced, n_amb, n_flip


def oracle_with_hip_decisions(run_oracle, model, taps):
    """run_oracle() -> anything (it evaluates the oracle, forward + backward).  Runs it recorded, reconciles the rectifier
    decisions with the HIP path's taps, and — if the HIP path decided any ambiguous unit the other way — runs it again with
    those decisions.  -> (run_oracle's result, ambiguous units, flipped units)."""
    sites = OracleSites()
    with sites.record():
        res = run_oracle()
    forced, n_amb, n_flip = reconcile(model, taps, sites)
    if n_flip:
        with sites.force(forced):
            res = run_oracle()
    return res, n_amb, n_flip
