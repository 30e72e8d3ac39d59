"""Whole-step parity (forward + ELBO + backward + fused Adam) against the CPU oracle, and the BaseTrainer loop
(BASELINE.json configs[0]: MVTCAE, default MLP enc/dec, batch 64 — here on the GPU through the HIP kernels)."""
import os

import numpy as np
import pytest
import torch

import golden_cases as G
from oracle import elbo, nets
from oracle import train as otrain

pytestmark = pytest.mark.gpu


def test_three_training_steps_match_oracle():
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.models import MoPoE, MoPoEConfig
    from multivae_amd.trainers import FlatParams, FusedAdam

    d = torch.device("cuda:0")
    dims, L, B, K, lr = G.TINY_DIMS, 5, 12, 3, 1e-3
    shapes = G.P.default_mlp_shapes(dims, L)
    sd_np = G.P.make_state_dict(shapes, 321)
    model = MoPoE(MoPoEConfig(n_modalities=4, latent_dim=L, input_dims=dict(dims), beta=2.0, K=K))
    model.load_state_dict({k: G.t(v) for k, v in sd_np.items()})
    model = model.to(d).train()
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=lr)
    data = {m: G.t(G.P.uniform((B,) + s, 40 + i)) for i, (m, s) in enumerate(dims.items())}
    inputs = DatasetOutput(data={m: v.to(d) for m, v in data.items()})
    sd = {k: G.t(v).clone().requires_grad_(True) for k, v in sd_np.items()}
    st = otrain.AdamState(sd)
    names = list(dims)
    gen = torch.Generator().manual_seed(5)
    losses = []
    for step in range(3):
        eps = torch.randn(K, B, L, generator=gen)
        opt.zero_grad()
        out = model(inputs, noise=eps.to(d))
        out.loss.backward()
        opt.step()

        def loss_fn(s):
            enc_f, dec_f = nets.build_default_mlp(s, dims)
            e = {m: enc_f[m](data[m]) for m in names}
            return elbo.mopoe_forward(e, data, dec_f, eps, names=names, beta=2.0)

        o = otrain.train_step(sd, st, loss_fn, lr=lr)
        losses.append((float(out.loss.detach()), float(o["loss"].detach())))
    for a, b in losses:
        assert abs(a - b) <= 1e-4 * abs(b), losses
    bad = total = 0
    for k, p in model.named_parameters():
        diff = (p.detach().cpu() - sd[k].detach()).abs()
        bad += int((diff > 0.5 * lr).sum())
        total += diff.numel()
    assert bad <= 2e-3 * total, (bad, total)


def test_base_trainer_mvtcae_cfg1(tmp_path):
    """configs[0]: MVTCAE n_modalities=2, latent_dim=20, default MLP enc/dec, batch 64, BaseTrainer."""
    from multivae_amd.data.datasets.base import MultimodalBaseDataset
    from multivae_amd.models import MVTCAE, MVTCAEConfig
    from multivae_amd.trainers import BaseTrainer, BaseTrainerConfig

    torch.manual_seed(0)
    n = 640
    ds = MultimodalBaseDataset(data=dict(mnist=torch.rand(n, 1, 28, 28), svhn=torch.rand(n, 3, 32, 32)))
    ev = MultimodalBaseDataset(data=dict(mnist=torch.rand(128, 1, 28, 28), svhn=torch.rand(128, 3, 32, 32)))
    model = MVTCAE(MVTCAEConfig(n_modalities=2, latent_dim=20, input_dims=dict(mnist=(1, 28, 28), svhn=(3, 32, 32))))
    before = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = BaseTrainerConfig(output_dir=str(tmp_path), per_device_train_batch_size=64, per_device_eval_batch_size=64,
                            num_epochs=3, learning_rate=1e-3, steps_saving=2)
    trainer = BaseTrainer(model, train_dataset=ds, eval_dataset=ev, training_config=cfg)
    hist = trainer.train()
    assert len(hist) == 3 and all(np.isfinite(h["train_epoch_loss"]) for h in hist)
    assert hist[-1]["train_epoch_loss"] < hist[0]["train_epoch_loss"]
    assert set(hist[0]) >= {"train_epoch_loss", "eval_epoch_loss", "train_joint_divergence", "train_mnist", "train_kld_svhn"}
    after = trainer.model.state_dict()
    assert any(not torch.equal(before[k].to(after[k].device), after[k]) for k in before)
    tdir = trainer.training_dir
    assert set(os.listdir(os.path.join(tdir, "final_model"))) >= {"model.pt", "model_config.json", "environment.json",
                                                                  "training_config.json"}
    ck = os.path.join(tdir, "checkpoint_epoch_2")
    assert set(os.listdir(ck)) >= {"model.pt", "optimizer.pt", "model_config.json", "training_config.json",
                                   "info_checkpoint.json"}
    re = MVTCAE.load_from_folder(os.path.join(tdir, "final_model"))
    best = trainer._best_model.state_dict()
    assert all(torch.equal(v.cpu(), best[k].cpu()) for k, v in re.state_dict().items())


def _mnist_svhn_mopoe(d, K=3, L=8, seed=0):
    from multivae_amd.models import MoPoE, MoPoEConfig
    from multivae_amd.models.base.base_config import BaseAEConfig
    from multivae_amd.models.nn.default_architectures import Decoder_AE_MLP, Encoder_VAE_MLP
    from multivae_amd.models.nn.svhn import Decoder_VAE_SVHN, Encoder_VAE_SVHN

    torch.manual_seed(seed)
    dims = dict(mnist=(1, 28, 28), svhn=(3, 32, 32))
    enc = dict(mnist=Encoder_VAE_MLP(BaseAEConfig(latent_dim=L, input_dim=dims["mnist"])),
               svhn=Encoder_VAE_SVHN(BaseAEConfig(latent_dim=L, input_dim=dims["svhn"])))
    dec = dict(mnist=Decoder_AE_MLP(BaseAEConfig(latent_dim=L, input_dim=dims["mnist"])),
               svhn=Decoder_VAE_SVHN(BaseAEConfig(latent_dim=L, input_dim=dims["svhn"])))
    cfg = MoPoEConfig(n_modalities=2, latent_dim=L, input_dims=dims, beta=1.0, K=K)
    return MoPoE(cfg, enc, dec).to(d).train()


def _run_steps(d, steps, graphed, branch_streams=True, B=64, K=3, L=8):
    """Losses and final parameters of `steps` Adam steps; the two modality branches on separate streams or not,
    launches enqueued from Python or replayed from one hipGraph."""
    from multivae_amd import kernels
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.trainers import FlatParams, FusedAdam, GraphedStep

    old = kernels.BRANCH_STREAMS
    kernels.BRANCH_STREAMS = branch_streams
    try:
        model = _mnist_svhn_mopoe(d, K=K, L=L)
        flat = FlatParams(model)
        opt = FusedAdam(flat, lr=1e-3)
        g = torch.Generator().manual_seed(3)
        inputs = DatasetOutput(data=dict(mnist=torch.rand(B, 1, 28, 28, generator=g).to(d),
                                         svhn=torch.rand(B, 3, 32, 32, generator=g).to(d)))
        gs = GraphedStep(model, flat, inputs, noise=torch.zeros(K, B, L, device=d)) if graphed else None
        losses = []
        for _ in range(steps):
            eps = torch.randn(K, B, L, generator=g).to(d)
            if gs is not None:
                out = gs(inputs, eps)
            else:
                opt.zero_grad()
                out = model(inputs, noise=eps)
                out.loss.backward()
            opt.step()
            losses.append(float(out.loss.detach()))
        torch.cuda.synchronize()
        return losses, flat.dense(flat.flat.detach()).cpu().clone()
    finally:
        kernels.BRANCH_STREAMS = old


def _same_training(r1, r2, lr=1e-3):
    """Same kernels on the same data: only the order of the bias-gradient atomics may differ (1e-8 relative on a
    gradient), which Adam's g / sqrt(v) can turn into a different step for parameters whose gradient is ~0."""
    (l1, p1), (l2, p2) = r1, r2
    for a, b in zip(l1, l2):
        assert abs(a - b) <= 1e-6 * abs(a), (l1, l2)
    diff = (p1 - p2).abs()
    assert int((diff > 0.5 * lr).sum()) <= 1e-3 * diff.numel(), float(diff.max())


def test_branch_streams_do_not_change_the_result():
    """Encoders / decoders of the two modalities on separate HIP streams (kernels.run_branches) vs one stream."""
    d = torch.device("cuda:0")
    _same_training(_run_steps(d, 4, graphed=False, branch_streams=False), _run_steps(d, 4, graphed=False, branch_streams=True))


def test_graph_replay_matches_eager():
    """GraphedStep (zero_grad + forward + backward as one hipGraph replay) vs launches enqueued from Python."""
    d = torch.device("cuda:0")
    _same_training(_run_steps(d, 5, graphed=False), _run_steps(d, 5, graphed=True))


def test_user_decoder_with_autograd_reconstruction_nll_is_ordered_behind_the_loss_assembly():
    """the fused-tail protocol is duck-typed — a USER decoder may implement `reconstruction_nll` with ordinary
    autograd ops.  Its backward reads the row gradients the loss-assembly launch fills, so that launch must not move to the
    late-leaf stream for it (in a captured graph: a missing dependency, silently wrong gradients).  `kernels.orders_behind_loss`
    keeps `async_ok` off unless every fused term comes from one of the package's nodes: the gradients of such a model — eager and
    replayed — equal the generic path's (forward + likelihood kernel)."""
    import math

    from multivae_amd import kernels
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.models.base.base_config import BaseAEConfig
    from multivae_amd.models.nn.default_architectures import Decoder_AE_MLP
    from multivae_amd.trainers import FlatParams, GraphedStep

    class UserDecoder(Decoder_AE_MLP):
        def reconstruction_nll(self, z, x, dist="normal", scale=1.0, row_weight=None):
            rec = self.forward(z).reconstruction  # [K, B, *D] (the in-package forward node), then plain torch ops
            d = (rec - x) / scale
            n = x[0].numel()
            return 0.5 * (d * d).flatten(2).sum(-1) + n * (math.log(scale) + 0.5 * math.log(2 * math.pi))

    d = torch.device("cuda:0")
    B, K, L = 64, 4, 8
    g = torch.Generator().manual_seed(9)
    inputs = DatasetOutput(data=dict(mnist=torch.rand(B, 1, 28, 28, generator=g).to(d), svhn=torch.rand(B, 3, 32, 32, generator=g).to(d)))
    eps = torch.randn(K, B, L, generator=g).to(d)
    seen = []
    orig = kernels.orders_behind_loss
    kernels.orders_behind_loss = lambda rows: (seen.append(orig(rows)), seen[-1])[1]
    try:
        res = {}
        for mode in ("generic", "user eager", "user graph"):
            model = _mnist_svhn_mopoe(d, K=K, L=L)
            if mode != "generic":
                ud = UserDecoder(BaseAEConfig(latent_dim=L, input_dim=(1, 28, 28))).to(d)
                ud.load_state_dict(model.decoders["mnist"].state_dict())
                model.decoders["mnist"] = ud
            else:
                model.fused_decoder_tail = False
            flat = FlatParams(model)
            if mode == "user graph":
                gs = GraphedStep(model, flat, inputs, noise=torch.zeros(K, B, L, device=d))
                out = gs(inputs, eps)
            else:
                flat.zero_grad()
                with kernels.deferred_reductions(flat):
                    out = model(inputs, noise=eps)
                    out.loss.backward(gradient=kernels.unit_seed(out.loss))
            torch.cuda.synchronize()
            res[mode] = (float(out.loss), flat.dense(flat.grad).cpu().clone())
    finally:
        kernels.orders_behind_loss = orig
    assert seen and not any(seen)  # the user decoder's rows were recognised as foreign (it is asked first: all() stops there)
    stock = _mnist_svhn_mopoe(d, K=K, L=L)  # ... and the package's own fused tails as its own
    rows = stock.decoders["svhn"].reconstruction_nll(torch.randn(K, B, L, device=d, requires_grad=True), inputs.data["svhn"], "normal", 1.0)
    assert rows is not None and kernels.orders_behind_loss(rows)
    ref_l, ref_g = res["generic"]
    for mode in ("user eager", "user graph"):
        l, gr = res[mode]
        assert l == pytest.approx(ref_l, rel=1e-5), mode
        err = float((gr - ref_g).abs().max() / ref_g.abs().max())
        assert err <= 1e-4, (mode, err)


@pytest.mark.parametrize("amsgrad", [False, True])
def test_optimizer_inside_the_graph_matches_the_host_scalar_step(amsgrad):
    """the fused Adam as the LAST NODE of the replayed step (mvk_adam_prepare + mvk_adam_step_dev: step and
    learning rate in device memory) against replay + FusedAdam.step() with host scalars (mvk_adam_step_fused) — same kernels,
    same gradients, so the parameters agree to the last bits of the scalar arithmetic (device pow vs libm pow); the learning
    rate changes mid-run (what a scheduler does, base_trainer_config.py:60), an eager host-scalar step is interleaved, and the
    host's step counter / state_dict follow the replays."""
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.trainers import FlatParams, FusedAdam, GraphedStep

    d = torch.device("cuda:0")
    B, K, L = 64, 3, 8
    g = torch.Generator().manual_seed(3)
    inputs = DatasetOutput(data=dict(mnist=torch.rand(B, 1, 28, 28, generator=g).to(d), svhn=torch.rand(B, 3, 32, 32, generator=g).to(d)))
    noise = [torch.randn(K, B, L, generator=g).to(d) for _ in range(7)]
    res = []
    for in_graph in (False, True):
        model = _mnist_svhn_mopoe(d, K=K, L=L)
        flat = FlatParams(model)
        opt = FusedAdam(flat, lr=1e-3, amsgrad=amsgrad, zero_grad_in_step=True)
        before = flat.dense(flat.flat).clone()
        gs = GraphedStep(model, flat, inputs, noise=torch.zeros(K, B, L, device=d), optimizer=opt if in_graph else None)
        assert gs.includes_optimizer == in_graph
        assert torch.equal(before, flat.dense(flat.flat)), "warm-up and capture must not move the parameters"
        losses = []
        for i, eps in enumerate(noise):
            if i == 3:
                opt.lr = 3e-4  # a scheduler step
            if i == 5:  # one eager step in between (the last, ragged batch of an epoch goes this way)
                opt.zero_grad()
                out = model(inputs, noise=eps)
                out.loss.backward()
                opt.step()
            else:
                out = gs(inputs, eps)
                if not in_graph:
                    opt.step()
            losses.append(float(out.loss.detach()))
        torch.cuda.synchronize()
        assert opt.step_count == len(noise) and flat.grads_zero
        sd = opt.state_dict()
        assert float(sd["state"][0]["step"]) == len(noise) and sd["param_groups"][0]["lr"] == 3e-4
        res.append((losses, flat.dense(flat.flat).cpu().clone(), flat.dense(opt.m).cpu().clone(), flat.dense(opt.v).cpu().clone()))
    (l0, p0, m0, v0), (l1, p1, m1, v1) = res
    assert l0 == pytest.approx(l1, rel=1e-6)
    assert torch.allclose(m0, m1, rtol=1e-5, atol=1e-9) and torch.allclose(v0, v1, rtol=1e-5, atol=1e-12)
    diff = (p0 - p1).abs()
    assert float(diff.max()) <= 2e-6, float(diff.max())  # 7 steps of <= 1e-3: a ulp of the step size per step at most


@pytest.mark.parametrize("seed_kind", ["unit seed", "autograd's default seed"])
def test_loss_assembly_postponed_to_the_end_of_the_step(seed_kind):
    """Round 6: with the unit backward seed nothing inside the step reads what the loss assembly fills (the fused tails hold their
    gradient pre-multiplied, the posterior node takes the KL rows' constant from the host), so MoPoE lets the launch run LAST
    (`spec["assembly_last"]`, kernels.ASSEMBLY_LAST) — loss, every metric and the whole gradient buffer are bit for bit those of
    the launch at the head of the backward pass; with autograd's own seed (a tensor whose value the host does not know) the
    first reader runs the postponed launch itself (kernels.wait_loss) — same result."""
    from multivae_amd import kernels
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.trainers import FlatParams

    d = torch.device("cuda:0")
    B, K, L = 128, 10, 20
    g = torch.Generator().manual_seed(21)
    inputs = DatasetOutput(data=dict(mnist=torch.rand(B, 1, 28, 28, generator=g).to(d), svhn=torch.rand(B, 3, 32, 32, generator=g).to(d)))
    eps = torch.randn(K, B, L, generator=g).to(d)
    old = kernels.ASSEMBLY_LAST
    res, postponed = [], []
    orig_run_last = kernels.run_last
    try:
        for last in (False, True):
            kernels.ASSEMBLY_LAST = last
            kernels.run_last = lambda dev, fn, *a, **k: (postponed.append((last, getattr(fn, "__name__", ""))), orig_run_last(dev, fn, *a, **k))[1]
            model = _mnist_svhn_mopoe(d, K=K, L=L)
            flat = FlatParams(model)
            flat.zero_grad()
            with kernels.deferred_reductions(flat):
                out = model(inputs, noise=eps)
                if seed_kind == "unit seed":
                    out.loss.backward(gradient=kernels.unit_seed(out.loss))
                else:
                    out.loss.backward()
            torch.cuda.synchronize()
            res.append((float(out.loss.detach()), float(out.loss_sum.detach()), {k: float(v.detach()) for k, v in out.metrics.items()}, flat.grad.detach().clone()))
    finally:
        kernels.ASSEMBLY_LAST, kernels.run_last = old, orig_run_last
    assert (True, "assemble") in postponed and (False, "assemble") not in postponed  # the launch WAS handed to the postponed leaves
    (l0, s0, m0, g0), (l1, s1, m1, g1) = res
    assert l0 == l1 and s0 == s1 and m0 == m1
    assert torch.equal(g0, g1) and float(g0.abs().max()) > 0


@pytest.mark.parametrize("amsgrad,which", [(False, "both"), (True, "both"), (False, "svhn"), (False, "mlp"), (False, "small"), (True, "small")])
def test_rotated_step_is_bit_identical(amsgrad, which):
    """GraphedStep(rotate=optimizer) — the decoders' late weight gradients of step N, their ordered finishes,
    their share of Adam and the weight packs that read them run at the HEAD of replay N + 1 (kernels.Rotation), the caller's
    optimizer.step() covers the rest and publishes its scalars (mvk_adam_step_pub / mvk_adam_step_dev).  Same noise, same
    kernels, same scalars: after 7 steps + drain the parameters, both Adam moments (and max_exp_avg_sq) and every loss are
    BIT FOR BIT those of the unrotated GraphedStep — with a learning-rate change mid-run, a drain + eager step in the middle
    (the ragged last batch of an epoch), warm-up and capture moving nothing, and the optimizer refusing an eager step or a
    state_dict while an update is pending.  Decoder rows = 128 x 10: the scaled-fp16 convolution chain and the MLP decoder's
    fused tail (the two nodes that register rotatable leaves) take the batch."""
    from multivae_amd import kernels
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.trainers import FlatParams, FusedAdam, GraphedStep

    d = torch.device("cuda:0")
    B, K, L = 128, 10, 20
    g = torch.Generator().manual_seed(11)
    inputs = DatasetOutput(data=dict(mnist=torch.rand(B, 1, 28, 28, generator=g).to(d), svhn=torch.rand(B, 3, 32, 32, generator=g).to(d)))
    noise = [torch.randn(K, B, L, generator=g).to(d) for _ in range(7)]
    old = kernels.ROT_SVHN, kernels.ROT_MLP
    # which leaves are rotated: every weight-gradient leaf of both decoders / of one of them / only the two first-layer ones ("small")
    kernels.ROT_SVHN, kernels.ROT_MLP = {"both": (1, 1), "svhn": (1, 0), "mlp": (0, 1), "small": (2, 2)}[which]
    res = []
    try:
        for rotate in (False, True):
            model = _mnist_svhn_mopoe(d, K=K, L=L)
            flat = FlatParams(model)
            opt = FusedAdam(flat, lr=1e-3, amsgrad=amsgrad, zero_grad_in_step=True)
            before = flat.dense(flat.flat).clone()
            gs = GraphedStep(model, flat, inputs, noise=torch.zeros(K, B, L, device=d), rotate=opt if rotate else None)
            assert gs.rotated == rotate
            assert torch.equal(before, flat.dense(flat.flat)), "warm-up and capture must not move the parameters"
            if rotate:
                want = {"both": 6, "svhn": 3, "mlp": 3, "small": 2}[which]
                assert len(gs.rotation.params) == want and all(o >= flat.late_start for o, _ in opt._rot_ranges)
                if which in ("both", "small"):  # one contiguous range at the very end of the buffer: two optimizer launches per step
                    assert len(opt._rot_ranges) == 1 and sum(opt._rot_ranges[0]) == flat.numel
            losses = []
            for i, eps in enumerate(noise):
                if i == 3:
                    opt.lr = 3e-4  # a scheduler step
                if i == 5:  # one eager step in between: the pending update first
                    if rotate:
                        with pytest.raises(RuntimeError):
                            opt.state_dict()
                        with pytest.raises(RuntimeError):
                            opt.step()
                    gs.drain()
                    opt.zero_grad()
                    with kernels.deferred_reductions(flat):
                        out = model(inputs, noise=eps)
                        out.loss.backward()
                    opt.step()
                else:
                    out = gs(inputs, eps)
                    opt.step()
                losses.append(float(out.loss.detach()))
            gs.drain()
            gs.drain()  # idempotent
            torch.cuda.synchronize()
            assert opt.step_count == len(noise) and flat.grads_zero and float(flat.grad.abs().max()) == 0.0
            sd = opt.state_dict()
            assert float(sd["state"][0]["step"]) == len(noise)
            res.append((losses, flat.dense(flat.flat).cpu().clone(), flat.dense(opt.m).cpu().clone(), flat.dense(opt.v).cpu().clone(),
                        flat.dense(opt.vmax).cpu().clone() if amsgrad else None))
    finally:
        kernels.ROT_SVHN, kernels.ROT_MLP = old
    (l0, p0, m0, v0, x0), (l1, p1, m1, v1, x1) = res
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1), float((p0 - p1).abs().max())
    assert torch.equal(m0, m1) and torch.equal(v0, v1)
    if amsgrad:
        assert torch.equal(x0, x1)
    assert float((p0 - before.cpu()).abs().max()) > 1e-3  # ... and the run did train


@pytest.mark.parametrize("K,B", [(3, 16), (1, 24), (10, 64)])
def test_fused_decoder_tail_matches_the_generic_path(K, B):
    """MoPoE MnistSvhn with the SVHN decoder scoring its own output (Decoder_VAE_SVHN.reconstruction_nll: Normal NLL row sums and
    d rows / d pre-activation out of the last layer's epilogue, mvk_conv4s2_small_up_fwd_nll / _bwd_pre) against the same model
    decoding the images and running the generic likelihood kernel: loss, every metric, every gradient."""
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.trainers import FlatParams

    d = torch.device("cuda:0")
    L = 8
    model = _mnist_svhn_mopoe(d, K=K, L=L, seed=5)
    model.model_config.uses_likelihood_rescaling = True
    model.rescale_factors = dict(mnist=3072 / 784, svhn=1.0)
    flat = FlatParams(model)
    g = torch.Generator().manual_seed(21)
    inputs = DatasetOutput(data=dict(mnist=torch.rand(B, 1, 28, 28, generator=g).to(d), svhn=torch.rand(B, 3, 32, 32, generator=g).to(d)))
    eps = torch.randn(K, B, L, generator=g).to(d)
    res = {}
    for fused in (True, False):
        model.fused_decoder_tail = fused
        flat.zero_grad()
        out = model(inputs, noise=eps)
        out.loss.backward()
        torch.cuda.synchronize()
        res[fused] = (float(out.loss), {k: float(v) for k, v in out.metrics.items()}, flat.grad.detach().clone())
    lf, mf, gf = res[True]
    lp, mp, gp = res[False]
    assert abs(lf - lp) <= 1e-6 * abs(lp), (lf, lp)
    for k in mp:
        assert abs(mf[k] - mp[k]) <= 1e-6 * abs(mp[k]) + 1e-12, (k, mf[k], mp[k])
    for (name, p), off in zip(model.named_parameters(), flat.offsets):
        n = p.numel()
        a, b = gf[off:off + n], gp[off:off + n]
        scale = float(b.abs().max().clamp_min(1e-30))
        assert float((a - b).abs().max()) <= 3e-6 * scale, (name, float((a - b).abs().max()), scale)
    with torch.no_grad():  # evaluation never takes the fused tail (it has no gradient to prepare)
        model.fused_decoder_tail = True
        assert abs(float(model(inputs, noise=eps).loss) - lp) <= 1e-6 * abs(lp)


def test_deferred_reductions_match_immediate_finishes():
    """kernels.deferred_reductions (mvk_defer_begin / _end): every ordered finish of a weight / bias gradient queued and
    run in one launch at the end of the backward pass vs one launch each behind its producer.  Same partial results, a
    different (fixed) summation tree for the column-sum finishes; bit-identical from run to run; nothing left pending.
    n = K * B = 2560 images: the register-stationary convolution kernels and their slabs are on the path."""
    from multivae_amd import _lib, kernels
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.trainers import FlatParams

    d = torch.device("cuda:0")
    B, K, L = 256, 10, 20
    model = _mnist_svhn_mopoe(d, K=K, L=L)
    flat = FlatParams(model)
    g = torch.Generator().manual_seed(11)
    inputs = DatasetOutput(data=dict(mnist=torch.rand(B, 1, 28, 28, generator=g).to(d),
                                     svhn=torch.rand(B, 3, 32, 32, generator=g).to(d)))
    eps = torch.randn(K, B, L, generator=g).to(d)

    def run(deferred, settled=True):
        flat.zero_grad()
        if deferred:
            with kernels.deferred_reductions(flat) as ctx:
                assert ctx.on
                out = model(inputs, noise=eps)
                out.loss.backward()
                assert _lib.load().mvk_defer_pending() >= (10 if settled else 1)  # the finishes are queued, not run
        else:
            out = model(inputs, noise=eps)
            out.loss.backward()
        assert _lib.load().mvk_defer_pending() == 0
        torch.cuda.synchronize()
        return float(out.loss.detach()), flat.grad.detach().clone()

    l0, g0 = run(False)
    run(True, settled=False)  # the arena grows to what the step asks for (kernels.deferred_reductions): which finishes are queued settles here
    l1, g1 = run(True)
    l2, g2 = run(True)
    assert l0 == l1 == l2
    assert torch.equal(g1, g2), "deferred finishes are not reproducible"
    for (name, p), off in zip(model.named_parameters(), flat.offsets):  # FlatParams lays the gradients out in this order
        n = p.numel()
        a, b = g1[off:off + n], g0[off:off + n]
        scale = float(b.abs().max().clamp_min(1e-30))
        assert float((a - b).abs().max()) <= 2e-6 * scale, (name, float((a - b).abs().max()), scale)


def test_deferred_reductions_fall_back_when_the_arena_is_full():
    """A 1 MB arena holds a few of the ~30 partial-result regions of a step: the producers that do not fit finish
    immediately (their own launch), the others are queued — same gradients as with the full arena."""
    from multivae_amd import kernels
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.trainers import FlatParams

    d = torch.device("cuda:0")
    B, K, L = 256, 10, 20
    model = _mnist_svhn_mopoe(d, K=K, L=L)
    flat = FlatParams(model)
    g = torch.Generator().manual_seed(12)
    inputs = DatasetOutput(data=dict(mnist=torch.rand(B, 1, 28, 28, generator=g).to(d),
                                     svhn=torch.rand(B, 3, 32, 32, generator=g).to(d)))
    eps = torch.randn(K, B, L, generator=g).to(d)

    def run():
        flat.zero_grad()
        with kernels.deferred_reductions(flat):
            out = model(inputs, noise=eps)
            out.loss.backward()
        torch.cuda.synchronize()
        return flat.grad.detach().clone()

    full = run()
    saved = kernels._ARENA.get(d)
    kernels._ARENA[d] = torch.empty(1 << 18, dtype=torch.float32, device=d)
    try:
        small = run()
    finally:
        kernels._ARENA[d] = saved
    scale = float(full.abs().max())
    assert float((small - full).abs().max()) <= 2e-6 * scale


@pytest.mark.parametrize("model_name,graph_opt", [("MoPoE", False), ("JMVAE", False), ("MVAE", False), ("MoPoE", True), ("JMVAE", True)])
def test_trainer_with_hip_graph(tmp_path, model_name, graph_opt):
    """BaseTrainerConfig.use_hip_graph: every batch shape gets one captured graph (the full batches and the short last
    one; fresh noise on every replay through the registered default generator), JMVAE re-captures while its annealing
    factor changes.  The loss goes down like in the eager run of the same seed."""
    from multivae_amd.data.datasets.base import MultimodalBaseDataset
    from multivae_amd.models import JMVAE, MVAE, JMVAEConfig, MoPoE, MoPoEConfig, MVAEConfig
    from multivae_amd.trainers import BaseTrainer, BaseTrainerConfig

    def run(use_graph):
        torch.manual_seed(0)
        n = 64 * 5 + 24  # a short last batch
        ds = MultimodalBaseDataset(data=dict(a=torch.rand(n, 1, 28, 28), b=torch.rand(n, 40)))
        dims = dict(a=(1, 28, 28), b=(40,))
        if model_name == "MoPoE":
            model = MoPoE(MoPoEConfig(n_modalities=2, latent_dim=12, input_dims=dims))
        elif model_name == "MVAE":  # epoch 1: the KL weight changes every batch -> eager (graph_key is False)
            model = MVAE(MVAEConfig(n_modalities=2, latent_dim=12, input_dims=dims, warmup=2))
        else:
            model = JMVAE(JMVAEConfig(n_modalities=2, latent_dim=12, input_dims=dims, warmup=2))
        cfg = BaseTrainerConfig(output_dir=str(tmp_path), per_device_train_batch_size=64, num_epochs=4,
                                learning_rate=1e-3, use_hip_graph=use_graph, graph_optimizer=graph_opt and use_graph,
                                scheduler_cls="StepLR" if graph_opt else None,
                                scheduler_params=dict(step_size=2, gamma=0.5) if graph_opt else None)
        trainer = BaseTrainer(model, train_dataset=ds, training_config=cfg)
        hist = trainer.train()
        if graph_opt:  # graph_optimizer: the optimizer is the last node of every captured graph; the scheduler still drives it
            assert trainer.optimizer.step_count == 4 * 6 and trainer.optimizer.lr == pytest.approx(1e-3 * 0.25)
            if use_graph:
                assert all(g.includes_optimizer for g in trainer._graphs.values() if g is not None)
        metrics.append([{k: v for k, v in h.items() if k.startswith("train_") and k != "train_epoch_loss"} for h in hist])
        return [h["train_epoch_loss"] for h in hist], trainer

    metrics = []
    eager, _ = run(False)
    graphed, tr = run(True)
    # epoch metrics are means over the batches in both modes (a replayed graph overwrites its output tensors in
    # place: the running sums must not alias them)
    for me, mg in zip(*metrics):
        for k, v in me.items():
            if isinstance(v, float) and np.isfinite(v) and abs(v) > 1e-6:
                assert abs(mg[k] - v) <= 0.1 * abs(v), (k, v, mg[k])
    graphs = [g for g in tr._graphs.values() if g is not None]
    assert len(graphs) == (4 if model_name == "JMVAE" else 2), tr._graphs.keys()  # JMVAE: x2 for epochs 1 and >= 2
    assert all(np.isfinite(v) for v in graphed) and graphed[-1] < graphed[0]
    for a, b in zip(eager, graphed):  # same data order and initial weights, different noise stream
        assert abs(a - b) <= 0.05 * abs(a), (eager, graphed)


def test_trainer_rotate_step_is_bit_identical(tmp_path):
    """BaseTrainerConfig.rotate_step (the rotated GraphedStep under the trainer's loop): two epochs of three full batches + a ragged
    one (captured as a graph of its own: the pending update of the other shape is drained first), a StepLR scheduler, a checkpoint
    per epoch — the trainer drains before eager steps, shape changes and at the end of every epoch, so the final parameters and
    optimizer moments are bit for bit those of the run without rotation (same seed: same data order, same device noise)."""
    from multivae_amd.data.datasets.base import MultimodalBaseDataset
    from multivae_amd.trainers import BaseTrainer, BaseTrainerConfig

    d = torch.device("cuda:0")
    res = []
    for rotate in (False, True):
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        n = 128 * 3 + 112
        ds = MultimodalBaseDataset(data=dict(mnist=torch.rand(n, 1, 28, 28), svhn=torch.rand(n, 3, 32, 32)))
        model = _mnist_svhn_mopoe(d, K=10, L=20)
        cfg = BaseTrainerConfig(output_dir=str(tmp_path / str(rotate)), per_device_train_batch_size=128, num_epochs=2,
                                learning_rate=1e-3, use_hip_graph=True, rotate_step=rotate, steps_saving=1,
                                scheduler_cls="StepLR", scheduler_params=dict(step_size=1, gamma=0.5))
        trainer = BaseTrainer(model, train_dataset=ds, training_config=cfg)
        hist = trainer.train()
        graphs = [g for g in trainer._graphs.values() if g is not None]
        assert len(graphs) == 2 and all(g.rotated == rotate for g in graphs)
        assert trainer.optimizer.step_count == 8 and not trainer.optimizer._rot_dirty
        res.append(([h["train_epoch_loss"] for h in hist], trainer.flat.dense(trainer.flat.flat).cpu().clone(),
                    trainer.flat.dense(trainer.optimizer.m).cpu().clone(), trainer.flat.dense(trainer.optimizer.v).cpu().clone()))
    (l0, p0, m0, v0), (l1, p1, m1, v1) = res
    assert l0 == l1 and torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1), (l0, l1, float((p0 - p1).abs().max()))


def test_cfg4_cfg5_architectures_take_training_steps():
    """BASELINE.json configs[3] / configs[4] in miniature: MMVAE+ with the PolyMNIST ResNets (K samples, iwae_looser,
    laplace, beta 2.5) and JMVAE with the 64x64 CUB ResNet + the default MLP for a Bernoulli attribute vector: a few
    Adam steps on the flat buffers run, the loss is finite and goes down."""
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.models import JMVAE, JMVAEConfig, MMVAEPlus, MMVAEPlusConfig
    from multivae_amd.models.base.base_config import BaseAEConfig
    from multivae_amd.models.nn.cub import CUB_Resnet_Decoder, CUB_Resnet_Encoder
    from multivae_amd.models.nn.default_architectures import Decoder_AE_MLP, Encoder_VAE_MLP
    from multivae_amd.models.nn.mmnist import DecoderResnetMMNIST, EncoderResnetMMNIST
    from multivae_amd.trainers import FlatParams, FusedAdam

    d = torch.device("cuda:0")
    torch.manual_seed(0)
    mods = ["m0", "m1", "m2"]
    cfg4 = MMVAEPlusConfig(n_modalities=3, latent_dim=32, input_dims={m: (3, 28, 28) for m in mods}, K=2,
                           modalities_specific_dim=32, beta=2.5, loss="iwae_looser",
                           prior_and_posterior_dist="laplace_with_softmax",
                           decoders_dist={m: "laplace" for m in mods}, decoder_dist_params={m: dict(scale=0.75) for m in mods})
    m4 = MMVAEPlus(cfg4, {m: EncoderResnetMMNIST(32, 32) for m in mods}, {m: DecoderResnetMMNIST(64) for m in mods}).to(d)
    in4 = DatasetOutput(data={m: torch.rand(6, 3, 28, 28, device=d) for m in mods})
    cfg5 = JMVAEConfig(n_modalities=2, latent_dim=16, input_dims=dict(image=(3, 64, 64), attributes=(18,)),
                       decoders_dist=dict(image="normal", attributes="bernoulli"))
    enc5 = dict(image=CUB_Resnet_Encoder(16), attributes=Encoder_VAE_MLP(BaseAEConfig(latent_dim=16, input_dim=(18,))))
    dec5 = dict(image=CUB_Resnet_Decoder(16), attributes=Decoder_AE_MLP(BaseAEConfig(latent_dim=16, input_dim=(18,))))
    m5 = JMVAE(cfg5, enc5, dec5).to(d)
    in5 = DatasetOutput(data=dict(image=torch.rand(4, 3, 64, 64, device=d),
                                  attributes=(torch.rand(4, 18, device=d) > 0.5).float()))
    for model, inputs in ((m4, in4), (m5, in5)):
        model.train()
        flat = FlatParams(model)
        opt = FusedAdam(flat, lr=1e-3)
        losses = []
        for _ in range(4):
            opt.zero_grad()
            out = model(inputs, epoch=20)
            out.loss.backward()
            opt.step()
            losses.append(float(out.loss.detach()))
        assert all(np.isfinite(v) for v in losses), losses
        # (not losses[-1]: with 4-6 samples per batch a single step can jump on an unlucky noise draw, under any generator)
        assert min(losses[1:]) < losses[0], losses
        assert float(flat.grad.abs().max()) > 0


def test_resume_from_checkpoint_continues_the_same_training(tmp_path):
    """checkpoint_epoch_N (model.pt + optimizer.pt in torch.optim.Adam layout + info_checkpoint.json +
    metrics_best_model.json, base_trainer.py:777-828) -> BaseTrainer(checkpoint=...) resumes at epoch N+1 with the
    optimizer moments and step count of the interrupted run (resume_training, :402-440)."""
    import json

    from multivae_amd.data.datasets.base import MultimodalBaseDataset
    from multivae_amd.models import MVTCAE, MVTCAEConfig
    from multivae_amd.trainers import BaseTrainer, BaseTrainerConfig

    torch.manual_seed(0)
    dims = dict(a=(6,), b=(2, 5))
    ds = MultimodalBaseDataset(data=dict(a=torch.rand(96, 6), b=torch.rand(96, 2, 5)))

    def make():
        torch.manual_seed(1)
        return MVTCAE(MVTCAEConfig(n_modalities=2, latent_dim=4, input_dims=dims))

    def cfg(out, epochs):
        return BaseTrainerConfig(output_dir=str(out), per_device_train_batch_size=32, num_epochs=epochs,
                                 learning_rate=1e-3, steps_saving=1, use_hip_graph=False)

    t1 = BaseTrainer(make(), train_dataset=ds, training_config=cfg(tmp_path / "a", 1))
    t1.train()
    ck = os.path.join(t1.training_dir, "checkpoint_epoch_1")
    with open(os.path.join(ck, "info_checkpoint.json")) as f:
        info = json.load(f)
    assert set(info) == {"training_dir", "trained_epochs", "best_train_loss", "best_eval_loss"} and info["trained_epochs"] == 1
    with open(os.path.join(ck, "metrics_best_model.json")) as f:
        assert "train_epoch_loss" in json.load(f)
    osd = torch.load(os.path.join(ck, "optimizer.pt"), map_location="cpu")
    assert set(osd) == {"state", "param_groups"} and float(osd["state"][0]["step"]) == 3.0
    m_after_1 = {k: v.clone() for k, v in t1._best_model.state_dict().items()}
    t2 = BaseTrainer(make(), train_dataset=ds, training_config=cfg(tmp_path / "b", 3), checkpoint=ck)
    assert all(torch.equal(v.cpu(), m_after_1[k].cpu()) for k, v in t2.model.state_dict().items())
    hist = t2.train()
    assert len(hist) == 2  # epochs 2 and 3 only
    assert t2.optimizer.step_count == 9
    assert t2.training_dir == info["training_dir"]
    assert os.path.isdir(os.path.join(t2.training_dir, "checkpoint_epoch_3"))
    assert hist[-1]["train_epoch_loss"] < float(json.load(open(os.path.join(ck, "metrics_best_model.json")))["train_epoch_loss"])


@pytest.mark.parametrize("model_name", ["MVAE", "CRMVAE", "DMVAE"])
@pytest.mark.parametrize("masked", [False, True])
def test_trainer_runs_the_poe_family(tmp_path, model_name, masked):
    """MVAE / CRMVAE / DMVAE through BaseTrainer on complete and incomplete data: finite decreasing loss, metrics
    averaged over batches, checkpoint reloadable through AutoModel."""
    from multivae_amd.data.datasets.base import IncompleteDataset, MultimodalBaseDataset
    from multivae_amd.models import (CRMVAE, DMVAE, MVAE, AutoModel, CRMVAEConfig, DMVAEConfig, MVAEConfig)
    from multivae_amd.trainers import BaseTrainer, BaseTrainerConfig

    torch.manual_seed(0)
    n = 160
    dims = dict(a=(12,), b=(2, 5), c=(7,))
    data = {m: torch.rand(n, *d) for m, d in dims.items()}
    if masked:
        masks = {m: torch.rand(n) > 0.3 for m in dims}
        masks["a"][:] = True
        ds = IncompleteDataset(data=data, masks=masks)
    else:
        ds = MultimodalBaseDataset(data=data)
    common = dict(n_modalities=3, latent_dim=6, input_dims=dims)
    if model_name == "MVAE":
        model = MVAE(MVAEConfig(k=1, warmup=2, **common))
    elif model_name == "CRMVAE":
        model = CRMVAE(CRMVAEConfig(**common))
    else:
        model = DMVAE(DMVAEConfig(modalities_specific_dim=dict(a=2, b=3, c=2), **common))
    cfg = BaseTrainerConfig(output_dir=str(tmp_path), per_device_train_batch_size=32, num_epochs=4, learning_rate=2e-3,
                            steps_saving=4)
    trainer = BaseTrainer(model, train_dataset=ds, training_config=cfg)
    hist = trainer.train()
    losses = [h["train_epoch_loss"] for h in hist]
    assert all(np.isfinite(v) for v in losses), losses
    if model_name != "MVAE":  # MVAE's KL weight rises during its warm-up epochs
        assert losses[-1] < losses[0], losses
    assert all(np.isfinite(float(v)) for h in hist for v in h.values())
    back = AutoModel.load_from_folder(os.path.join(trainer.training_dir, "final_model"))
    assert type(back) is type(model)


def test_full_size_step_properties():
    """BASELINE.json headline size (MoPoE MnistSvhn, B = 512, K = 10), where the oracle would need minutes: properties
    that do not depend on the size.
      * K-linearity: the K-sample loss is the mean over k of the single-sample losses on the same noise rows (same
        analytic KL);
      * K-linearity of every parameter gradient, checked with IDENTICAL launch shapes (sample k repeated K times): a
        K = 1 run takes other kernels, whose last-bit differences flip the ReLU mask of a unit with a ~1e-10
        pre-activation now and then, which moves a cancelling gradient sum by a whole sample's contribution (measured:
        one flip in 8.4 M units changes decoders.svhn.dec.0.weight by 2 % of its largest entry -- in fp32 on any device);
      * the K axis is a batch axis of the decoders and of the reconstruction kernel: permuting the K noise slabs leaves
        loss and gradients unchanged (up to summation order);
      * hipGraph replay = eager launches, bit for bit in the loss."""
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.trainers import FlatParams, GraphedStep

    d = torch.device("cuda:0")
    B, K, L = 512, 10, 20
    model = _mnist_svhn_mopoe(d, K=K, L=L)
    g = torch.Generator().manual_seed(5)
    inputs = DatasetOutput(data=dict(mnist=torch.rand(B, 1, 28, 28, generator=g).to(d),
                                     svhn=torch.rand(B, 3, 32, 32, generator=g).to(d)))
    eps = torch.randn(K, B, L, generator=g).to(d)

    def run(noise, Kk):
        model.zero_grad(set_to_none=True)
        out = model(inputs, noise=noise, K=Kk)
        out.loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        return float(out.loss.detach()), grads, out

    def worst(ga, gb):
        return max(float((ga[n] - gb[n]).abs().max() / gb[n].abs().max().clamp_min(1e-30)) for n in ga)

    loss_k, grads_k, out_k = run(eps, K)
    single = sum(run(eps[k:k + 1].contiguous(), 1)[0] for k in range(K)) / K
    assert abs(loss_k - single) <= 2e-6 * abs(loss_k), (loss_k, single)
    acc_loss, acc = 0.0, None
    for k in range(K):
        lk, gk, _ = run(eps[k:k + 1].expand(K, B, L).contiguous(), K)
        acc_loss += lk / K
        acc = gk if acc is None else {n: acc[n] + gk[n] for n in acc}
    assert abs(loss_k - acc_loss) <= 2e-6 * abs(loss_k), (loss_k, acc_loss)
    assert worst(grads_k, {n: v / K for n, v in acc.items()}) <= 2e-5
    perm = torch.randperm(K, generator=g)
    loss_p, grads_p, _ = run(eps[perm].contiguous(), K)
    assert abs(loss_p - loss_k) <= 1e-6 * abs(loss_k)
    assert worst(grads_p, grads_k) <= 2e-5
    # graph replay on the same noise
    flat = FlatParams(model)
    gs = GraphedStep(model, flat, inputs, noise=eps)
    out_g = gs(inputs, eps)
    assert float(out_g.loss.detach()) == loss_k
    assert float(out_g.metrics["joint_divergence"]) == float(out_k.metrics["joint_divergence"])


def test_fused_adam_amsgrad_and_scheduler_in_the_trainer(tmp_path):
    """The reference's MMVAE+ optimizer setting (`optimizer_params=dict(amsgrad=True)`, examples/mmvae_plus/mmnist.py:61-62)
    and an lr scheduler stay on the fused one-launch Adam; the run matches torch.optim.Adam(amsgrad) + StepLR on the same
    model / seed (use_fused_adam=False) and the checkpoint carries max_exp_avg_sq and scheduler.pt."""
    from multivae_amd.data.datasets.base import MultimodalBaseDataset
    from multivae_amd.models import MVTCAE, MVTCAEConfig
    from multivae_amd.trainers import BaseTrainer, BaseTrainerConfig, FusedAdam

    def run(fused):
        torch.manual_seed(0)
        ds = MultimodalBaseDataset(data=dict(a=torch.rand(96, 12), b=torch.rand(96, 7)))
        model = MVTCAE(MVTCAEConfig(n_modalities=2, latent_dim=4, input_dims=dict(a=(12,), b=(7,))))
        cfg = BaseTrainerConfig(output_dir=str(tmp_path / ("f" if fused else "t")), per_device_train_batch_size=32,
                                num_epochs=3, learning_rate=2e-3, optimizer_params=dict(amsgrad=True, weight_decay=0.01),
                                scheduler_cls="StepLR", scheduler_params=dict(step_size=1, gamma=0.5), steps_saving=3,
                                use_fused_adam=fused, seed=3)
        tr = BaseTrainer(model, train_dataset=ds, training_config=cfg)
        hist = tr.train()
        return tr, [h["train_epoch_loss"] for h in hist], {k: v.detach().cpu().clone() for k, v in model.named_parameters()}

    tf, lf, pf = run(True)
    tt, lt, pt = run(False)
    assert isinstance(tf.optimizer, FusedAdam) and tf.optimizer.amsgrad and not isinstance(tt.optimizer, FusedAdam)
    assert tf.optimizer.lr == pytest.approx(2e-3 * 0.5 ** 3) and tf.optimizer.step_count == 9
    for a, b in zip(lf, lt):
        assert abs(a - b) <= 1e-4 * abs(b), (lf, lt)
    for k in pf:
        assert float((pf[k] - pt[k]).abs().max()) <= 2e-4 * float(pt[k].abs().max()) + 1e-6, k
    ck = os.path.join(tf.training_dir, "checkpoint_epoch_3")
    sd = torch.load(os.path.join(ck, "optimizer.pt"), map_location="cpu")
    assert sd["param_groups"][0]["amsgrad"] is True and "max_exp_avg_sq" in next(iter(sd["state"].values()))
    assert os.path.exists(os.path.join(ck, "scheduler.pt"))
    ref_opt = torch.optim.Adam(tt.model.parameters(), lr=1.0, amsgrad=True)
    ref_opt.load_state_dict(sd)  # loads into the reference's optimizer class


def test_rccl_path_on_one_gpu_matches_the_plain_step(tmp_path):
    """bench.py with MVK_FORCE_DIST=1: RCCL initialisation, the parameter broadcast, ONE all-reduce of the flat gradient
    buffer per step and the 1/world_size folded into Adam on a single GPU (world_size 1) give the loss of the plain run."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--batch", "64"]
    outs = []
    for force in ("0", "1"):
        env = dict(os.environ, MVK_FORCE_DIST=force, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0",
                   WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        outs.append(json.loads(line))
    assert outs[0]["config"]["final_loss"] == pytest.approx(outs[1]["config"]["final_loss"], rel=1e-6)
    assert outs[1]["n_gpus"] == 1


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu():
    """The driver's multi-GPU launch line (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) with N = 2
    on ONE GPU (gloo instead of RCCL, both ranks on cuda:0): every rank runs the same collectives in the same order — rank 0's
    profiling section issues none —, the line reports the whole-job rate and the weak-scaling fields."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MVK_DIST_BACKEND="gloo", MVK_BENCH_SAME_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "64"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 128
    assert d["value"] == pytest.approx(2 * 64 * 4 / (d["ms_per_step"] * 4e-3), rel=1e-3)
    assert "cpu_baseline" not in d and "roofline" in d


@pytest.mark.timeout(900)
def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """VERDICT r4 missing #1: `python bench.py --gpus 2` with NO launcher environment re-executes itself under
    torch.distributed.run (one process per rank, as base_trainer_config.py:80-100 expects them) and the printed n_gpus is the
    size of the gradient collective's group — here two ranks on ONE GPU over gloo (RCCL refuses a duplicate device)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MVK_DIST_BACKEND="gloo", MVK_BENCH_SAME_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "64"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 128
    assert "gloo" in d["config"]["collective"]
    # a launcher environment of another size than --gpus: refused, nothing on stdout
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r2 = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 2 and not r2.stdout.strip(), (r2.returncode, r2.stdout[-500:])


# ------------------------------------------------------------------------------------------------------------------------
# The REAL distributed step with world_size 2: two processes on cuda:0, gloo all-reducing the CUDA
# gradient buffer, BaseTrainer with FusedAdam + hipGraph replay (thread-local capture) on MoPoE MnistSvhn.
# Reference semantics: base_trainer.py:92-117 (DDP average), :198-211 (DistributedSampler without set_epoch), :748 (epoch loss
# over the full dataset length); MoPoE's row-range subset selection is per local shard (mopoe_model.py:444-455).
# ------------------------------------------------------------------------------------------------------------------------
_DP = dict(n=256, bs=32, epochs=2, K=2, L=8, lr=1e-3, seed=11)


def _dp_dataset():
    from multivae_amd.data.datasets.base import MultimodalBaseDataset

    n = _DP["n"]
    return MultimodalBaseDataset(data=dict(mnist=G.t(G.P.uniform((n, 1, 28, 28), 9001)), svhn=G.t(G.P.uniform((n, 3, 32, 32), 9002))))


def _dp_worker(rank, world, port, ret, outdir, overlap=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from multivae_amd.trainers import BaseTrainer, BaseTrainerConfig, FusedAdam
    from multivae_amd.trainers.flat import FlatParams

    d = torch.device("cuda:0")
    model = _mnist_svhn_mopoe(d, K=_DP["K"], L=_DP["L"], seed=100 + rank)  # different weights per rank: rank 0's are broadcast
    calls, range_calls = [], []
    orig = FlatParams.all_reduce_mean
    FlatParams.all_reduce_mean = lambda self, group=None: (calls.append(1), orig(self, group))[1]
    orig_r = FlatParams.all_reduce_mean_ranges
    FlatParams.all_reduce_mean_ranges = lambda self, ranges, group=None: (
        range_calls.append((torch.cuda.current_stream().cuda_stream, [tuple(r) for r in ranges])), orig_r(self, ranges, group))[1]
    cfg = BaseTrainerConfig(output_dir=outdir, per_device_train_batch_size=_DP["bs"], num_epochs=_DP["epochs"],
                            learning_rate=_DP["lr"], optimizer_cls="Adam", use_fused_adam=True, use_hip_graph=True,
                            dist_backend="gloo", world_size=world, rank=rank, local_rank=0, seed=_DP["seed"], steps_saving=None,
                            overlap_collective=overlap)
    tr = BaseTrainer(model, _dp_dataset(), training_config=cfg)
    hist = tr.train()
    torch.cuda.synchronize()
    assert isinstance(tr.optimizer, FusedAdam)
    graphs = [g for g in tr.__dict__.get("_graphs", {}).values() if g is not None]
    # the overlapped collective: what the capture marked as final early / late, in parameter names
    g0 = graphs[0]
    late_names = sorted(k for k, p in model.named_parameters()
                        if any(lo <= tr.flat.ranges_of([p])[0][0] < lo + n for lo, n in (g0.late_ranges or [])))
    covered = sorted((g0.early_ranges or []) + (g0.late_ranges or []))
    ret[rank] = dict(params=tr.flat.dense(tr.flat.flat).cpu(), loss=[h["train_epoch_loss"] for h in hist], allreduce=len(calls),
                     steps=tr.optimizer.step_count, graphs=len(graphs), range_calls=len(range_calls),
                     comm_streams=len({c[0] for c in range_calls}), main_stream=torch.cuda.current_stream().cuda_stream,
                     streams_used=[c[0] for c in range_calls[:2]], late_names=late_names, covered=covered, numel=tr.flat.numel,
                     first_two=[c[1] for c in range_calls[:2]], early=g0.early_ranges, late=g0.late_ranges)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("overlap", [False, True])
def test_distributed_step_two_ranks_on_one_gpu(tmp_path, overlap):
    """world_size 2 on ONE GPU: GraphedStep (thread-local capture, overlap point recorded) -> reduce_and_step: the early ranges'
    all-reduce behind the graph's event node on the communication stream, the late ranges behind the replay (gloo on the CUDA
    buffer), FusedAdam.step(grad_scale = 1 / W).  The ranks end bit-identical; they equal a single-process replay in which every step's
    gradient is the mean over the ranks of the per-shard gradients (each shard through MoPoE on its own, i.e. with its
    own row-range subset selection) — the noise of step t is draw number `warmup + t` of the device generator after
    `set_seed` (the eager warm-up passes of the capture consume the first ones), identical on both ranks as under the
    reference's DDP (every rank seeds alike); the epoch loss is the local sum over the FULL dataset length; flat.grad is
    all-reduced exactly once per step."""
    import inspect
    import socket

    import torch.multiprocessing as mp

    from multivae_amd import kernels
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.trainers import FlatParams, FusedAdam
    from multivae_amd.trainers.base import shard_indices
    from multivae_amd.trainers.base.base_trainer import set_seed
    from multivae_amd.trainers.graph import GraphedStep

    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, ret, str(tmp_path), overlap), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    n, bs, epochs, K, L = _DP["n"], _DP["bs"], _DP["epochs"], _DP["K"], _DP["L"]
    steps = epochs * (n // world // bs)
    assert r0["steps"] == r1["steps"] == steps
    # the collective of a step = the early ranges (behind the graph's external event node, beside the long encoder's backward
    # chain) + the late ranges (behind the replay), on ONE communication stream that is not the step's; every element of the
    # buffer exactly once
    if not overlap:  # the default: ONE all-reduce of the flat gradient buffer per step, behind the replay
        assert r0["allreduce"] == r1["allreduce"] == steps and r0["range_calls"] == 0
    else:
        assert r0["allreduce"] == r1["allreduce"] == 0 and r0["range_calls"] == r1["range_calls"] == 2 * steps
        assert r0["comm_streams"] == 1 and r0["streams_used"][0] != r0["main_stream"]
        assert r0["first_two"] == [r0["early"], r0["late"]] and r0["early"] and r0["late"]
        pos = 0
        for o, n_ in r0["covered"]:
            assert o == pos, "early and late ranges must tile the buffer"
            pos = o + n_
        assert pos == r0["numel"]
        # late = the LAST backward node's parameters (the convolutional encoder) + the leaf postponed into the tail of the step
        assert all(k.startswith("encoders.svhn.") or k == "decoders.mnist.layers.0.0.weight" for k in r0["late_names"]), r0["late_names"]
        assert sum(k.startswith("encoders.svhn.") for k in r0["late_names"]) == 10 and r0["late_names"] == r1["late_names"]
    assert r0["graphs"] == r1["graphs"] == 1, "the steps ran through the captured hipGraph"
    assert torch.equal(r0["params"], r1["params"]), "ranks diverged"

    # single-process replay on this GPU
    d = torch.device("cuda:0")
    model = _mnist_svhn_mopoe(d, K=K, L=L, seed=100)
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=_DP["lr"], zero_grad_in_step=True)
    ds = _dp_dataset()
    data = {m: v.to(d) for m, v in ds.data.items()}
    set_seed(_DP["seed"])
    warm = inspect.signature(GraphedStep.__init__).parameters["warmup"].default
    for _ in range(warm):
        kernels.device_randn((K, bs, L), d)
    losses = {r: [] for r in range(world)}
    for epoch in range(epochs):
        idx = [shard_indices(n, world, r).to(d) for r in range(world)]  # no set_epoch: the same permutation every epoch
        ep = {r: 0.0 for r in range(world)}
        for b in range(n // world // bs):
            eps = kernels.device_randn((K, bs, L), d)
            flat.zero_grad()
            for r in range(world):
                sel = idx[r][b * bs:(b + 1) * bs]
                out = model(DatasetOutput(data={m: v.index_select(0, sel) for m, v in data.items()}), noise=eps)
                out.loss.backward()  # accumulates into the flat buffer: sum over the ranks
                ep[r] += float(out.loss_sum)
            opt.step(grad_scale=1.0 / world)
        for r in range(world):
            losses[r].append(ep[r] / n)
    ref = flat.dense(flat.flat).cpu()
    err = float((r0["params"] - ref).abs().max())
    assert err <= 2e-6, f"distributed parameters differ from the replay by {err:.3e} after {steps} Adam steps of {_DP['lr']}"
    for r, got in ((0, r0["loss"]), (1, r1["loss"])):
        assert got == pytest.approx(losses[r], rel=1e-5), (r, got, losses[r])


def test_shipped_configuration_without_mvk_tune():
    """The suite runs under MVK_TUNE=1 (tests/conftest.py) so that single tests can pick a secondary kernel; users ship
    WITHOUT it (`mvk_tune()` returns nullptr, `_lib.tune()` the default).  The full-size reference goldens and a trainer run,
    once more in a process where the switches are off."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MVK_TUNE="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_golden.py", "tests/test_gpu_trainer.py",
                        "-k", "fullsize_golden or base_trainer_mvtcae_cfg1 or (mopoe_golden and mnistsvhn)"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
