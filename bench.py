"""Benchmark of one full training step (forward + fused ELBO + backward + Adam [+ one RCCL gradient all-reduce]) on
synthetic batches of the BASELINE.json configurations, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5                       # headline: MoPoE MnistSvhn K=10, batch 512
    python bench.py --config cfg2|cfg3k1|cfg4|cfg5 ...                    # the other BASELINE configurations
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline       the fused reconstruction-NLL kernel (HBM-bound): algorithmic bytes / device-clock duration,
  roofline_mfma  the register-stationary convolution kernels (MFMA-bound) and the whole step's GEMM FLOP rate,
  roofline_image the 3-channel image-layer kernels (HBM-bound),
  cpu_baseline   the CPU oracle (oracle/, a port of the reference path) timed on the host cores.
Kernel durations come from the library's device-timestamp profiler (mvk_prof_enable: first workgroup in to last
workgroup out on the constant-rate clock, the quantity a kernel trace reports) in eager steps of the same workload
right after the graph-replayed timed region; a HIP-event bracket of the same launch is reported beside it.
Nothing here reads /root/reference.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_TFLOPS = 157.3    # fp32-input MFMA = the fp32 vector rate
MFMA_BF16_TFLOPS = 2500.0  # dense bf16 MFMA; an fp32 product on the split-bf16 path costs 6 bf16 products
PROF_KINDS = {1: "recon_nll", 2: "imgconv_up", 3: "imgconv_down", 4: "imgconv_wgrad", 5: "image_layer_fwd",
              6: "image_layer_bwd", 7: "conv3_rs", 8: "conv3_wgrad", 9: "dense16_fwd_nll", 10: "dense16_bwd", 11: "elbo_small"}


def tracked_rocprof_avg_us(kernel):
    """(avg us, file) of `kernel` in the newest tracked profiles/rNN_kernel_stats.md (the rocprofv3 --kernel-trace --stats summary
    of this command), or None."""
    import glob
    import re

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_stats.md")), reverse=True):
        for ln in open(path):
            if ln.startswith("| ") and kernel in ln:
                cols = [c.strip() for c in ln.strip().strip("|").split("|")]
                try:
                    return float(cols[3]), os.path.relpath(path, ROOT)
                except (IndexError, ValueError):
                    break
    return None


# ---- workloads --------------------------------------------------------------------------------------------------------
def mnist_svhn_nets(L):
    from multivae_amd.models.base.base_config import BaseAEConfig
    from multivae_amd.models.nn.default_architectures import Decoder_AE_MLP, Encoder_VAE_MLP
    from multivae_amd.models.nn.svhn import Decoder_VAE_SVHN, Encoder_VAE_SVHN

    enc = dict(mnist=Encoder_VAE_MLP(BaseAEConfig(latent_dim=L, input_dim=(1, 28, 28))),
               svhn=Encoder_VAE_SVHN(BaseAEConfig(latent_dim=L, input_dim=(3, 32, 32))))
    dec = dict(mnist=Decoder_AE_MLP(BaseAEConfig(latent_dim=L, input_dim=(1, 28, 28))),
               svhn=Decoder_VAE_SVHN(BaseAEConfig(latent_dim=L, input_dim=(3, 32, 32))))
    return enc, dec


def mnist_svhn_batch(B, device, seed):
    g = torch.Generator().manual_seed(seed)
    return {"mnist": torch.rand(B, 1, 28, 28, generator=g).to(device),
            "svhn": torch.rand(B, 3, 32, 32, generator=g).to(device)}


def build_workload(name, args, device, rank):
    """-> dict(model, data, B, K, noise (callable(gen) -> noise tensor for forward, or None), adam (kwargs), text)"""
    from multivae_amd import models as M

    torch.manual_seed(0)  # default nn.Linear / nn.Conv2d initialisation
    w = dict(noise=None, adam=dict(lr=1e-3), fwd_kwargs={})
    if name in ("cfg3", "cfg3k1"):
        K = args.K if name == "cfg3" else 1
        B, L = args.batch or 512, args.latent_dim
        enc, dec = mnist_svhn_nets(L)
        cfg = M.MoPoEConfig(n_modalities=2, latent_dim=L, input_dims=dict(mnist=(1, 28, 28), svhn=(3, 32, 32)), beta=1.0, K=K)
        w.update(model=M.MoPoE(cfg, enc, dec), data=mnist_svhn_batch(B, device, rank), B=B, K=K,
                 noise=lambda gen: torch.randn(K, B, L, device=device, generator=gen),
                 metric=f"train samples/sec (ELBO step) MoPoE MnistSvhn K={K}",
                 text=f"MoPoE MnistSvhn (mnist MLP + svhn conv), K={K}, per-device batch {B}, latent_dim {L}, Adam lr 1e-3, "
                      "fwd+ELBO+bwd+optimizer")
    elif name == "cfg2":
        B, L, K = args.batch or 256, args.latent_dim, 1
        enc, dec = mnist_svhn_nets(L)
        cfg = M.MMVAEConfig(n_modalities=2, latent_dim=L, input_dims=dict(mnist=(1, 28, 28), svhn=(3, 32, 32)), K=K,
                            prior_and_posterior_dist=args.family, loss=args.loss)
        w.update(model=M.MMVAE(cfg, enc, dec), data=mnist_svhn_batch(B, device, rank), B=B, K=K,
                 metric="train samples/sec (ELBO step) MMVAE MnistSvhn K=1",
                 text=f"MMVAE MnistSvhn (mnist MLP + svhn conv), K=1, {args.family}, {args.loss}, per-device batch {B}, "
                      f"latent_dim {L}, Adam lr 1e-3, fwd+ELBO+bwd+optimizer")
    elif name == "cfg4":
        from multivae_amd.models.nn.mmnist import DecoderResnetMMNIST, EncoderResnetMMNIST

        B, K = args.batch or 32, 10  # global batch 256 = 8 GPUs x 32 (the reference example's per-device batch)
        names = [f"m{i}" for i in range(5)]
        cfg = M.MMVAEPlusConfig(n_modalities=5, latent_dim=32, input_dims={m: (3, 28, 28) for m in names}, K=K,
                                modalities_specific_dim=32, prior_and_posterior_dist="laplace_with_softmax",
                                loss="iwae_looser", beta=2.5, decoders_dist={m: "laplace" for m in names},
                                decoder_dist_params={m: dict(scale=0.75) for m in names}, learn_shared_prior=False,
                                learn_modality_prior=True)
        model = M.MMVAEPlus(cfg, {m: EncoderResnetMMNIST(32, 32) for m in names},
                            {m: DecoderResnetMMNIST(64) for m in names})
        g = torch.Generator().manual_seed(rank)
        data = {m: torch.rand(B, 3, 28, 28, generator=g).to(device) for m in names}
        w.update(model=model, data=data, B=B, K=K, adam=dict(lr=1e-3, amsgrad=True),  # examples/mmvae_plus/mmnist.py:61-62
                 metric="train samples/sec (ELBO step) MMVAE+ PolyMNIST 5 modalities K=10",
                 text=f"MMVAE+ PolyMNIST-shaped (5 x 3x28x28, ResNet enc/dec), K=10, latent 32+32, laplace_with_softmax, "
                      f"iwae_looser, beta 2.5, per-device batch {B}, Adam(amsgrad) lr 1e-3, fwd+ELBO+bwd+optimizer")
    elif name == "cfg5":
        from multivae_amd.models.base.base_config import BaseAEConfig
        from multivae_amd.models.nn.cub import CUB_Resnet_Decoder, CUB_Resnet_Encoder
        from multivae_amd.models.nn.default_architectures import Decoder_AE_MLP, Encoder_VAE_MLP

        B, L, K = args.batch or 128, 64, 1
        cfg = M.JMVAEConfig(n_modalities=2, latent_dim=L, input_dims=dict(image=(3, 64, 64), attributes=(40,)),
                            decoders_dist=dict(image="normal", attributes="bernoulli"))
        enc = dict(image=CUB_Resnet_Encoder(L), attributes=Encoder_VAE_MLP(BaseAEConfig(latent_dim=L, input_dim=(40,))))
        dec = dict(image=CUB_Resnet_Decoder(L), attributes=Decoder_AE_MLP(BaseAEConfig(latent_dim=L, input_dim=(40,))))
        g = torch.Generator().manual_seed(rank)
        data = dict(image=torch.rand(B, 3, 64, 64, generator=g).to(device),
                    attributes=(torch.rand(B, 40, generator=g) > 0.5).float().to(device))
        w.update(model=M.JMVAE(cfg, enc, dec), data=data, B=B, K=K, fwd_kwargs=dict(epoch=20),
                 metric="train samples/sec (ELBO step) JMVAE CelebA-shaped 64x64 + attributes",
                 text=f"JMVAE CelebA-shaped (3x64x64 image: CUB ResNet enc/dec, all 40 binary CelebA attributes "
                      f"(reference data/datasets/celeba.py:18; not the 18-attribute subset): MLP), latent {L}, "
                      f"per-device batch {B}, Adam lr 1e-3, fwd+ELBO+bwd+optimizer")
    else:
        raise SystemExit(f"unknown --config {name}")
    w["model"] = w["model"].to(device).train()
    w["name"], w["L"] = name, args.latent_dim
    return w


# ---- CPU baseline (oracle port of the reference path) -------------------------------------------------------------------
def oracle_loss_fn(w, B):
    """-> (state_dict of fp32 leaves, loss_fn(sd) drawing fresh noise) for the workload, on `B` samples of its batch."""
    from oracle import elbo, nets
    from oracle import train as otrain

    name, model = w["name"], w["model"]
    sd = {k: v.detach().cpu().clone().requires_grad_(v.requires_grad) for k, v in model.named_parameters()}
    data = {m: v[:B].cpu() for m, v in w["data"].items()}
    g = torch.Generator().manual_seed(1234)
    K = w["K"]
    if name in ("cfg3", "cfg3k1"):
        L = w["L"]
        shape = (K, B, L) if name == "cfg3" else (B, L)
        return sd, lambda s: otrain.mopoe_mnist_svhn_loss(s, data, torch.randn(*shape, generator=g))
    if name == "cfg2":
        L = w["L"]
        fam, loss = model.model_config.prior_and_posterior_dist, model.model_config.loss
        names = ["mnist", "svhn"]

        def f(s):
            enc, dec = nets.build_mnist_svhn(s)
            e = {m: enc[m](data[m]) for m in names}
            if fam == "laplace_with_softmax":
                noise = {m: torch.empty(K, B, L).uniform_(torch.finfo(torch.float32).eps - 1, 1, generator=g) for m in names}
            else:
                noise = {m: torch.randn(K, B, L, generator=g) for m in names}
            return elbo.mmvae_forward(e, data, dec, noise, names=names, K=K, family=fam, loss=loss,
                                      prior_log_var=s["prior_log_var"] if "prior_log_var" in s else None)
        return sd, f
    if name == "cfg4":
        names = list(data.keys())
        L = S = 32
        lo = torch.finfo(torch.float32).eps - 1

        def f(s):
            e = {m: nets.mmnist_resnet_encoder(s, f"encoders.{m}.", data[m]) for m in names}
            dec = {m: (lambda z, m=m: nets.mmnist_resnet_decoder(s, f"decoders.{m}.", z)) for m in names}
            noise = {c: dict(u=torch.empty(K, B, L).uniform_(lo, 1, generator=g), w=torch.empty(K, B, S).uniform_(lo, 1, generator=g),
                             **{r: torch.empty(K, B, S).uniform_(lo, 1, generator=g) for r in names if r != c}) for c in names}
            plv = {"shared": s["logvars_priors.shared"], **{m: s["logvars_priors." + m] for m in names}}
            return elbo.mmvaeplus_forward(e, data, dec, noise, names=names, K=K, family="laplace_with_softmax",
                                          loss="iwae_looser", beta=2.5, prior_logvars=plv,
                                          dists={m: "laplace" for m in names}, dist_scales={m: 0.75 for m in names})
        return sd, f
    if name == "cfg5":
        names = ["image", "attributes"]
        fns = dict(image=nets.cub_resnet_encoder, attributes=nets.mlp_encoder)

        def f(s):
            e = {m: fns[m](s, f"encoders.{m}.", data[m]) for m in names}
            dec = dict(image=lambda z: nets.cub_resnet_decoder(s, "decoders.image.", z),
                       attributes=lambda z: nets.mlp_decoder(s, "decoders.attributes.", z, (40,)))
            joint = nets.joint_encoder_generic(s, fns, data)
            return elbo.jmvae_forward(joint, e, data, dec, torch.randn(B, 64, generator=g), names=names, epoch=20,
                                      dists=dict(image="normal", attributes="bernoulli"))
        return sd, f
    raise ValueError(name)


def cpu_baseline(w, budget_s=20.0):
    """The oracle's training step (torch CPU, same architecture / K) on the host cores, on a bounded sample."""
    from oracle import train as otrain

    # torch's intra-op pool scales poorly on these small convolutions.  Measured on the GPU box's host (2 x EPYC
    # 9575F, 128 cores / 256 threads), samples/s by thread count: 16 -> 542, 32 -> 524, 64 -> 331, 128 -> 173,
    # 256 -> 12.  The baseline uses the best setting (16); MVK_CPU_THREADS overrides it.
    ncores = os.cpu_count() or 1
    try:
        import psutil

        ncores = psutil.cpu_count(logical=False) or ncores
    except Exception:
        pass
    ncores = max(1, min(ncores, int(os.environ.get("MVK_CPU_THREADS", "16"))))
    torch.set_num_threads(ncores)
    B = w["B"]
    Bc = B if w["name"] in ("cfg3", "cfg3k1", "cfg2") else min(B, 8)  # ResNet configurations: a reduced batch
    sd, loss_fn = oracle_loss_fn(w, Bc)
    st = otrain.AdamState(sd)

    def step():
        otrain.train_step(sd, st, loss_fn, lr=1e-3)

    step()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 8:
            break
    note = "" if Bc == B else f" at a reduced batch of {Bc} (per-sample cost taken as batch-independent)"
    return {"value": round(n * Bc / el, 2), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full training steps (B={Bc}, K={w['K']}) of the oracle after 1 warm-up step, {el:.1f}s{note}"}


# ---- device-timestamp profiler ----------------------------------------------------------------------------------------
class DeviceProfiler:
    """Records of the library's device-timestamp profiler: one per instrumented launch (eager) or per captured launch
    (hipGraph: a record accumulates one duration per replay)."""

    def __init__(self, device, nslots=2048):
        from multivae_amd import _lib

        self.lib = _lib.load()
        self.n = nslots
        self.slots = torch.zeros((nslots, 520), dtype=torch.int64, device=device)  # MVK_PROF_SLOT_U64 (include/mvk.h)
        self.slots[:, 8:264:8] = -1  # 0xFFFF... as unsigned: the armed start stamps
        self.kinds = (ctypes.c_int32 * nslots)()
        self.work = (ctypes.c_double * nslots)()
        self.khz = self.lib.mvk_prof_clock_khz()
        self.count = 0

    def start(self):
        torch.cuda.synchronize()
        self.lib.mvk_prof_enable(ctypes.c_void_p(self.slots.data_ptr()), self.n, ctypes.cast(self.kinds, ctypes.c_void_p),
                                 ctypes.cast(self.work, ctypes.c_void_p))
        self.active = True

    def stop(self):
        torch.cuda.synchronize()
        if self.active:
            self.count = self.lib.mvk_prof_count()
            self.lib.mvk_prof_enable(None, 0, None, None)
            self.active = False

    def boundary_seconds(self, n=64):
        """The dependent-kernel boundary on this stream (median gap between back-to-back one-wave kernels)."""
        from multivae_amd._lib import stream_ptr

        ticks = torch.zeros((n, 2), dtype=torch.int64, device=self.slots.device)
        torch.cuda._sleep(200_000)  # let the host enqueue all of them behind a short spin
        self.lib.mvk_prof_calibrate(ctypes.c_void_p(ticks.data_ptr()), n, stream_ptr())
        torch.cuda.synchronize()
        d = (ticks[1:, 0] - ticks[:-1, 1]).double()  # last instruction of one -> first instruction of the next
        return float(d.median()) / (self.khz * 1e3) if self.khz > 0 else 0.0

    def reset_sums(self):
        """Forget what the records accumulated so far (warm-up launches); they stay armed."""
        torch.cuda.synchronize()
        self.slots[:, :3] = 0

    def records(self):
        """[(kind name, work per launch, launches, total seconds (first workgroup in -> the launch has retired),
        total seconds (first workgroup in -> last workgroup out))]"""
        n = self.count
        s = self.slots[:n, :3].cpu()
        out = []
        for i in range(n):
            cnt = int(s[i, 1])
            if cnt and self.khz > 0:
                out.append((PROF_KINDS.get(self.kinds[i], str(self.kinds[i])), self.work[i], cnt,
                            int(s[i, 2]) / (self.khz * 1e3), int(s[i, 0]) / (self.khz * 1e3)))
        return out


def measured_copy_gbs(device, mb=512, reps=20, streaming=True):
    """Device-to-device copy bandwidth (read + write bytes per second) of a `mb`-MB fp32 buffer, HIP events around `reps`
    copies: the measured denominator SURVEY.md section 8(d) asks for beside the 8.0 TB/s nominal one.  streaming=True: the
    library's float4 nontemporal copy kernel (mvk_probe_stream_copy, the shape of kernel MI355X_MICROARCH.md quotes 6.29 TB/s
    for); False: torch's copy_ (the probe of rounds 1-4: 4.7-5.5 TB/s, not a ceiling)."""
    from multivae_amd import _lib

    n = mb * (1 << 20) // 4
    src = torch.empty(n, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)

    def copy():
        if streaming:
            _lib.call("mvk_probe_stream_copy", _lib.ptr(dst), _lib.ptr(src), n, _lib.stream_ptr())
        else:
            dst.copy_(src)

    for _ in range(3):
        copy()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        copy()
    e1.record()
    torch.cuda.synchronize()
    if streaming and not torch.equal(dst[-4096:], src[-4096:]):
        raise RuntimeError("mvk_probe_stream_copy did not copy")
    return 2.0 * 4 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def measured_mfma_tflops(device, target_us, random_operands=True, f16=False):
    """bf16 MFMA rate (TFLOP/s) a register-only loop sustains in launches of about `target_us` (mvk_probe_mfma_bf16: the pipe is
    100 % busy; what varies is the clock the chip holds under that load) — the measured ceiling beside the data-sheet one."""
    from multivae_amd import _lib

    out = torch.empty(65536, dtype=torch.float32, device=device)
    per_iter = 256 * 4 * 64 * 32768.0  # FLOP of one loop iteration of the whole launch
    iters = max(8, int(target_us * 1e-6 * 1.6e15 / per_iter))

    def run(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _lib.call("mvk_probe_mfma_bf16", _lib.ptr(out), iters, (2 if f16 else 1) if random_operands else 0, _lib.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    run(5)
    sec = run(20)
    return per_iter * iters / sec / 1e12, sec * 1e6


def summarise(recs, kinds, boundary_s=0.0):
    """Duration per launch = first workgroup in -> first instruction of the one-wave kernel queued behind it (what
    rocprofv3's begin -> end of the dispatch shows, within ~1 %); `busy_avg_us` takes one calibrated kernel boundary off."""
    sel = [r for r in recs if r[0] in kinds]
    if not sel:
        return None
    n = sum(r[2] for r in sel)
    work, outer, inner = sum(r[1] * r[2] for r in sel), sum(r[3] for r in sel), sum(r[4] for r in sel)
    busy = max(outer - n * min(boundary_s, 2.5e-6), inner)
    return dict(launches=n, work=work, seconds=outer, avg_us=1e6 * outer / n, inner_avg_us=1e6 * inner / n,
                busy_avg_us=1e6 * busy / n)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute this command line under torch.distributed.run with N ranks on
    this node (rendezvous on 127.0.0.1, a free port) and return its exit status.  Rank 0 of the child job prints the line."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MVK_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without a launcher environment: starting {n} ranks ({' '.join(cmd[1:9])} ...)", file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: WORLD_SIZE when a launcher started the process, else 1)")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg3k1", "cfg4", "cfg5"],
                    help="BASELINE.json configuration (cfg3 = configs[2], the headline; cfg3k1 = the same with the "
                         "reference's single sample)")
    ap.add_argument("--batch", type=int, default=0, help="per-device batch (default: the configuration's)")
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--latent-dim", type=int, default=20)
    ap.add_argument("--family", default="normal", choices=["normal", "laplace_with_softmax"], help="cfg2 posterior family")
    ap.add_argument("--loss", default="iwae_looser", choices=["iwae_looser", "dreg_looser"], help="cfg2 objective")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every launch from Python instead of replaying a hipGraph")
    ap.add_argument("--rotate", action="store_true",
                    help="single GPU: the decoders' late weight gradients, their finishes and their share of Adam run at the head of "
                         "the NEXT replay (GraphedStep(rotate=...); exact, bit-identical parameters; the timed region ends with the "
                         "drain).  Built for VERDICT r5 item 1 and measured SLOWER on one MI355X (profiles/NOTES_r06.md section 2): opt-in")
    ap.add_argument("--no-rotate", action="store_true", help=argparse.SUPPRESS)  # (the default; kept for the round's A/B scripts)
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    # One process per GPU, launched from env as the reference's trainer expects (trainers/base/base_trainer_config.py:80-100,
    # examples/distributed_training.py:56-71).  `--gpus N` is a CONTRACT, not a hint (VERDICT r4 missing #1):
    #   no launcher environment and N > 1 -> this process becomes the launcher (torch.distributed.run, N ranks on 127.0.0.1);
    #   a launcher environment whose WORLD_SIZE differs from N -> exit status 2, nothing printed on stdout;
    #   the line's n_gpus is what the gradient collective's communicator reports (ncclCommCount), checked against N.
    if args.gpus is None:  # `torchrun --nproc-per-node N bench.py` without --gpus reports the launcher's size
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a number for a "
              "job of another size", file=sys.stderr)
        raise SystemExit(2)
    local_rank = 0 if os.environ.get("MVK_BENCH_SAME_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: multivae_amd has no CPU compute path")
    if local_rank >= torch.cuda.device_count():
        print(f"[bench] rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        raise SystemExit(2)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist

    use_dist = world > 1 or os.environ.get("MVK_FORCE_DIST") == "1"  # the latter exercises the RCCL path on 1 GPU
    result_fd = 1
    if use_dist:
        # RCCL prints a version banner ("RCCL version : ...", "Librccl path : ...") on STDOUT when a communicator comes and goes:
        # everything the libraries write to fd 1 is sent to stderr, the ONE JSON line goes to the real stdout at the very end
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)
        # nccl == RCCL on ROCm.  MVK_DIST_BACKEND=gloo: the control flow of the multi-rank run on ONE GPU (RCCL refuses two
        # ranks on one device; tests/test_gpu_trainer.py runs bench.py that way with MVK_BENCH_SAME_GPU=1)
        dist.init_process_group(os.environ.get("MVK_DIST_BACKEND", "nccl"), init_method="env://", world_size=world, rank=rank)

    from multivae_amd import _lib, kernels
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.trainers import FlatParams, FusedAdam

    w = build_workload(args.config, args, device, rank)
    model, B, K = w["model"], w["B"], w["K"]
    flat = FlatParams(model)
    if use_dist:
        flat.broadcast(0)  # C1: one parameter broadcast
    # the update clears the gradients it consumes (MVK_ADAM_ZERO=0: a separate zero_grad pass per step, for A/B)
    opt = FusedAdam(flat, zero_grad_in_step=os.environ.get("MVK_ADAM_ZERO", "1") != "0", **w["adam"])
    inputs = DatasetOutput(data=w["data"])
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    torch.cuda.manual_seed(1000 + rank)  # the model draws its noise from the device generator: another stream per rank
    grad_scale = 1.0 / world
    draw = w["noise"]
    fkw = w["fwd_kwargs"]

    def eager_step(collective=True):
        kw = dict(fkw)
        if draw is not None:
            kw["noise"] = draw(gen)
        opt.zero_grad()
        with kernels.deferred_reductions(flat):
            out = model(inputs, **kw)
            out.loss.backward(gradient=kernels.unit_seed(out.loss))
        scale = 1.0
        if use_dist and collective:
            scale = flat.all_reduce_mean()  # C2: ONE collective over the flat gradient buffer (mvk_allreduce_avg on RCCL)
        opt.step(grad_scale=scale if (use_dist and collective) else grad_scale)
        return out

    # forward + backward replayed as ONE hipGraph launch (the host needs about as long to enqueue the
    # launches of a step as the GPU needs to run them); all-reduce and the fused Adam launch stay outside.
    graphed = None
    if not args.no_graph:
        from multivae_amd.trainers import GraphedStep

        try:
            # noise=None: the model draws its reparameterisation noise itself, inside the captured graph (as it does under
            # BaseTrainer and in the reference's forward), instead of a host-side draw + one more copy per step
            # Two forms of the step that were built, tested and MEASURED slower on one GPU stay opt-in:
            #   MVK_GRAPH_ADAM=1  the fused Adam as the graph's last node (device-resident step / lr): +7 ... +10 us per step against
            #                     the host-scalar launch behind the replay (four same-box pairs, profiles/r05_bench_lines.jsonl);
            #   MVK_OVERLAP=1     the gradient collective in two parts, the first behind an external event node of the graph: the
            #                     node alone costs the replayed step +65 us on one GPU (hipGraph re-partitions its queues around
            #                     it), +100 us with the second collective launch.
            graph_adam = not use_dist and os.environ.get("MVK_GRAPH_ADAM", "0") == "1"
            # Rotated step (single GPU, --rotate): the decoders' late weight gradients of step N, their finishes and their share of
            # the update run at the head of replay N + 1, beside the encoders (exact: tests/test_gpu_trainer.py
            # test_rotated_step_is_bit_identical); the timed region below ends with the drain of the last step.
            rotate = (not use_dist and not graph_adam and args.rotate and not args.no_rotate and opt.zero_grad_in_step)
            graphed = GraphedStep(model, flat, inputs, noise=None,
                                  capture_error_mode="thread_local" if use_dist else "global",
                                  optimizer=opt if graph_adam else None, rotate=opt if rotate else None,
                                  overlap=use_dist and os.environ.get("MVK_OVERLAP", "0") in ("1", "2"), **fkw)
        except Exception as e:  # capture is an optimisation, not a requirement
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphed = None

    def graph_step():
        out = graphed(inputs)  # copies the batch into the captured buffers, replays
        if use_dist:
            graphed.reduce_and_step(opt)  # C2: the gradient collective (early ranges beside the end of the backward pass), Adam
        elif not graphed.includes_optimizer:
            opt.step(grad_scale=1.0)
        return out

    step = graph_step if graphed is not None else eager_step
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    # Settling: the clock ramp of a freshly started process belongs outside the timed region by
    # construction, not by the choice of --warmup.  Untimed groups of SETTLE_GROUP steps (device time of a group between two
    # events) follow the W warm-up steps until two consecutive groups agree within 3 %, at most SETTLE_MAX groups; every rank
    # runs the same number (the ranks agree on "settled" through an all-reduce), and the count goes into the line.
    SETTLE_GROUP, SETTLE_MAX = 5, 10
    settle_groups, prev_ms = 0, None
    while settle_groups < SETTLE_MAX:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(SETTLE_GROUP):
            out = step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / SETTLE_GROUP
        settle_groups += 1
        settled = prev_ms is not None and abs(ms - prev_ms) <= 0.03 * min(ms, prev_ms)
        if use_dist:
            flag = torch.tensor([1 if settled else 0], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            settled = bool(int(flag.item()))
        prev_ms = ms
        if settled:
            break
    if graphed is not None:
        graphed.drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # per-step device times inside the timed region: one event record per step on the step's stream (no sync, ~1 us of
    # host time each); the median is reported beside the mean that `value` is computed from
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out = step()
        if i + 1 == args.steps and graphed is not None:
            graphed.drain()  # rotated step: the last step's late leaves + their update belong to the timed region
        marks[i + 1].record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    loss = float(out.loss.detach())
    if loss != loss:
        raise ArithmeticError("NaN detected in train loss")

    recs, events, step_flops, ms_instr, boundary = [], [], 0.0, None, 0.0
    if rank == 0:
        # Kernel durations: the same step again with the library's device-timestamp records switched on (captured into a
        # second graph: every replay accumulates into the records), `steps` replays right after the timed region.  The
        # stamps and the one-wave fold kernels behind the instrumented launches cost ~2 % of the step, which is why the
        # headline number above is measured without them.
        # (rank 0 only: NO collective in here — the other ranks are already waiting in the all-reduce of the elapsed time below)
        prof = DeviceProfiler(device)
        prof.start()
        step2 = lambda: eager_step(collective=False)
        if graphed is not None:
            try:
                g2 = GraphedStep(model, flat, inputs, noise=None, rotate=opt if (graphed.rotated) else None,
                                 capture_error_mode="thread_local" if use_dist else "global", **fkw)
                prof.stop()  # the captured launches keep their records; nothing else is stamped from here on

                def step2():
                    o = g2(inputs)
                    opt.step(grad_scale=grad_scale)
                    return o
            except Exception as e:
                print(f"[bench] instrumented capture failed ({type(e).__name__}: {e}); profiling eagerly", file=sys.stderr)
        for _ in range(2):
            step2()
        prof.reset_sums()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step2()
        if graphed is not None and graphed.rotated:
            g2.drain()
        torch.cuda.synchronize()
        ms_instr = 1e3 * (time.perf_counter() - t1) / args.steps
        prof.stop()
        recs = prof.records()
        boundary = prof.boundary_seconds()
        # beside it: the HIP-event bracket of the NLL launch in eager steps, and the GEMM FLOP count of one step
        kernels.PROFILE["recon_nll"] = events
        kernels.PROFILE["presleep_cycles"] = 200_000  # ~0.1 ms GPU spin: the host enqueues the launch meanwhile
        for _ in range(3):
            eager_step(collective=False)
        torch.cuda.synchronize()
        kernels.PROFILE.pop("recon_nll", None)
        kernels.PROFILE.pop("presleep_cycles", None)
        _lib.COUNT_FLOPS = [0.0]  # GEMM-shaped FLOP of one step (every mvk_linear / gemm / conv entry point)
        eager_step(collective=False)
        torch.cuda.synchronize()
        step_flops, _lib.COUNT_FLOPS = _lib.COUNT_FLOPS[0], None

    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # the size of the job = the number of ranks the gradient collective ran over, asked of the communicator itself
    n_ranks = flat.world_size() if use_dist else 1
    if n_ranks != args.gpus:
        print(f"[bench] the gradient collective ran over {n_ranks} rank(s), --gpus says {args.gpus}", file=sys.stderr)
        raise SystemExit(3)

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        res = {
            "metric": w["metric"],
            "value": round(n_ranks * B * args.steps / elapsed, 2),
            "unit": "samples/s",
            "n_gpus": n_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_groups * SETTLE_GROUP,  # untimed, behind the warm-up: groups of 5 until two agree within 3 % (<= 50)
            "ms_per_step": round(ms, 4),
            "ms_per_step_median": round(per_step_ms[len(per_step_ms) // 2], 4),
            "ms_per_step_min_max": [round(per_step_ms[0], 4), round(per_step_ms[-1], 4)],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (error-corrected 2xfp16 / 3xbf16 products, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": w["text"] + (", 1 RCCL all-reduce/step" if n_ranks > 1 else ""),
                       "baseline_config": args.config, "global_batch": n_ranks * B, "K": K, "parallelism": f"dp{n_ranks}",
                       "collective": (("mvk_allreduce_avg (RCCL, ncclCommCount = %d)" % n_ranks) if getattr(flat, "_comm", None)
                                      else ("torch.distributed all_reduce (%s)" % dist.get_backend())) if use_dist else None,
                       "final_loss": round(loss, 4),
                       "peak_device_memory_gb": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
                       "launch": ("eager" if graphed is None else
                                  "ONE hipGraph replay (fwd + bwd + Adam with device-resident step / lr)" if graphed.includes_optimizer else
                                  ("hipGraph replay (fwd + bwd) || all-reduce of %d early + %d late ranges (%.1f + %.1f MB) + Adam" % (
                                      len(graphed.early_ranges), len(graphed.late_ranges),
                                      4e-6 * sum(n for _, n in graphed.early_ranges), 4e-6 * sum(n for _, n in graphed.late_ranges))
                                   if graphed.early_ranges else ("hipGraph replay (fwd + bwd) + mvk_allreduce_avg + Adam" if use_dist else
                                                                  ("ROTATED hipGraph replay: [late weight gradients of the previous step + "
                                                                   "their finishes + Adam over %.2f MB of decoder weights + their packs] beside "
                                                                   "[encoders + posterior], then decoders + backward; Adam over the rest behind "
                                                                   "the replay; the timed region ends with the drain of the last step"
                                                                   % (4e-6 * sum(n for _, n in opt._rot_ranges)))
                                                                  if graphed.rotated else "hipGraph replay (fwd + bwd) + Adam"))),
                       # how the fp32 GEMM / convolution products are formed (operands, results and accumulation are fp32; the
                       # tests hold every form to the same float64-referenced tolerance)
                       "fp32_product": ("register-stationary convolutions: 3 fp16 MFMAs on scaled (hi, lo) pairs"
                                        if (kernels.IMG_F16 or kernels.C3_F16) else "6 bf16 MFMAs on three pieces")
                                       + ("; MLP decoder at the decoder batch: 3 fp16 MFMAs on pre-split (hi, lo) planes"
                                          if kernels.DENSE16 else "")
                                       + "; tiled engine: 6 bf16 MFMAs on three pieces; latency-sized layers: exact fp32 FMA / MFMA"},
        }
        nll_generic = summarise(recs, {"recon_nll"}, boundary)
        imgf = summarise(recs, {"image_layer_fwd"}, boundary)
        # With the fused decoder tail (round 3) the large modality's reconstruction NLL lives in the epilogue of its decoder's last
        # layer: THAT launch is the fused reconstruction-NLL kernel now, the generic kernel only scores the small modality
        # (round 4: the small modality's MLP decoder scores itself too — no generic launch is left in the headline step)
        fused_tail = bool(imgf and imgf["work"] / imgf["launches"] > 2.0e8 and
                          (nll_generic is None or nll_generic["work"] / nll_generic["launches"] < 0.5 * imgf["work"] / imgf["launches"]))
        nll = imgf if fused_tail else nll_generic
        if nll:
            ev = [s.elapsed_time(e) * 1e-3 for s, e in events]
            traffic, traffic_meta = None, {}
            tf = os.path.join(ROOT, "profiles", "fused_tail_traffic.json" if fused_tail else "recon_nll_traffic.json")
            if args.config == "cfg3" and os.path.exists(tf):
                with open(tf) as f:
                    traffic_meta = json.load(f)
                traffic = traffic_meta.get("hbm_bytes_per_launch")
            # the average of the SAME kernel in the tracked rocprofv3 summary of this command (profiles/: written by
            # tools/make_profiles.py from `rocprofv3 --kernel-trace --stats -- python bench.py`): the second denominator
            tail_kernel = "small_up_fwd_h_kernel<3, 512, true>" if (kernels.TAIL_F16 and kernels.IMG_F16) else "small_up_fwd_bf_kernel<3, 512, true>"
            tracked_us = (tracked_rocprof_avg_us(tail_kernel) or tracked_rocprof_avg_us("small_up_fwd_bf_kernel<3, 512, true>")) if fused_tail \
                else tracked_rocprof_avg_us("recon_nll_kernel<1, true>")
            tracked_us = tracked_us if args.config == "cfg3" else None
            ach = nll["work"] / nll["seconds"] / 1e9
            copy_gbs = measured_copy_gbs(device)
            res["roofline"] = {
                "kernel": (tail_kernel + " (decoder tail + reconstruction NLL + d NLL / d pre-activation of the "
                           "svhn modality in one launch: reads the 16x16x32 input map and the targets, writes the gradient; the "
                           "image and d_recon of SURVEY 8(d)'s byte count are never written)") if fused_tail else
                          "recon_nll_kernel<vec,fwd> (fused reconstruction NLL + d_recon, all modalities, one launch)",
                "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                "frac_is": "algorithmic bytes / (avg_launch_us measured in THIS run with device timestamps) / peak",
                "tracked_rocprof_avg_us": tracked_us[0] if tracked_us else None,
                "tracked_rocprof_file": tracked_us[1] if tracked_us else None,
                "frac_tracked_rocprof": round(nll["work"] / nll["launches"] / (tracked_us[0] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if tracked_us else None,
                "traffic_source": (("profiles/fused_tail_traffic.json" if fused_tail else "profiles/recon_nll_traffic.json") + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                   "command by tools/gpu_profile.sh (counters cannot be read from inside the process); "
                                   f"measured at commit {traffic_meta.get('commit', 'unrecorded')}") if traffic else None,
                "measured_copy_GBs": round(copy_gbs, 1), "frac_vs_measured_copy": round(ach / copy_gbs, 4),
                "measured_copy_is": "float4 nontemporal streaming copy of 512 MB (mvk_probe_stream_copy), read + write bytes / time, "
                                    "HIP events over 20 launches in this run; torch copy_ of the same buffers: "
                                    f"{round(measured_copy_gbs(device, streaming=False), 1)} GB/s",
                "algorithmic_bytes": nll["work"] / nll["launches"], "avg_launch_us": round(nll["avg_us"], 2),
                "launches_timed": nll["launches"],
                "first_in_last_out_us": round(nll["inner_avg_us"], 2),
                "minus_kernel_boundary_us": round(nll["busy_avg_us"], 2), "kernel_boundary_us": round(1e6 * boundary, 2),
                "frac_minus_kernel_boundary": round(nll["work"] / nll["launches"] / nll["busy_avg_us"] / 1e3 / HBM_PEAK_GBS, 4),
                "method": "device timestamps (mvk_prof_enable, constant-rate clock): first workgroup in -> first "
                          "instruction of the one-wave kernel queued behind the launch (its stores have drained) = the "
                          "begin -> end rocprofv3 reports for the dispatch (profiles/r02_kernel_stats.md); accumulated over "
                          "`steps` replays of the instrumented step right after the timed region.  minus_kernel_boundary_us "
                          "takes off one dependent-kernel boundary calibrated in the same run",
                "instrumented_ms_per_step": round(ms_instr, 4),
                "hip_event_pair_us": round(1e6 * sum(ev) / len(ev), 2) if ev and not fused_tail else None}
            if fused_tail and nll_generic:  # the generic likelihood kernel that is left (a modality without a fused tail)
                g_ach = nll_generic["work"] / nll_generic["seconds"] / 1e9
                res["roofline_generic_nll"] = {"kernel": "recon_nll_kernel<vec,fwd> (mnist only)", "bound": "hbm",
                                               "achieved": round(g_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                               "frac": round(g_ach / HBM_PEAK_GBS, 4),
                                               "algorithmic_bytes": nll_generic["work"] / nll_generic["launches"],
                                               "avg_launch_us": round(nll_generic["avg_us"], 2)}
        # The ELBO group of north_star ("fused MoPoE ELBO"): every launch between the decoders' last hidden activation and the
        # gradient of their pre-activations + the posterior / KL kernels + the scalar assembly — minimal bytes of the group /
        # the sum of its in-step launch durations (the members overlap other streams: a lower bound of what each achieves)
        d16f = summarise(recs, {"dense16_fwd_nll"}, boundary)
        small = summarise(recs, {"elbo_small"}, boundary)
        members = [m for m in (imgf if fused_tail else None, nll_generic, d16f, small) if m]
        if members and "roofline" in res:
            gb = sum(m["work"] / m["launches"] * (m["launches"] / args.steps) for m in members)  # bytes per step
            gus = sum(1e6 * m["seconds"] / args.steps for m in members)
            # SURVEY section 8(d)'s algorithmic bytes of the fused ELBO: 4 [B D (2K + 1) + B L (4M + 3K)] — recon read, x read,
            # d_recon written, posterior / reparameterisation traffic (cfg3: 167.4 MB per step)
            D_all = sum(int(v[0].numel()) for v in w["data"].values())
            survey_bytes = 4.0 * (B * D_all * (2 * K + 1) + B * w["L"] * (4 * len(w["data"]) + 3 * K))
            res["roofline"]["elbo_group"] = {
                "survey_8d_bytes_per_step": round(survey_bytes), "frac_vs_survey_8d_bytes": round(survey_bytes / gus / 1e3 / HBM_PEAK_GBS, 4),
                "bound": "mixed: the svhn tail is HBM-bound (the `roofline` object above), the mnist tail lives in the epilogue of a "
                         "GEMM launch (dense16 fwd_nll, MFMA work: roofline_mfma.dense16_*), the posterior kernels and the assembly "
                         "are latency-sized — the group's fraction says how little of the ELBO's wall time is bandwidth, it is not "
                         "a roofline of any one kernel",
                "members": "fused svhn tail" + (" + generic recon_nll" if nll_generic else "") + (" + dense16 fwd_nll (mnist tail)" if d16f else "")
                           + (" + mopoe_posterior fwd / bwd + reduce_terms" if small else ""),
                "bytes_per_step": round(gb), "us_per_step": round(gus, 1), "achieved": round(gb / gus / 1e3, 1), "unit": "GB/s",
                "frac": round(gb / gus / 1e3 / HBM_PEAK_GBS, 4),
                "launches_per_step": round(sum(m["launches"] for m in members) / args.steps, 1)}
        d16b = summarise(recs, {"dense16_bwd"}, boundary)
        conv = summarise(recs, {"imgconv_up", "imgconv_down", "imgconv_wgrad", "conv3_rs", "conv3_wgrad"}, boundary)
        conv3 = summarise(recs, {"conv3_rs", "conv3_wgrad"}, boundary) is not None
        SPLIT_PEAK = MFMA_BF16_TFLOPS / 6
        mf = {"bound": "mfma", "unit": "TFLOP/s", "peak": round(SPLIT_PEAK, 1),
              "step_gemm_gflop": round(step_flops / 1e9, 2),
              "step_achieved": round(step_flops / (ms * 1e-3) / 1e12, 2),
              "step_frac": round(step_flops / (ms * 1e-3) / 1e12 / SPLIT_PEAK, 4),
              "step_frac_vs_3mfma_peak": round(step_flops / (ms * 1e-3) / 1e12 / (MFMA_BF16_TFLOPS / 3), 4),
              "note": "peak = 2500 / 6 TFLOP/s of fp32 work for products formed from 6 bf16 MFMAs (tiled engine), 2500 / 3 for the "
                      "3-fp16-MFMA form the large launches of the step run (kernel_peak, step_frac_vs_3mfma_peak: the denominator "
                      "for a step whose FLOPs are almost all on that form); step_* = every GEMM-shaped FLOP of the step / ms_per_step; "
                      "achieved / us_per_step of the named kernels are IN-STEP durations: the launches share the chip with "
                      "the other modality's stream and the late weight gradients (alone: 98-113 us per launch, "
                      "tools/imgconv_probe.py, DESIGN.md section 6)"}
        if conv:
            ach = conv["work"] / conv["seconds"] / 1e12
            m_tf, m_us = measured_mfma_tflops(device, 1e6 * conv["seconds"] / conv["launches"])
            # the register-stationary kernels run the scaled-fp16 form (csrc/bf3.hpp): 3 fp16 MFMAs per fp32
            # product instead of the 6 bf16 ones — their ceiling is 2500 / 3 (fp16 and bf16 MFMAs run at the same rate)
            # (the headline's six decoder launches at n = K B take the same form since the second half of round 3; the four
            # encoder launches at n = B, 1/10 of the FLOPs, still use bf16 pieces)
            prods = 3 if ((conv3 and kernels.C3_F16) or (not conv3 and kernels.IMG_F16)) else 6
            KPEAK = MFMA_BF16_TFLOPS / prods
            if prods == 3:  # the instruction the named kernels issue: the same loop on v_mfma_f32_32x32x16_f16
                h_tf, h_us = measured_mfma_tflops(device, 1e6 * conv["seconds"] / conv["launches"], f16=True)
                mf.update({"measured_sustained_f16": round(h_tf, 1), "frac_vs_measured_sustained_f16": round(ach / (h_tf / prods), 4)})
            mf.update({"kernel_mfmas_per_fp32_product": prods, "kernel_peak": round(KPEAK, 1),
                       "measured_sustained_bf16": round(m_tf, 1), "measured_sustained_fp32_equiv": round(m_tf / prods, 1),
                       "measured_launch_us": round(m_us, 1),
                       "frac_vs_measured_sustained": round(ach / (m_tf / prods), 4),
                       "measured_note": "register-only v_mfma_f32_32x32x16_bf16 loop with hashed operand values in launches of the "
                                        "named kernels' length (mvk_probe_mfma_bf16): pipe 100 % busy, clock as sustained under "
                                        "that load — the power-limited ceiling of this chip for real operand data"})
            mf.update({"kernel": ("c3rs_kernel / c3wg_kernel (register-stationary 3x3 convolutions of the ResNet blocks)" if conv3 else
                                  "imgconv_kernel / imgwgrad_kernel (register-stationary 4x4/stride-2 convolutions)"),
                       "achieved": round(ach, 1), "frac": round(ach / KPEAK, 4),
                       "gflop_per_step": round(conv["work"] / args.steps / 1e9, 2),
                       "us_per_step": round(1e6 * conv["seconds"] / args.steps, 1), "launches_timed": conv["launches"]})
        if d16b:  # the MLP decoder's backward GEMMs on pre-split planes (3 fp16 MFMAs per product; the forward one is in elbo_group)
            mf["dense16_bwd"] = {"kernel": "d16_nt_kernel<64, BWD> + d16_tn_kernel (MLP decoder backward data / weight gradient on fp16 pair planes)",
                                 "achieved": round(d16b["work"] / d16b["seconds"] / 1e12, 1), "peak": round(MFMA_BF16_TFLOPS / 3, 1),
                                 "frac": round(d16b["work"] / d16b["seconds"] / 1e12 / (MFMA_BF16_TFLOPS / 3), 4),
                                 "us_per_step": round(1e6 * d16b["seconds"] / args.steps, 1)}
        res["roofline_mfma"] = mf
        img = {k: summarise(recs, {k}, boundary) for k in ("image_layer_fwd", "image_layer_bwd")}
        if any(img.values()):
            res["roofline_image"] = {k: {"bound": "hbm", "achieved": round(v["work"] / v["seconds"] / 1e9, 1),
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": round(v["work"] / v["seconds"] / 1e9 / HBM_PEAK_GBS, 4),
                                         "avg_launch_us": round(v["avg_us"], 2)} for k, v in img.items() if v}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(w, args.cpu_budget)
        result_line = json.dumps(res)
    else:
        result_line = None
    if use_dist:
        flat.close()  # mvk_comm_destroy: the RCCL communicator goes before the process group it was built through
        dist.destroy_process_group()
    sys.stdout.flush()
    if result_line is not None:
        os.write(result_fd, (result_line + "\n").encode())


if __name__ == "__main__":
    main()
