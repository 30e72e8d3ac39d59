"""CPU tests of the host side: the drop-in surface (names, arguments, error behaviour of the reference's plugin
API), the C-ABI library (loads, exports every symbol include/mvk.h declares), and that the product path refuses
to compute on the CPU (no fallback)."""
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from multivae_amd import _lib

    header = open(os.path.join(ROOT, "include", "mvk.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mvk_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 35
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mvk.h but not exported by libmvk.so"
    bound = set(_lib.PROTOTYPES) | {"mvk_splitk_workspace_floats", "mvk_conv4s2_small_up_supported", "mvk_conv4s2_small_up_nll_supported",
                                    "mvk_imgconv_frag_bytes",
                                    "mvk_debug_set_phase_buffer", "mvk_debug_set_flags", "mvk_dense16_debug", "mvk_dense16_debug_stamps",  # void hooks, bound ad hoc
                                    "mvk_dense16_ok", "mvk_dense16_fwd_nll_rows", "mvk_dense16_colsum_rows", "mvk_comm_id_bytes", "mvk_comm_available",
                                    "mvk_prof_enable", "mvk_prof_count", "mvk_prof_clock_khz", "mvk_prof_calibrate",
                                    "mvk_defer_pending", "mvk_defer_wanted", "mvk_conv3x3_fused_ok", "mvk_conv3x3_scaled_ok", "mvk_conv3x3_wgrad_scaled_ok", "mvk_conv4s2_scaled_ok", "mvk_conv4s2_wgrad_scaled_ok"}
    assert declared == bound, (declared - bound, bound - declared)
    assert lib.mvk_version() >= 100


def test_argument_counts_match_header():
    from multivae_amd import _lib

    header = open(os.path.join(ROOT, "include", "mvk.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    for name, argtypes in _lib.PROTOTYPES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", header, flags=re.S)
        assert m, name
        args = m.group(1).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert n == len(argtypes), (name, n, len(argtypes))


def _header_struct_layouts(tmp_path):
    """{struct name: (sizeof, [(field, offset, size), ...])} of every `typedef struct` in include/mvk.h, as gcc lays them out
    (the header is compiled as C: the host ABI of hipcc's x86-64 side is the same)."""
    import subprocess

    text = open(os.path.join(ROOT, "include", "mvk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if decl:
                fields += [re.sub(r"\[.*\]", "", f).strip().lstrip("*").strip().split()[-1].lstrip("*") for f in decl.split(",")]
        structs[m.group(3)] = fields
    assert structs, "no structs parsed"
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "mvk.h")}"', "int main(void){"]
    for name, fields in structs.items():
        src.append(f'printf("S {name} %zu\\n", sizeof({name}));')
        for f in fields:
            src.append(f'printf("F {name} {f} %zu %zu\\n", offsetof({name}, {f}), sizeof((({name}*)0)->{f}));')
    src.append("return 0;}")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(c)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    res = {}
    for ln in out.splitlines():
        t = ln.split()
        if t[0] == "S":
            res[t[1]] = (int(t[2]), [])
        else:
            res[t[1]][1].append((t[2], int(t[3]), int(t[4])))
    return res


def _assert_layout(cls, want, what):
    import ctypes as C

    size, fields = want
    assert C.sizeof(cls) == size, (what, C.sizeof(cls), size)
    got = [(n, getattr(cls, n).offset, getattr(cls, n).size) for n, _ in cls._fields_]
    assert got == fields, (what, got, fields)


def integration_md_python():
    """The Python code blocks of INTEGRATION.md section B (the binding a maintainer of the reference would add), joined."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    text = text[text.index("## B."):]
    return "\n\n".join(re.findall(r"```python\n(.*?)```", text, flags=re.S))


def test_struct_layouts_match_header(tmp_path):
    """Descriptor tables cross the C ABI as ARRAYS: sizeof and every field offset of the ctypes mirrors — the package's and
    the ones INTEGRATION.md documents — must be what a C compiler makes of include/mvk.h (the documented
    ReconDesc had lost `n_classes`, stride 72 instead of 80)."""
    from multivae_amd import _lib

    want = _header_struct_layouts(tmp_path)
    mirrors = {"mvk_recon_desc": _lib.ReconDesc, "mvk_term_desc": _lib.TermDesc, "mvk_seed_desc": _lib.SeedDesc,
               "mvk_pack_desc": _lib.PackDesc}
    assert set(want) == set(mirrors), (set(want) ^ set(mirrors))
    for name, cls in mirrors.items():
        _assert_layout(cls, want[name], f"_lib.{cls.__name__}")
    assert want["mvk_recon_desc"][0] == 80
    # the documented binding: execute the code blocks (they only define classes / functions and set prototypes)
    code = integration_md_python().replace('C.CDLL("libmvk.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    documented = {"ReconDesc": "mvk_recon_desc", "TermDesc": "mvk_term_desc", "SeedDesc": "mvk_seed_desc", "PackDesc": "mvk_pack_desc"}
    seen = [k for k in documented if k in ns]
    assert "ReconDesc" in seen
    for k in seen:
        _assert_layout(ns[k], want[documented[k]], f"INTEGRATION.md {k}")
    # and its prototypes have the header's argument counts
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mvk.h")).read(), flags=re.S)
    lib = ns["_lib"]
    for name in re.findall(r"_lib\.(mvk_\w+)\.argtypes", code) + re.findall(r'\("(mvk_\w+)",\s*\[', code):
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", header, flags=re.S)
        assert m, name
        args = m.group(1).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert len(getattr(lib, name).argtypes) == n, (name, n)


def test_bench_refuses_a_world_size_other_than_gpus():
    """bench.py --gpus N under a launcher environment of another size exits with status 2 and prints no result line (the check
    sits in front of everything that needs a GPU)."""
    import subprocess
    import sys

    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29549")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 2 and not r.stdout.strip() and "WORLD_SIZE=2" in r.stderr, (r.returncode, r.stdout, r.stderr[-500:])


def test_no_cpu_compute_path():
    from multivae_amd import _lib, kernels

    x = torch.randn(4, 3)
    w = torch.randn(5, 3)
    b = torch.zeros(5)
    with pytest.raises(_lib.MvkError):
        kernels.MLPDecoderFn.apply(x, w, b, torch.randn(2, 5), torch.zeros(2), (2,))
    from multivae_amd.data.datasets.base import DatasetOutput
    from multivae_amd.models import MoPoE, MoPoEConfig

    model = MoPoE(MoPoEConfig(n_modalities=2, latent_dim=3, input_dims=dict(a=(2,), b=(3,))))
    with pytest.raises(_lib.MvkError):
        model(DatasetOutput(data=dict(a=torch.rand(4, 2), b=torch.rand(4, 3))))


def test_model_output_and_dataset_contract():
    from multivae_amd.data.datasets.base import DatasetOutput, IncompleteDataset, MultimodalBaseDataset
    from multivae_amd.models.base.base_utils import ModelOutput

    o = ModelOutput(loss=1.0, metrics={})
    o.loss_sum = 2.0
    o["extra"] = 3
    assert (o.loss, o["loss_sum"], o.extra, o[0]) == (1.0, 2.0, 3, 1.0)
    assert list(o.keys()) == ["loss", "metrics", "loss_sum", "extra"]
    ds = MultimodalBaseDataset(data=dict(a=torch.rand(5, 2), b=torch.rand(5, 3)), labels=torch.arange(5))
    assert len(ds) == 5 and set(ds[1].data) == {"a", "b"} and int(ds[1].labels) == 1 and not hasattr(ds[1], "masks")
    with pytest.raises(AttributeError):
        len(MultimodalBaseDataset(data=dict(a=torch.rand(5, 2), b=torch.rand(4, 3))))
    inc = IncompleteDataset(data=dict(a=torch.rand(5, 2)), masks=dict(a=torch.ones(5, dtype=torch.bool)))
    assert hasattr(inc[0], "masks") and isinstance(inc[0], DatasetOutput)
    with pytest.raises(AttributeError):
        IncompleteDataset(data=dict(a=torch.rand(5, 2)), masks=dict(a=torch.ones(4, dtype=torch.bool)))


def test_config_json_round_trip(tmp_path):
    from multivae_amd.models import MMVAEConfig, MoPoEConfig, MVTCAEConfig

    c = MoPoEConfig(n_modalities=2, latent_dim=20, input_dims=dict(mnist=[1, 28, 28], svhn=[3, 32, 32]), beta=2.5)
    assert c.name == "MoPoEConfig" and c.input_dims["mnist"] == (1, 28, 28)
    c.save_json(str(tmp_path), "model_config")
    d = json.load(open(tmp_path / "model_config.json"))
    assert d["name"] == "MoPoEConfig" and d["beta"] == 2.5 and d["K"] == 1
    c2 = MoPoEConfig.from_json_file(str(tmp_path / "model_config.json"))
    assert c2.to_dict() == c.to_dict()
    assert MVTCAEConfig(n_modalities=2).alpha == 0.1 and MVTCAEConfig(n_modalities=2).beta == 2.5
    m = MMVAEConfig(n_modalities=2)
    assert (m.K, m.prior_and_posterior_dist, m.loss, m.learn_prior) == (10, "laplace_with_softmax", "dreg_looser", True)


def test_base_multivae_constructor_checks():
    """Error behaviour of base_ae_model.py:42-99,154-180."""
    from multivae_amd.models import MoPoE, MoPoEConfig
    from multivae_amd.models.base.base_config import BaseAEConfig
    from multivae_amd.models.nn.default_architectures import Decoder_AE_MLP, Encoder_VAE_MLP

    with pytest.raises(AttributeError):  # n_modalities vs input_dims
        MoPoE(MoPoEConfig(n_modalities=3, input_dims=dict(a=(2,), b=(3,))))
    with pytest.raises(AttributeError):  # neither encoders nor input_dims
        MoPoE(MoPoEConfig(n_modalities=2))
    enc = dict(a=Encoder_VAE_MLP(BaseAEConfig(input_dim=(2,), latent_dim=4)),
               b=Encoder_VAE_MLP(BaseAEConfig(input_dim=(3,), latent_dim=4)))
    dec = dict(a=Decoder_AE_MLP(BaseAEConfig(input_dim=(2,), latent_dim=4)),
               c=Decoder_AE_MLP(BaseAEConfig(input_dim=(3,), latent_dim=4)))
    with pytest.raises(AttributeError):  # names differ between encoders and decoders
        MoPoE(MoPoEConfig(n_modalities=2, latent_dim=4), enc, dec)
    dec = dict(a=dec["a"], b=dec["c"])
    with pytest.raises(KeyError):  # input_dims keys != encoder keys
        MoPoE(MoPoEConfig(n_modalities=2, latent_dim=4, input_dims=dict(a=(2,), z=(3,))), enc, dec)
    with pytest.raises(AttributeError):  # not a BaseEncoder
        MoPoE(MoPoEConfig(n_modalities=2, latent_dim=4), dict(a=torch.nn.Linear(2, 2), b=enc["b"]), dec)
    m = MoPoE(MoPoEConfig(n_modalities=2, latent_dim=4, input_dims=dict(a=(2,), b=(3,))), enc, dec)
    assert m.model_config.custom_architectures == ["encoders", "decoders"]
    # parameter order of the reference: decoders are registered before encoders
    assert list(m.state_dict().keys())[0].startswith("decoders.")
    with pytest.raises(ValueError):
        MoPoE(MoPoEConfig(n_modalities=2, input_dims=dict(a=(2,), b=(3,)), decoders_dist=dict(a="normal", b="foo")))


def test_rescale_factors_and_dists():
    from multivae_amd.models import MVTCAE, MVTCAEConfig

    dims = dict(mnist=(1, 28, 28), svhn=(3, 32, 32))
    m = MVTCAE(MVTCAEConfig(n_modalities=2, latent_dim=20, input_dims=dims, uses_likelihood_rescaling=True,
                            decoders_dist=dict(mnist="laplace", svhn="bernoulli"),
                            decoder_dist_params=dict(mnist=dict(scale=0.75))))
    assert abs(m.rescale_factors["mnist"] - 3072 / 784) < 1e-12 and m.rescale_factors["svhn"] == 1.0
    assert m.recon_dists == {"mnist": (1, 0.75), "svhn": (2, 1.0)}
    assert sum(p.numel() for p in m.parameters()) == 4541280  # SURVEY.md §2.3 C1 (MVTCAE MLP)


def test_mopoe_subset_enumeration_and_selection():
    from multivae_amd.models import MoPoE, MoPoEConfig
    from oracle import elbo

    dims = dict(zeta=(2,), alpha=(3,), mid=(4,))  # deliberately unsorted
    m = MoPoE(MoPoEConfig(n_modalities=3, latent_dim=4, input_dims=dims))
    assert list(m.subsets.keys()) == ["", "zeta", "alpha", "mid", "alpha_zeta", "mid_zeta", "alpha_mid",
                                      "alpha_mid_zeta"]
    assert m._subset_keys == [k for k, _ in elbo.mopoe_subsets(list(dims))]
    assert m._poe_order == ["alpha", "mid", "zeta"]
    assert m._subset_bits == [4, 1, 2, 5, 6, 3, 7]
    for B in (1, 6, 16, 512, 513):
        sel = m._row_range_selection(B, torch.device("cpu")).tolist()
        bnd = elbo.mopoe_row_bounds(B, 7)
        ref = [k for k in range(7) for _ in range(bnd[k + 1] - bnd[k])]
        assert sel == ref
    with pytest.raises(AttributeError):
        MoPoE(MoPoEConfig(n_modalities=3, latent_dim=4, input_dims=dims, subsets=[["zeta", "nope"]]))


def test_shard_indices_match_distributed_sampler():
    from torch.utils.data import DistributedSampler

    from multivae_amd.trainers.base import shard_indices

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 103

        def __getitem__(self, i):
            return i

    for W in (2, 4, 8):
        for r in range(W):
            ref = list(DistributedSampler(DS(), num_replicas=W, rank=r))  # shuffle=True, seed=0, no set_epoch
            assert shard_indices(103, W, r).tolist() == ref


def test_trainer_config_validation(monkeypatch):
    from multivae_amd.trainers import BaseTrainerConfig

    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    c = BaseTrainerConfig()
    assert (c.per_device_train_batch_size, c.learning_rate, c.optimizer_cls, c.seed, c.dist_backend) == \
        (64, 1e-4, "Adam", 8, "nccl")
    with pytest.raises(AttributeError):
        BaseTrainerConfig(optimizer_cls="NotAnOptimizer")
    with pytest.raises(TypeError):
        BaseTrainerConfig(optimizer_params=dict(wrong_kw=1))
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("LOCAL_RANK", "1")
    c = BaseTrainerConfig()
    assert (c.world_size, c.rank, c.local_rank) == (4, 3, 1)


def test_jmvae_and_mmvaeplus_save_load_roundtrip(tmp_path):
    """The reference folder layout (model.pt / model_config.json / environment.json) for the joint-encoder and the
    split-latent models; state_dict keys follow the reference's module tree (joint_encoder.*, mean_priors.*, ...)."""
    from multivae_amd.models import JMVAE, JMVAEConfig, MMVAEPlus, MMVAEPlusConfig

    dims = dict(a=(2, 3), b=(7,))
    jm = JMVAE(JMVAEConfig(n_modalities=2, latent_dim=5, input_dims=dims, alpha=0.3, warmup=4))
    assert {"joint_encoder.fc1.weight", "joint_encoder.encoders.a.embedding.weight", "joint_encoder.enc.1.0.bias"} <= set(
        jm.state_dict())
    jm.save(str(tmp_path / "jm"))
    jm2 = JMVAE.load_from_folder(str(tmp_path / "jm"))
    assert jm2.model_config.alpha == 0.3 and jm2.warmup == 4
    assert all(torch.equal(v, jm2.state_dict()[k]) for k, v in jm.state_dict().items())
    mp = MMVAEPlus(MMVAEPlusConfig(n_modalities=2, latent_dim=5, input_dims=dims, modalities_specific_dim=3, beta=2.5))
    assert {"logvars_priors.shared", "mean_priors.a", "encoders.b.style_log_var.bias"} <= set(mp.state_dict())
    assert mp.logvars_priors["a"].requires_grad and not mp.logvars_priors["shared"].requires_grad
    assert mp.state_dict()["decoders.a.layers.0.0.weight"].shape == (512, 8)  # shared + private latent
    mp.save(str(tmp_path / "mp"))
    mp2 = MMVAEPlus.load_from_folder(str(tmp_path / "mp"))
    assert mp2.beta == 2.5 and mp2.modalities_specific_dim == 3
    assert all(torch.equal(v, mp2.state_dict()[k]) for k, v in mp.state_dict().items())
    with pytest.raises(AttributeError):
        JMVAE(JMVAEConfig(n_modalities=2, latent_dim=5, input_dims=dims), joint_encoder=torch.nn.Linear(2, 2))


def test_fused_adam_state_dict_is_torch_adam_layout():
    """`optimizer.pt` compatibility (base_trainer.py:790-793 writes torch.optim.Adam.state_dict(), :413-419 reads it):
    the fused Adam's state loads into torch.optim.Adam and back, parameter by parameter, including parameters that do
    not require grad (they keep their index, they have no state)."""
    from multivae_amd.models import MMVAE, MMVAEConfig
    from multivae_amd.trainers.flat import FlatParams, FusedAdam

    torch.manual_seed(0)
    model = MMVAE(MMVAEConfig(n_modalities=2, latent_dim=4, input_dims=dict(a=(3,), b=(2, 2)), learn_prior=True))
    assert not model.prior_mean.requires_grad  # a frozen parameter in the middle of model.parameters()
    ref_opt = torch.optim.Adam(model.parameters(), lr=3e-4, betas=(0.8, 0.95), eps=1e-7)
    for _ in range(2):
        for p in model.parameters():
            p.grad = torch.randn_like(p) if p.requires_grad else None
        ref_opt.step()
    sd = ref_opt.state_dict()
    for p in model.parameters():
        p.grad = None
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=1.0)
    opt.load_state_dict(sd)
    assert opt.step_count == 2 and opt.lr == 3e-4 and opt.betas == (0.8, 0.95) and opt.eps == 1e-7
    mine = opt.state_dict()
    assert mine["param_groups"][0]["params"] == sd["param_groups"][0]["params"]
    assert set(mine["state"]) == set(sd["state"])
    for i, st in sd["state"].items():
        assert torch.equal(mine["state"][i]["exp_avg"], st["exp_avg"])
        assert torch.equal(mine["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
        assert float(mine["state"][i]["step"]) == 2.0
    fresh = torch.optim.Adam(model.parameters(), lr=1.0)
    fresh.load_state_dict(mine)  # torch validates group sizes and casts the state
    assert fresh.state_dict()["param_groups"][0]["lr"] == 3e-4
    assert FusedAdam(flat).state_dict()["state"] == {}  # like torch: no state before the first step
    bad = {"state": {}, "param_groups": [dict(sd["param_groups"][0], params=[0, 1])]}
    with pytest.raises(ValueError):
        opt.load_state_dict(bad)


def test_fused_adam_is_a_torch_optimizer_with_schedulers_and_amsgrad_state():
    """FusedAdam is a torch.optim.Optimizer with one parameter group: lr schedulers drive it unchanged (the trainer
    accepts `scheduler_cls` with the fused optimizer), and the amsgrad state round-trips in torch.optim.Adam's layout."""
    from multivae_amd.models import MVTCAE, MVTCAEConfig
    from multivae_amd.trainers.flat import FlatParams, FusedAdam

    torch.manual_seed(0)
    model = MVTCAE(MVTCAEConfig(n_modalities=2, latent_dim=3, input_dims=dict(a=(4,), b=(5,))))
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=1e-2, amsgrad=True, weight_decay=0.1)
    assert isinstance(opt, torch.optim.Optimizer) and len(opt.param_groups) == 1
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    opt.step_count = 1  # (no GPU here: pretend a step happened; the kernel launch itself is covered by the GPU tests)
    sched.step()
    assert opt.lr == pytest.approx(5e-3)
    plateau = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.1, patience=0)
    plateau.step(1.0)
    plateau.step(2.0)
    assert opt.lr == pytest.approx(5e-4)
    # amsgrad state in torch's layout, both directions
    ref = torch.optim.Adam(model.parameters(), lr=1e-3, amsgrad=True)
    for p in model.parameters():
        p.grad = torch.randn_like(p)
    ref.step()
    sd = ref.state_dict()
    opt.load_state_dict(sd)
    assert opt.amsgrad and opt.vmax is not None and opt.step_count == 1
    mine = opt.state_dict()
    assert mine["param_groups"][0]["amsgrad"] is True
    for i, st in sd["state"].items():
        assert torch.equal(mine["state"][i]["max_exp_avg_sq"], st["max_exp_avg_sq"])
    fresh = torch.optim.Adam(model.parameters(), lr=1.0, amsgrad=True)
    fresh.load_state_dict(mine)
    plain = FusedAdam(flat, lr=1e-3)
    plain.load_state_dict(torch.optim.Adam(model.parameters(), lr=1e-3).state_dict())
    assert plain.vmax is None and not plain.amsgrad


def test_training_config_json_is_reference_compatible(tmp_path):
    """training_config.json holds the reference's fields only (its from_json_file rejects unknown keys); the
    multivae_amd switches live in a side file and come back on load."""
    import json

    from multivae_amd.trainers import BaseTrainerConfig

    cfg = BaseTrainerConfig(num_epochs=3, use_hip_graph=True, sync_every_step=True, scheduler_cls="StepLR",
                            scheduler_params=dict(step_size=2))
    cfg.save_json(str(tmp_path), "training_config")
    with open(tmp_path / "training_config.json") as f:
        d = json.load(f)
    assert not ({"use_fused_adam", "sync_every_step", "use_hip_graph"} & set(d)) and d["num_epochs"] == 3
    back = BaseTrainerConfig.from_json_file(str(tmp_path / "training_config.json"))
    assert back.use_hip_graph and back.sync_every_step and back.scheduler_cls == "StepLR"
    (tmp_path / "training_config_mvk.json").unlink()
    assert BaseTrainerConfig.from_json_file(str(tmp_path / "training_config.json")).use_hip_graph is False


def test_auto_model_and_auto_config(tmp_path):
    """AutoModel / AutoConfig pick the class from the `name` field of model_config.json (auto_model.py:41-98)."""
    import json

    from multivae_amd.models import AutoConfig, AutoModel, MoPoE, MoPoEConfig, MVTCAE, MVTCAEConfig

    dims = dict(a=(2, 3), b=(7,))
    m = MoPoE(MoPoEConfig(n_modalities=2, latent_dim=5, input_dims=dims, beta=2.0))
    m.save(str(tmp_path / "mopoe"))
    with open(tmp_path / "mopoe" / "model_config.json") as f:
        d = json.load(f)
    assert d["name"] == "MoPoEConfig" and d["input_dims"] == {"a": [2, 3], "b": [7]} and d["subsets"]["a_b"] == ["a", "b"]
    back = AutoModel.load_from_folder(str(tmp_path / "mopoe"))
    assert isinstance(back, MoPoE) and back.model_config.beta == 2.0
    assert all(torch.equal(v, back.state_dict()[k]) for k, v in m.state_dict().items())
    MVTCAE(MVTCAEConfig(n_modalities=2, latent_dim=5, input_dims=dims, alpha=0.4)).save(str(tmp_path / "mvt"))
    cfg = AutoConfig.from_json_file(str(tmp_path / "mvt" / "model_config.json"))
    assert isinstance(cfg, MVTCAEConfig) and cfg.alpha == 0.4
    d["name"] = "SomethingElseConfig"
    with open(tmp_path / "mopoe" / "model_config.json", "w") as f:
        json.dump(d, f)
    with pytest.raises(NameError):
        AutoModel.load_from_folder(str(tmp_path / "mopoe"))


def test_flat_params_align_every_parameter_and_keep_the_optimizer_layout():
    """FlatParams: every parameter (and its gradient) starts on a 256-byte boundary of the flat buffer — with dense packing a
    3-element bias pushes everything behind it off the 16-byte alignment the kernels' vector loads need —, the padding stays
    zero, `dense()` strips it, and the fused Adam's state_dict is still keyed / shaped like torch.optim.Adam's."""
    from multivae_amd.trainers import FlatParams, FusedAdam

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 7), torch.nn.Linear(7, 2))  # odd sizes: 15, 3, 21, 7, 14, 2
    ref = [p.detach().clone() for p in model.parameters()]
    flat = FlatParams(model)
    assert len(flat.offsets) == 6 and flat.numel == 6 * FlatParams.ALIGN
    for p, r, off in zip(model.parameters(), ref, flat.offsets):
        assert off % FlatParams.ALIGN == 0
        assert (p.data_ptr() - flat.flat.data_ptr()) == 4 * off and (p.grad.data_ptr() - flat.grad.data_ptr()) == 4 * off
        assert torch.equal(p.detach(), r)
    assert torch.equal(flat.dense(flat.flat), torch.cat([r.reshape(-1) for r in ref]))
    mask = torch.ones(flat.numel, dtype=torch.bool)
    for p, off in zip(flat.params, flat.offsets):
        mask[off:off + p.numel()] = False
    assert float(flat.flat[mask].abs().sum()) == 0.0  # padding
    opt = FusedAdam(flat, lr=1e-3)
    opt.step_count = 3  # as if three steps had run (the kernel itself needs the GPU)
    opt.m.copy_(torch.arange(flat.numel, dtype=torch.float32))
    sd = opt.state_dict()
    tref = torch.optim.Adam(model.parameters(), lr=1e-3)
    assert sd["param_groups"][0]["params"] == tref.state_dict()["param_groups"][0]["params"]
    for i, (p, off) in enumerate(zip(flat.params, flat.offsets)):
        assert sd["state"][i]["exp_avg"].shape == p.shape
        assert torch.equal(sd["state"][i]["exp_avg"].reshape(-1), torch.arange(off, off + p.numel(), dtype=torch.float32))
    opt2 = FusedAdam(FlatParams(torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 7), torch.nn.Linear(7, 2))), lr=1e-3)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 3 and torch.equal(opt2.flat.dense(opt2.m), flat.dense(opt.m))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/multivae"), reason="needs the reference checkout (build container only)")
def test_checkpoint_compat_script_against_the_real_reference():
    """tests/golden/checkpoint_compat.py (folders written by either side load on the other side with identical
    configuration fields, state_dict and Adam moments) runs green as committed — INTEGRATION.md cites it."""
    import subprocess
    import sys

    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint_compat.py")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "optimizer state back OK" in r.stdout
