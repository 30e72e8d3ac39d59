"""Checkpoint / configuration on-disk compatibility with the REAL reference.  Build container only
(needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/checkpoint_compat.py

For MoPoE, MVTCAE, MMVAE, MMVAE+ and JMVAE (default architectures):
  1. a folder written by `multivae_amd` (`model.save`) is loaded by the reference's `AutoModel.load_from_folder`
     and must give the same configuration fields and bit-identical state_dict;
  2. a folder written by the reference is loaded by `multivae_amd.models.AutoModel.load_from_folder`, same checks.
  3. a `checkpoint_epoch_N` folder written by the reference's BaseTrainer (CPU, 2 epochs of MVTCAE) is read by
     multivae_amd: `training_config.json` -> BaseTrainerConfig, `info_checkpoint.json` keys, `optimizer.pt`
     (torch.optim.Adam state) -> FusedAdam.load_state_dict, and FusedAdam.state_dict() loads back into the
     reference trainer's torch.optim.Adam with identical moments.
Nothing is written into the repository.
"""
import sys

sys.dont_write_bytecode = True
import json
import os
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import torch

import _reference_import as R

R.install()
import multivae.models as ref
from multivae.models.auto_model import AutoModel as RefAutoModel

import multivae_amd.models as mine

DIMS = dict(mod1=(2,), mod2=(3,), mod3=(4,), mod4=(4,))
CASES = {
    "MoPoE": dict(beta=2.5, decoders_dist=dict(mod1="normal", mod2="laplace", mod3="bernoulli", mod4="normal")),
    "MVTCAE": dict(alpha=0.3, beta=1.5, uses_likelihood_rescaling=True),
    "MMVAE": dict(K=3, prior_and_posterior_dist="normal", learn_prior=True),
    "MMVAEPlus": dict(K=2, modalities_specific_dim=3, beta=2.5, learn_shared_prior=True),
    "JMVAE": dict(alpha=0.2, warmup=5),
}


EXTENSIONS = {"K"}  # multivae_amd-only configuration fields (MoPoE / MVTCAE K-sample extension)


def same_config(ref_cfg, my_cfg):
    """Every field of the reference's configuration has the same value; multivae_amd may add EXTENSIONS (the
    reference's pydantic dataclasses ignore unknown keys when they read the JSON)."""
    dr, dm = json.loads(ref_cfg.to_json_string()), json.loads(my_cfg.to_json_string())
    assert set(dm) - set(dr) <= EXTENSIONS and not set(dr) - set(dm), (set(dm) ^ set(dr))
    for k in dr:
        assert dr[k] == dm[k], (k, dr[k], dm[k])


def same_state(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys()), (set(sa) ^ set(sb))
    for k in sa:
        assert torch.equal(sa[k].cpu(), sb[k].cpu()), k


def trainer_checkpoint():
    from multivae.data.datasets.base import MultimodalBaseDataset
    from multivae.trainers import BaseTrainer as RefTrainer
    from multivae.trainers import BaseTrainerConfig as RefTrainerConfig

    from multivae_amd.trainers import BaseTrainerConfig
    from multivae_amd.trainers.flat import FlatParams, FusedAdam

    dims = dict(mod1=(2,), mod2=(3,))
    torch.manual_seed(0)
    data = MultimodalBaseDataset(data={m: torch.rand(24, *d) for m, d in dims.items()}, labels=torch.zeros(24))
    common = dict(n_modalities=2, latent_dim=4, input_dims=dict(dims))
    model = ref.MVTCAE(ref.MVTCAEConfig(**common))
    with tempfile.TemporaryDirectory() as d:
        cfg = RefTrainerConfig(output_dir=d, num_epochs=2, steps_saving=2, per_device_train_batch_size=8,
                               learning_rate=1e-3, no_cuda=True, optimizer_params=dict(betas=(0.8, 0.95)))
        trainer = RefTrainer(model, data, training_config=cfg)
        trainer.train()
        ckpt = os.path.join(trainer.training_dir, "checkpoint_epoch_2")
        files = sorted(os.listdir(ckpt))
        assert {"model.pt", "optimizer.pt", "model_config.json", "training_config.json", "info_checkpoint.json",
                "metrics_best_model.json", "environment.json"} <= set(files), files
        # training configuration
        tc = BaseTrainerConfig.from_json_file(os.path.join(ckpt, "training_config.json"))
        dr = json.loads(cfg.to_json_string())
        dm = json.loads(tc.to_json_string())
        assert not set(dr) - set(dm), set(dr) - set(dm)
        for k in dr:
            assert dr[k] == dm[k], (k, dr[k], dm[k])
        print("  training_config.json: all", len(dr), "reference fields read back; multivae_amd adds", sorted(set(dm) - set(dr)))
        with open(os.path.join(ckpt, "info_checkpoint.json")) as f:
            assert set(json.load(f)) == {"training_dir", "trained_epochs", "best_train_loss", "best_eval_loss"}
        # model + optimizer state into the flat fused Adam
        m2 = mine.AutoModel.load_from_folder(ckpt)
        flat = FlatParams(m2)
        opt = FusedAdam(flat, lr=1.0)
        sd = torch.load(os.path.join(ckpt, "optimizer.pt"), map_location="cpu")
        opt.load_state_dict(sd)
        assert opt.step_count == 6 and opt.lr == 1e-3 and opt.betas == (0.8, 0.95)
        ref_params = list(trainer._best_model.parameters())
        for p, off in zip(flat.params, flat.offsets):  # parameters sit on 256-byte boundaries of the flat buffer
            i = [id(q) for q in flat.all_params].index(id(p))
            k = p.numel()
            assert torch.equal(opt.m[off:off + k], sd["state"][i]["exp_avg"].reshape(-1))
            assert torch.equal(opt.v[off:off + k], sd["state"][i]["exp_avg_sq"].reshape(-1))
        # and back: the fused Adam's state loads into the reference trainer's optimizer
        trainer.optimizer.load_state_dict(opt.state_dict())
        back = trainer.optimizer.state_dict()
        for i, st in sd["state"].items():
            assert torch.equal(back["state"][i]["exp_avg"], st["exp_avg"])
            assert torch.equal(back["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
            assert float(back["state"][i]["step"]) == float(st["step"])
        assert len(ref_params) == len(flat.all_params)
    print("trainer checkpoint: reference -> multivae_amd (config, info, model, optimizer) OK, optimizer state back OK")


def main():
    for name, kw in CASES.items():
        common = dict(n_modalities=4, latent_dim=5, input_dims=dict(DIMS))
        torch.manual_seed(0)
        m_mine = getattr(mine, name)(getattr(mine, name + "Config")(**common, **kw))
        torch.manual_seed(1)
        m_ref = getattr(ref, name)(getattr(ref, name + "Config")(**common, **kw))
        with tempfile.TemporaryDirectory() as d:
            m_mine.save(d)
            back = RefAutoModel.load_from_folder(d)
            assert type(back).__name__ == name
            same_config(back.model_config, m_mine.model_config)
            same_state(back, m_mine)
        with tempfile.TemporaryDirectory() as d:
            m_ref.save(d)
            back = mine.AutoModel.load_from_folder(d)
            assert type(back).__name__ == name and type(back).__module__.startswith("multivae_amd")
            same_config(m_ref.model_config, back.model_config)
            same_state(back, m_ref)
        print(f"{name}: multivae_amd -> reference OK, reference -> multivae_amd OK")
    trainer_checkpoint()


if __name__ == "__main__":
    main()
