"""`multivae/trainers/base/base_trainer_config.py:10-152` — same fields, defaults and validation, plus two
MI355X-specific switches (fused flat Adam, deferred host sync)."""
import os
from dataclasses import field
from typing import Union

import torch
import torch.nn as nn
import torch.optim.lr_scheduler  # noqa: F401  (resolved by name in __post_init__)
from pydantic.dataclasses import dataclass

from ...models.base.base_config import BaseConfig


@dataclass
class BaseTrainerConfig(BaseConfig):
    output_dir: str = None
    per_device_train_batch_size: int = 64
    per_device_eval_batch_size: int = 64
    num_epochs: int = 100
    train_dataloader_num_workers: int = 0
    eval_dataloader_num_workers: int = 0
    optimizer_cls: str = "Adam"
    optimizer_params: Union[dict, None] = None
    scheduler_cls: Union[str, None] = None
    scheduler_params: Union[dict, None] = None
    learning_rate: float = 1e-4
    steps_saving: Union[int, None] = None
    steps_predict: Union[int, None] = None
    keep_best_on_train: bool = False
    seed: int = 8
    no_cuda: bool = False
    world_size: int = field(default=-1)
    local_rank: int = field(default=-1)
    rank: int = field(default=-1)
    dist_backend: str = field(default="nccl")  # "nccl" IS RCCL on ROCm
    master_addr: str = field(default="localhost")
    master_port: str = field(default="12345")
    drop_last: bool = False
    gradient_clipping_max_norm: Union[float, None] = None
    # --- multivae_amd extensions -----------------------------------------------------------------
    use_fused_adam: bool = True     # optimizer_cls == "Adam": one mvk_adam_step launch over the flat buffer
    sync_every_step: bool = False   # True reproduces the reference's per-step `.item()` host sync
    use_hip_graph: bool = False     # replay zero_grad+forward+backward of full-size batches as ONE hipGraph launch
    # the fused Adam launch clears the gradients it consumes and the next zero_grad() is a flag test.  Correct for the loop
    # zero_grad -> backward -> step of this trainer; a callback that writes gradients between step() and the next
    # zero_grad() would see them survive, so None (default) = on only while no user callback is installed
    fused_zero_grad: Union[bool, None] = None
    # with use_hip_graph: the fused Adam as the last node of the replayed graph (step and learning rate in device memory; single
    # GPU), and the data-parallel gradient collective in two parts, the first started by an event node inside the graph beside
    # the end of the backward pass.  Both are correct (tested) and both measured SLOWER on the MnistSvhn step (+7 us; +65 us for
    # the event node alone): off by default, worth trying where the optimizer launch or the collective is a large part of the step
    graph_optimizer: bool = False
    overlap_collective: bool = False
    # with use_hip_graph, single GPU, FusedAdam(zero_grad_in_step): the ROTATED step (trainers/graph.py, kernels.Rotation) — the
    # decoders' late weight gradients of step N, their finishes and their share of optimizer.step() run at the head of replay
    # N + 1 beside the encoders' forward pass; the trainer drains what is pending before anything but another replay reads the
    # parameters (eager steps of ragged batches, the end of an epoch: evaluation, checkpoints, callbacks).  Exact (parameters
    # bit for bit those of the sequential loop, tested) and measured SLOWER on the MnistSvhn step on one MI355X (+2 % with the
    # SVHN decoder's leaves, +7 % with both decoders': the leaves delay the encoders' chain by what they take): off by default
    rotate_step: bool = False

    # the reference's BaseTrainerConfig.from_json_file rejects unknown fields: the extension fields go to a side file
    _EXTENSION_FIELDS = ("use_fused_adam", "sync_every_step", "use_hip_graph", "fused_zero_grad", "graph_optimizer",
                         "overlap_collective", "rotate_step")

    def save_json(self, dir_path, filename):
        """`<filename>.json` holds exactly the reference's fields (loads in either trainer); the multivae_amd switches
        go to `<filename>_mvk.json` beside it."""
        import json

        d = self.to_dict()
        ext = {k: d.pop(k) for k in self._EXTENSION_FIELDS}
        with open(os.path.join(dir_path, f"{filename}.json"), "w", encoding="utf-8") as f:
            f.write(json.dumps(d))
        with open(os.path.join(dir_path, f"{filename}_mvk.json"), "w", encoding="utf-8") as f:
            f.write(json.dumps(ext))

    @classmethod
    def from_json_file(cls, json_path):
        import json

        with open(json_path) as f:
            d = json.load(f)
        name = d.pop("name", None)
        if name is not None and name != cls.__name__:
            raise ValueError(f"config file is for {name}, not {cls.__name__}")
        side = json_path[:-5] + "_mvk.json" if json_path.endswith(".json") else None
        if side and os.path.exists(side):
            with open(side) as f:
                d.update(json.load(f))
        return cls.from_dict(d)

    # (field, environment variable, "unset" value, converter): torchrun's variables fill what the user left unset
    _LAUNCHER_ENV = (("local_rank", "LOCAL_RANK", -1, int), ("world_size", "WORLD_SIZE", -1, int), ("rank", "RANK", -1, int),
                     ("master_addr", "MASTER_ADDR", "localhost", str), ("master_port", "MASTER_PORT", "12345", str))

    def __post_init__(self):
        """Same contract as the reference's configuration (trainers/base/base_trainer_config.py:77-152): a distributed field
        left at its default takes the launcher's environment variable, the rendezvous address / port are exported for
        `init_process_group`, and a misspelt optimizer / scheduler name (AttributeError) or an argument its constructor
        rejects (TypeError) fails here, when the configuration is built, not in the middle of `train()`."""
        super().__post_init__()
        for attr, var, unset, conv in self._LAUNCHER_ENV:
            if getattr(self, attr) == unset and var in os.environ:
                setattr(self, attr, conv(os.environ[var]))
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = self.master_addr, self.master_port
        probe = self._probe("optimizer", torch.optim, self.optimizer_cls, self.optimizer_params,
                            lambda cls, kw: cls(nn.Linear(2, 2).parameters(), lr=self.learning_rate, **kw))
        if self.scheduler_cls is not None:
            self._probe("scheduler", torch.optim.lr_scheduler, self.scheduler_cls, self.scheduler_params,
                        lambda cls, kw: cls(probe, **kw))

    @staticmethod
    def _probe(kind, module, name, params, build):
        """Resolve `name` in `module` and construct it once on a throw-away 2 x 2 layer."""
        cls = getattr(module, name, None)
        if cls is None:
            raise AttributeError(f"{module.__name__} has no {kind} called `{name}`: check the spelling of `{kind}_cls`.")
        try:
            return build(cls, dict(params or {}))
        except TypeError as err:
            raise TypeError(f"`{kind}_params` = {params} does not fit {cls.__name__}: {err}") from err
