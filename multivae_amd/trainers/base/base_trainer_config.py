"""`multivae/trainers/base/base_trainer_config.py:10-152` — same fields, defaults and validation, plus two
MI355X-specific switches (fused flat Adam, deferred host sync)."""
import os
from dataclasses import field
from typing import Union

import torch.nn as nn
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

    # the reference's BaseTrainerConfig.from_json_file rejects unknown fields: the extension fields go to a side file
    _EXTENSION_FIELDS = ("use_fused_adam", "sync_every_step", "use_hip_graph", "fused_zero_grad")

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

    def __post_init__(self):
        super().__post_init__()
        env_local_rank = int(os.environ.get("LOCAL_RANK", -1))
        if self.local_rank == -1 and env_local_rank != -1:
            self.local_rank = env_local_rank
        env_world_size = int(os.environ.get("WORLD_SIZE", -1))
        if self.world_size == -1 and env_world_size != -1:
            self.world_size = env_world_size
        env_rank = int(os.environ.get("RANK", -1))
        if self.rank == -1 and env_rank != -1:
            self.rank = env_rank
        env_master_addr = os.environ.get("MASTER_ADDR", "localhost")
        if self.master_addr == "localhost" and env_master_addr != "localhost":
            self.master_addr = env_master_addr
        os.environ["MASTER_ADDR"] = self.master_addr
        env_master_port = os.environ.get("MASTER_PORT", "12345")
        if self.master_port == "12345" and env_master_port != "12345":
            self.master_port = env_master_port
        os.environ["MASTER_PORT"] = self.master_port

        import torch.optim as optim

        try:
            optimizer_cls = getattr(optim, self.optimizer_cls)
        except AttributeError:
            raise AttributeError(f"Unable to import `{self.optimizer_cls}` optimizer from 'torch.optim'. "
                                 "Check spelling and that it is part of 'torch.optim.Optimizers.'")
        try:
            optimizer = optimizer_cls(nn.Linear(2, 2).parameters(), lr=self.learning_rate,
                                      **(self.optimizer_params or {}))
        except TypeError as e:
            raise TypeError("Error in optimizer's parameters. Check that the provided dict contains only "
                            f"keys and values suitable for `{optimizer_cls}` optimizer. "
                            f"Got {self.optimizer_params} as parameters.\n"
                            f"Exception raised: {type(e)} with message: " + str(e)) from e
        if self.scheduler_cls is not None:
            import torch.optim.lr_scheduler as schedulers

            try:
                scheduler_cls = getattr(schedulers, self.scheduler_cls)
            except AttributeError:
                raise AttributeError(f"Unable to import `{self.scheduler_cls}` scheduler from "
                                     "'torch.optim.lr_scheduler'. Check spelling and that it is part of "
                                     "'torch.optim.lr_scheduler.'")
            try:
                scheduler_cls(optimizer, **(self.scheduler_params or {}))
            except TypeError as e:
                raise TypeError("Error in scheduler's parameters. Check that the provided dict contains only "
                                f"keys and values suitable for `{scheduler_cls}` scheduler. "
                                f"Got {self.scheduler_params} as parameters.\n"
                                f"Exception raised: {type(e)} with message: " + str(e)) from e
