"""BaseTrainer: the reference's training loop (`multivae/trainers/base/base_trainer.py`) on the MI355X-native
step: flat fp32 parameter / gradient buffers, ONE fused HIP Adam launch, ONE RCCL all-reduce of the flat
gradient buffer per step under data parallelism (one process per GPU), batches gathered on the device.

Kept from the reference: per-device batch size, shuffling semantics of
`DataLoader(shuffle=True)` / `DistributedSampler` (seed 0, never `set_epoch`), model call kwargs
(`epoch, dataset_size, uses_ddp, batch_ratio, beta`), `loss_sum` accumulation, division of the epoch loss by
the FULL dataset length, metric averaging over batches, best-model bookkeeping, checkpoint directory layout,
DDP gradient AVERAGING (sum all-reduce, 1/world_size folded into Adam), gradient clipping placed before
`backward()` (a no-op, reproduced by omitting it), `ArithmeticError` on a NaN loss.
Dropped (no effect on results): `torch.cuda.empty_cache()` per step, per-sample `__getitem__` + collate,
the per-step `.item()` host sync (one sync per epoch unless `sync_every_step=True`).
"""
import datetime
import json
import logging
import os
from copy import deepcopy
from typing import List, Optional

import torch
import torch.distributed as dist

from ... import kernels
from ...data.datasets.base import DatasetOutput
from ...models.base.base_model import BaseModel
from ..flat import FlatParams, FusedAdam
from .base_trainer_config import BaseTrainerConfig
from .callbacks import CallbackHandler, MetricConsolePrinterCallback, TrainingCallback

logger = logging.getLogger(__name__)


def set_seed(seed: int):
    import random

    import numpy as np

    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def update_dict(dict1, dict2):
    """Running sums of the batch metrics.  The first value is COPIED: under hipGraph replay the model output's tensors
    are static buffers that the next replay overwrites."""
    for k in dict2.keys():
        if k in dict1:
            dict1[k] = dict1[k] + dict2[k]
        else:
            dict1[k] = dict2[k].clone() if torch.is_tensor(dict2[k]) else dict2[k]


def shard_indices(n: int, world_size: int, rank: int, seed: int = 0, epoch: int = 0) -> torch.Tensor:
    """`torch.utils.data.DistributedSampler(shuffle=True, seed=0)` as used by the reference without `set_epoch`
    (base_trainer.py:198-211): a seeded permutation, padded to a multiple of world_size, rank r takes r::W."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g)
    total = ((n + world_size - 1) // world_size) * world_size
    if total > n:
        pad = total - n
        idx = torch.cat([idx, idx[:pad]]) if pad <= n else torch.cat([idx] * (total // n + 1))[:total]
    return idx[rank:total:world_size]


class _BatchIterator:
    """Device-resident replacement of DataLoader + default collate for `MultimodalBaseDataset`-style datasets
    (dict of tensors): one index_select per modality per batch instead of B `__getitem__` calls."""

    def __init__(self, dataset, batch_size, shuffle, device, world_size=1, rank=0, drop_last=False):
        self.dataset = dataset
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.device = device
        self.world_size, self.rank = world_size, rank
        self.drop_last = drop_last
        base = dataset
        self.index_map = None
        if isinstance(dataset, torch.utils.data.Subset):  # random_split of the reference's example
            self.index_map = torch.as_tensor(dataset.indices)
            base = dataset.dataset
        self.base = base
        # device-resident gather only for the stock datasets: a subclass that overrides __getitem__ (transforms,
        # normalisation, dtype conversion) goes through its own __getitem__ below
        from ...data.datasets.base import IncompleteDataset, MultimodalBaseDataset
        from ...data.datasets.mmnist import MMNISTDataset
        from ...data.datasets.utils import ResampleDataset

        stock = type(base).__getitem__ in (MultimodalBaseDataset.__getitem__, IncompleteDataset.__getitem__,
                                           MMNISTDataset.__getitem__)
        plain = lambda v: torch.is_tensor(v) or (isinstance(v, ResampleDataset) and v.transform is None)
        self.fast = stock and isinstance(getattr(base, "data", None), dict) and all(plain(v) for v in base.data.values())
        self.resample = {}
        if self.fast:
            self.data = {}
            for m, v in base.data.items():
                if isinstance(v, ResampleDataset):  # base[index[i]]: the base set and the index live on the device
                    self.data[m] = v.base.to(device)
                    self.resample[m] = v.index.to(device)
                else:
                    self.data[m] = v.to(device)
            self.masks = None
            if getattr(base, "masks", None) is not None:
                self.masks = {m: torch.as_tensor(v).to(device) for m, v in base.masks.items()}
            self.labels = None
            if getattr(base, "labels", None) is not None and torch.is_tensor(base.labels):
                self.labels = base.labels.to(device)

    def __len__(self):
        n = self._n_local()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _n_local(self):
        n = len(self.dataset)
        return (n + self.world_size - 1) // self.world_size if self.world_size > 1 else n

    def _order(self):
        n = len(self.dataset)
        if self.world_size > 1:
            return shard_indices(n, self.world_size, self.rank)
        if self.shuffle:
            # RandomSampler: seed drawn from the global generator, then a private generator (torch semantics)
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            g = torch.Generator()
            g.manual_seed(seed)
            return torch.randperm(n, generator=g)
        return torch.arange(n)

    def __iter__(self):
        order = self._order()
        nb = len(self)
        if self.fast:
            # the epoch's whole order goes to the device ONCE: a per-batch `.to(device)` of a pageable host slice is a
            # synchronous copy — the host then waits for every step it has enqueued and cannot run ahead of the device
            # (round 6: the trainer's step was 1.06 ms around a 0.96 ms replay)
            order_dev = (order if self.index_map is None else self.index_map[order]).to(self.device)
        for b in range(nb):
            idx = order[b * self.batch_size : (b + 1) * self.batch_size]
            if self.fast:
                gi = order_dev[b * self.batch_size : (b + 1) * self.batch_size]
                out = {"data": {m: v.index_select(0, self.resample[m].index_select(0, gi) if m in self.resample else gi)
                                for m, v in self.data.items()}}
                if self.masks is not None:
                    out["masks"] = {m: v.index_select(0, gi) for m, v in self.masks.items()}
                if self.labels is not None:
                    out["labels"] = self.labels.index_select(0, gi)
                yield DatasetOutput(**out)
            else:  # generic datasets: per-sample items, collated here
                items = [self.dataset[int(i)] for i in idx]
                out = {"data": {m: torch.stack([torch.as_tensor(it["data"][m]) for it in items]).to(self.device)
                                for m in items[0]["data"]}}
                if "masks" in items[0]:
                    out["masks"] = {m: torch.stack([torch.as_tensor(it["masks"][m]) for it in items]).to(self.device)
                                    for m in items[0]["masks"]}
                if "labels" in items[0] and items[0]["labels"] is not None:
                    out["labels"] = torch.stack([torch.as_tensor(it["labels"]) for it in items]).to(self.device)
                yield DatasetOutput(**out)


class BaseTrainer:
    def __init__(self, model: BaseModel, train_dataset, eval_dataset=None,
                 training_config: Optional[BaseTrainerConfig] = None, callbacks: List[TrainingCallback] = None,
                 checkpoint: str = None):
        if training_config is None:
            # resuming: the configuration saved with the checkpoint (base_trainer.py:71-75), else the defaults
            cfg_file = os.path.join(checkpoint, "training_config.json") if checkpoint is not None else None
            training_config = BaseTrainerConfig.from_json_file(cfg_file) if cfg_file and os.path.exists(cfg_file) \
                else BaseTrainerConfig()
        self.training_config = training_config
        self.model_config = model.model_config
        self.model_name = model.model_name
        self.start_keep_best_epoch = getattr(model, "start_keep_best_epoch", 0)  # base_trainer.py:138
        self.world_size = training_config.world_size
        self.local_rank = training_config.local_rank
        self.rank = training_config.rank
        self.dist_backend = training_config.dist_backend
        self.distributed = self.world_size > 1
        self.device = self._setup_devices() if self.distributed else (
            "cuda" if torch.cuda.is_available() and not training_config.no_cuda else "cpu")
        # MVK_TRAINER_ALLOW_CPU=1: host-logic tests of THIS loop with plain-PyTorch plugin models (gloo, no GPU); the
        # multivae_amd models themselves still refuse CPU tensors (their arithmetic lives in HIP kernels)
        if str(self.device) == "cpu" and os.environ.get("MVK_TRAINER_ALLOW_CPU") != "1":
            raise RuntimeError("multivae_amd trains on MI355X GPUs only: the model's arithmetic lives in HIP kernels "
                               "(no CPU compute path)")
        self.device = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
        self.checkpoint = checkpoint
        if checkpoint is not None:
            model = type(model).load_from_folder(checkpoint)
        model = model.to(self.device)
        model.device = self.device
        self.model = model
        self.train_dataset = train_dataset
        self.eval_dataset = eval_dataset
        if eval_dataset is None:
            self.training_config.keep_best_on_train = True  # base_trainer.py:132
        W, r = (self.world_size, self.rank) if self.distributed else (1, 0)
        self.train_loader = _BatchIterator(train_dataset, training_config.per_device_train_batch_size, True,
                                           self.device, W, r, training_config.drop_last)
        self.eval_loader = None
        if eval_dataset is not None:
            # the reference's eval DataLoader never drops the last batch (base_trainer.py:213-232)
            self.eval_loader = _BatchIterator(eval_dataset, training_config.per_device_eval_batch_size, False,
                                              self.device, W, r, False)
        self.callbacks = [TrainingCallback()] if callbacks is None else callbacks
        self.is_main_process = self.rank in (-1, 0)
        self.optimizer = None
        self.scheduler = None
        self.flat = None
        self._run_model_sanity_check(self.model, self.train_loader)

    # -- distributed -----------------------------------------------------------------------------------------
    def _setup_devices(self):
        if torch.cuda.is_available() and not self.training_config.no_cuda:
            torch.cuda.set_device(self.local_rank)
            device = torch.device("cuda", self.local_rank)
        else:
            device = "cpu"
        if not dist.is_initialized():
            dist.init_process_group(backend=self.dist_backend, init_method="env://", world_size=self.world_size,
                                    rank=self.rank)
        return device

    def _run_model_sanity_check(self, model, loader):
        try:
            inputs = next(iter(loader))
            with torch.no_grad():
                model(inputs)
        except Exception as e:
            raise Exception("Error when calling forward method from model. Potential issues: \n"
                            " - Wrong model architecture -> check encoder, decoder and metric architecture if "
                            "you provide yours \n"
                            " - The data input dimension provided is wrong -> when no encoder, decoder or metric "
                            "provided, a network is built automatically but requires the shape of the flatten "
                            "input data.\n"
                            f"Exception raised: {type(e)} with message: " + str(e)) from e

    # -- optimizer / scheduler ---------------------------------------------------------------------------------
    def set_optimizer(self):
        cfg = self.training_config
        params = cfg.optimizer_params or {}
        self.flat = FlatParams(self.model)
        if self.distributed:
            self.flat.broadcast(0)  # C1: one parameter broadcast from rank 0 (DDP.__init__ in the reference)
        fused_ok = cfg.optimizer_cls == "Adam" and cfg.use_fused_adam and \
            set(params.keys()) <= {"betas", "eps", "weight_decay", "amsgrad"}
        if fused_ok:  # amsgrad and every torch lr scheduler stay on the one-launch path
            self.optimizer = FusedAdam(self.flat, lr=cfg.learning_rate, betas=tuple(params.get("betas", (0.9, 0.999))),
                                       eps=params.get("eps", 1e-8), weight_decay=params.get("weight_decay", 0.0),
                                       amsgrad=params.get("amsgrad", False),
                                       zero_grad_in_step=self._fused_zero_grad())
        else:
            import torch.optim as optim

            self.optimizer = getattr(optim, cfg.optimizer_cls)(self.model.parameters(), lr=cfg.learning_rate, **params)

    def _fused_zero_grad(self):
        """Every step here is zero_grad -> backward -> step, so the optimizer launch may clear the gradients it consumes;
        user callbacks could write gradients in between: then only when the configuration asks for it."""
        want = self.training_config.fused_zero_grad
        if want is None:
            want = all(type(cb) is TrainingCallback for cb in self.callbacks)
        return bool(want)

    def set_scheduler(self):
        cfg = self.training_config
        if cfg.scheduler_cls is None:
            self.scheduler = None
            return
        import torch.optim.lr_scheduler as lr_scheduler

        self.scheduler = getattr(lr_scheduler, cfg.scheduler_cls)(self.optimizer, **(cfg.scheduler_params or {}))

    def _set_output_dir(self):
        cfg = self.training_config
        if cfg.output_dir is None:
            cfg.output_dir = "dummy_output_dir"
        os.makedirs(cfg.output_dir, exist_ok=True)
        sig = str(datetime.datetime.now())[0:19].replace(" ", "_").replace(":", "-")
        self._training_signature = sig
        self.training_dir = os.path.join(cfg.output_dir, f"{self.model_name}_training_{sig}")
        if self.is_main_process:
            os.makedirs(self.training_dir, exist_ok=True)

    def prepare_training(self):
        set_seed(self.training_config.seed)
        self.set_optimizer()
        self.set_scheduler()
        self._set_output_dir()
        self.callback_handler = CallbackHandler(callbacks=self.callbacks, model=self.model)
        self.callback_handler.add_callback(MetricConsolePrinterCallback())
        self.trained_epochs = 0
        self.best_train_loss = float("inf")
        self.best_eval_loss = float("inf")
        self.metrics_best_model = {}
        self._best_model = deepcopy(self.model)

    def resume_training(self, checkpoint):
        """Continue from a `checkpoint_epoch_N` folder (base_trainer.py:402-440), written by this trainer or by the
        reference's: optimizer / scheduler state, best losses, trained epochs and the training directory."""
        with open(os.path.join(checkpoint, "info_checkpoint.json"), "r") as fp:
            info = json.load(fp)
        with open(os.path.join(checkpoint, "metrics_best_model.json"), "r") as fp:
            self.metrics_best_model = json.load(fp)
        set_seed(self.training_config.seed)
        self.set_optimizer()
        self.optimizer.load_state_dict(torch.load(os.path.join(checkpoint, "optimizer.pt"), map_location=self.device))
        self.set_scheduler()
        if self.scheduler is not None:
            self.scheduler.load_state_dict(torch.load(os.path.join(checkpoint, "scheduler.pt"),
                                                      map_location=self.device))
        self.training_dir = info["training_dir"]
        if self.is_main_process:
            os.makedirs(self.training_dir, exist_ok=True)
        self.callback_handler = CallbackHandler(callbacks=self.callbacks, model=self.model)
        self.callback_handler.add_callback(MetricConsolePrinterCallback())
        self.trained_epochs = info["trained_epochs"]
        self.best_train_loss = info["best_train_loss"]
        self.best_eval_loss = info["best_eval_loss"]
        self._best_model = deepcopy(self.model)

    # -- one optimizer step -------------------------------------------------------------------------------------
    def _optimizers_step(self, model_output, backward_done=False):
        """zero_grad -> backward -> [one all-reduce] -> step  (base_trainer.py:350-361)."""
        if not backward_done:
            loss = model_output.loss
            # always through the flat buffer: torch's own zero_grad() sets .grad to None, backward() would then
            # allocate gradients OUTSIDE the buffer the all-reduce below exchanges
            self.flat.zero_grad()
            with kernels.deferred_reductions(self.flat):  # one launch finishes every weight / bias gradient
                # the registered unit seed (cuda): the loss node's backward then needs no launch (kernels.unit_seed)
                loss.backward(gradient=kernels.unit_seed(loss) if loss.is_cuda else None)
            if not isinstance(self.optimizer, FusedAdam):
                self.flat.ensure_attached()
        if isinstance(self.optimizer, FusedAdam):
            # C2: ONE collective over the flat gradient buffer — mvk_allreduce_avg (RCCL) leaves the mean, a plain sum (gloo) the
            # factor 1 / world_size for the optimizer launch
            self.optimizer.step(grad_scale=self.flat.all_reduce_mean() if self.distributed else 1.0)
        else:
            if self.distributed:
                self.flat.all_reduce()
                self.flat.grad.mul_(1.0 / self.world_size)
            self.optimizer.step()

    # -- hipGraph replay of the fixed-shape part of a step (training_config.use_hip_graph) ----------------------
    def _graphed_forward_backward(self, inputs, epoch, fwd_kwargs):
        """zero_grad + forward + backward through a captured hipGraph when this batch shape (and the model's
        graph_key, e.g. JMVAE's annealing epoch) has one; returns None when the step has to run eagerly."""
        if not (self.training_config.use_hip_graph and isinstance(self.optimizer, FusedAdam)
                and self.device.type == "cuda" and not hasattr(inputs, "masks")):
            return None
        from ..graph import GraphedStep

        cfg = self.training_config
        key_fn = getattr(self.model, "graph_key", None)
        model_key = key_fn(**fwd_kwargs) if key_fn else None
        if model_key is False:  # this step is not replayable (e.g. MVAE while its KL weight changes every batch)
            return None
        key = (tuple((m, tuple(v.shape)) for m, v in inputs.data.items()), model_key)
        graphs = self.__dict__.setdefault("_graphs", {})
        gs = graphs.get(key)
        if gs is None and key not in graphs:
            try:
                self._drain_rotation()  # a capture warms up with eager passes: nothing may be pending
                in_graph = bool(cfg.graph_optimizer and not self.distributed)
                rotate = bool(getattr(cfg, "rotate_step", False) and not self.distributed and not in_graph
                              and getattr(self.optimizer, "zero_grad_in_step", False))
                gs = GraphedStep(self.model, self.flat, inputs, noise=None,
                                 capture_error_mode="thread_local" if self.distributed else "global",
                                 optimizer=self.optimizer if in_graph else None, rotate=self.optimizer if rotate else None,
                                 overlap=bool(cfg.overlap_collective and self.distributed), **fwd_kwargs)
            except Exception as e:  # not capturable (host sync inside the model, ...): stay eager for this shape
                logger.warning(f"hipGraph capture failed ({type(e).__name__}: {e}); running this batch shape eagerly")
                gs = None
            graphs[key] = gs
        if gs is None:
            return None
        if self.__dict__.get("_rot_pending") is not gs:
            self._drain_rotation()  # another graph's (batch shape's) rotated update is pending: apply it first
        out = gs(inputs)
        if gs.rotated:
            self._rot_pending = gs
        # what is left of the step: nothing (single GPU: the optimizer is the graph's last node), or the overlapped collective +
        # Adam (data parallel)
        self._graph_tail = (lambda: None) if gs.includes_optimizer else \
            ((lambda: gs.reduce_and_step(self.optimizer)) if (self.distributed and gs.early_ranges) else None)
        return out

    def _drain_rotation(self):
        """Rotated steps (training_config.rotate_step): apply the update the last replay left pending — before an eager step,
        a replay of another captured shape, and at the end of an epoch (evaluation, checkpoints and callbacks read the
        parameters)."""
        gs = self.__dict__.pop("_rot_pending", None)
        if gs is not None:
            gs.drain()

    def train_step(self, epoch: int):
        self.callback_handler.on_train_step_begin(training_config=self.training_config, train_loader=self.train_loader,
                                                  epoch=epoch, rank=self.rank)
        self.model.train()
        cfg = self.training_config
        sync = cfg.sync_every_step
        epoch_loss = 0.0 if sync else torch.zeros((), dtype=torch.float64, device=self.device)
        epoch_model_metrics = {}
        n_batches = len(self.train_loader)
        for batch_idx, inputs in enumerate(self.train_loader):
            beta_epoch = cfg.beta_schedule[epoch - 1] if hasattr(cfg, "beta_schedule") else 1
            fwd_kwargs = dict(epoch=epoch, dataset_size=len(self.train_dataset), uses_ddp=self.distributed,
                              batch_ratio=batch_idx / n_batches, beta=beta_epoch)
            model_output = self._graphed_forward_backward(inputs, epoch, fwd_kwargs)
            if model_output is not None:  # gradients are already in the flat buffer
                tail = self.__dict__.pop("_graph_tail", None)
                if tail is not None:
                    tail()
                else:
                    self._optimizers_step(model_output, backward_done=True)
            else:
                self._drain_rotation()
                model_output = self.model(inputs, **fwd_kwargs)
                self._optimizers_step(model_output)
            loss = model_output.loss_sum if hasattr(model_output, "loss_sum") else model_output.loss
            if sync:
                epoch_loss += loss.item()
                if epoch_loss != epoch_loss:
                    raise ArithmeticError("NaN detected in train loss")
            else:
                epoch_loss += loss.detach().double()
            update_dict(epoch_model_metrics, {k: (v.detach() if torch.is_tensor(v) else v)
                                              for k, v in model_output.metrics.items()})
            self.callback_handler.on_train_step_end(training_config=cfg)
        self._drain_rotation()
        self.model.update()
        if not sync:
            epoch_loss = float(epoch_loss.item())  # the one host sync of the epoch
            if epoch_loss != epoch_loss:
                raise ArithmeticError("NaN detected in train loss")
        epoch_model_metrics = {k: epoch_model_metrics[k] / n_batches for k in epoch_model_metrics}
        epoch_loss = epoch_loss / len(self.train_dataset)  # full dataset length, also under DDP (:748)
        return epoch_loss, epoch_model_metrics

    def eval_step(self, epoch: int):
        self.callback_handler.on_eval_step_begin(training_config=self.training_config, eval_loader=self.eval_loader,
                                                 epoch=epoch, rank=self.rank)
        self.model.eval()
        epoch_loss = torch.zeros((), dtype=torch.float64, device=self.device)
        epoch_metrics = {}
        n_batches = len(self.eval_loader)
        with torch.no_grad():
            for batch_idx, inputs in enumerate(self.eval_loader):
                out = self.model(inputs, epoch=epoch, dataset_size=len(self.eval_dataset), uses_ddp=self.distributed,
                                 batch_ratio=batch_idx / n_batches)
                loss = out.loss_sum if hasattr(out, "loss_sum") else out.loss
                epoch_loss += loss.double()
                update_dict(epoch_metrics, out.metrics)
                self.callback_handler.on_eval_step_end(training_config=self.training_config)
        epoch_loss = float(epoch_loss.item())
        if epoch_loss != epoch_loss:
            raise ArithmeticError("NaN detected in eval loss")
        epoch_metrics = {k: epoch_metrics[k] / n_batches for k in epoch_metrics}
        return epoch_loss / len(self.eval_dataset), epoch_metrics

    # -- main loop -----------------------------------------------------------------------------------------------
    def train(self):
        if self.checkpoint is None:
            self.prepare_training()
        else:
            self.resume_training(self.checkpoint)
        cfg = self.training_config
        self.callback_handler.on_train_begin(training_config=cfg, model_config=self.model_config)
        history = []
        for epoch in range(self.trained_epochs + 1, cfg.num_epochs + 1):
            self.callback_handler.on_epoch_begin(training_config=cfg, epoch=epoch)
            metrics = {}
            epoch_train_loss, epoch_metrics = self.train_step(epoch)
            metrics["train_epoch_loss"] = epoch_train_loss
            metrics.update({"train_" + k: (float(v) if torch.is_tensor(v) else v) for k, v in epoch_metrics.items()})
            epoch_eval_loss = None
            if self.eval_dataset is not None:
                epoch_eval_loss, eval_metrics = self.eval_step(epoch)
                metrics["eval_epoch_loss"] = epoch_eval_loss
                metrics.update({"eval_" + k: (float(v) if torch.is_tensor(v) else v) for k, v in eval_metrics.items()})
                if self.scheduler is not None:
                    self._schedulers_step(epoch_eval_loss)
            elif self.scheduler is not None:
                self._schedulers_step(epoch_train_loss)
            if epoch_eval_loss is None:
                epoch_eval_loss = self.best_eval_loss  # base_trainer.py:528
            if epoch <= self.start_keep_best_epoch:  # e.g. JMVAE's warm-up: keep the latest model (:534-539)
                self._best_model = deepcopy(self.model)
                self.metrics_best_model = metrics
            elif epoch_eval_loss < self.best_eval_loss and not cfg.keep_best_on_train:
                self.best_eval_loss = epoch_eval_loss
                self._best_model = deepcopy(self.model)
                self.metrics_best_model = metrics
            elif epoch_train_loss < self.best_train_loss and cfg.keep_best_on_train:
                self.best_train_loss = epoch_train_loss
                self._best_model = deepcopy(self.model)
                self.metrics_best_model = metrics
            self.callback_handler.on_epoch_end(training_config=cfg)
            if cfg.steps_saving is not None and epoch % cfg.steps_saving == 0 and self.is_main_process:
                self.save_checkpoint(model=self._best_model, dir_path=self.training_dir, epoch=epoch)
                self.callback_handler.on_save(cfg)
            self.callback_handler.on_log(cfg, metrics, logger=logger, global_step=epoch, rank=self.rank)
            history.append(metrics)
        final_dir = os.path.join(self.training_dir, "final_model")
        if self.is_main_process:
            self.save_model(self._best_model, dir_path=final_dir)
        if self.distributed:
            if self.flat is not None:
                self.flat.close()  # the RCCL communicator of mvk_allreduce_avg, before the process group it was built through
            dist.destroy_process_group()
        self.callback_handler.on_train_end(cfg)
        self.history = history
        return history

    def _schedulers_step(self, metrics=None):
        import torch.optim.lr_scheduler as lr_scheduler

        if isinstance(self.scheduler, lr_scheduler.ReduceLROnPlateau):
            self.scheduler.step(metrics)
        else:
            self.scheduler.step()

    # -- persistence ------------------------------------------------------------------------------------------------
    def save_model(self, model: BaseModel, dir_path: str):
        os.makedirs(dir_path, exist_ok=True)
        model.save(dir_path)
        self.training_config.save_json(dir_path, "training_config")
        with open(os.path.join(dir_path, "metrics_best_model.json"), "w") as fp:
            json.dump(self.metrics_best_model, fp)
        self.callback_handler.on_save(self.training_config)

    def save_checkpoint(self, model: BaseModel, dir_path, epoch: int):
        """checkpoint_epoch_N/{model.pt, optimizer.pt [, scheduler.pt], model_config.json, training_config.json,
        environment.json, metrics_best_model.json, info_checkpoint.json} (base_trainer.py:777-828), same keys and
        layouts, so a checkpoint resumes in either trainer."""
        ckpt = os.path.join(dir_path, f"checkpoint_epoch_{epoch}")
        os.makedirs(ckpt, exist_ok=True)
        torch.save(deepcopy(self.optimizer.state_dict()), os.path.join(ckpt, "optimizer.pt"))
        if self.scheduler is not None:
            torch.save(deepcopy(self.scheduler.state_dict()), os.path.join(ckpt, "scheduler.pt"))
        model.save(ckpt)
        self.training_config.save_json(ckpt, "training_config")
        with open(os.path.join(ckpt, "metrics_best_model.json"), "w") as fp:
            json.dump(self.metrics_best_model, fp)
        info = dict(training_dir=self.training_dir, trained_epochs=epoch, best_train_loss=self.best_train_loss,
                    best_eval_loss=self.best_eval_loss)
        with open(os.path.join(ckpt, "info_checkpoint.json"), "w") as fp:
            json.dump(info, fp, sort_keys=True, indent=4)
