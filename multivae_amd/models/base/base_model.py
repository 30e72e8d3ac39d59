"""BaseModel: nn.Module + on-disk format of the reference (`multivae/models/base/base_model.py:27-211`):
`model.pt` = {"model_state_dict": state_dict}, `model_config.json`, `environment.json`, custom
architectures cloudpickled as `<name>.pkl`."""
import os
import sys
from copy import deepcopy

import torch
import torch.nn as nn

from .base_config import BaseConfig, EnvironmentConfig


class BaseModel(nn.Module):
    def __init__(self, model_config: BaseConfig):
        nn.Module.__init__(self)
        self.model_name = "BaseModel"
        self.model_config = model_config
        self.model_custom_architectures = []

    def forward(self, inputs, **kwargs):
        raise NotImplementedError()

    def __call__(self, *args, **kwargs):
        # one weight-pack launch for the whole forward pass (kernels.pack_scope; a no-op on the CPU oracle-less paths)
        from ... import kernels

        with kernels.pack_scope(self):
            return super().__call__(*args, **kwargs)

    def update(self):
        pass

    def save(self, dir_path: str):
        env_spec = EnvironmentConfig(python_version=f"{sys.version_info[0]}.{sys.version_info[1]}")
        model_dict = {"model_state_dict": deepcopy(self.state_dict())}
        os.makedirs(dir_path, exist_ok=True)
        env_spec.save_json(dir_path, "environment")
        self.model_config.save_json(dir_path, "model_config")
        torch.save(model_dict, os.path.join(dir_path, "model.pt"))
        for archi in self.model_config.custom_architectures:
            try:
                import cloudpickle

                with open(os.path.join(dir_path, archi + ".pkl"), "wb") as fp:
                    cloudpickle.dump(getattr(self, archi), fp)
            except Exception:  # same tolerance as the reference: the state_dict is what matters
                pass

    @classmethod
    def _load_model_weights_from_folder(cls, dir_path):
        path = os.path.join(dir_path, "model.pt")
        if not os.path.exists(path):
            raise FileNotFoundError(f"Missing model weights file ('model.pt') file in {dir_path}")
        blob = torch.load(path, map_location="cpu")
        if "model_state_dict" not in blob:
            raise KeyError("Model state dict is not available in 'model.pt' file.")
        return blob["model_state_dict"]

    @classmethod
    def _load_custom_archi_from_folder(cls, dir_path, archi):
        import pickle

        path = os.path.join(dir_path, archi + ".pkl")
        if not os.path.exists(path):
            raise FileNotFoundError(f"Missing architecture pkl file ('{archi}.pkl') in {dir_path}")
        with open(path, "rb") as fp:
            return pickle.load(fp)

    @classmethod
    def load_from_folder(cls, dir_path: str):
        """Rebuild a model saved by `save` (or by the reference, same layout)."""
        cfg_path = os.path.join(dir_path, "model_config.json")
        if not os.path.exists(cfg_path):
            raise FileNotFoundError(f"Missing model config file ('model_config.json') in {dir_path}")
        import inspect

        config_cls = inspect.signature(cls.__init__).parameters["model_config"].annotation
        model_config = config_cls.from_json_file(cfg_path)
        state = cls._load_model_weights_from_folder(dir_path)
        kwargs = {}
        for archi in list(model_config.custom_architectures):
            kwargs[archi] = cls._load_custom_archi_from_folder(dir_path, archi)
        model_config.custom_architectures = []
        model = cls(model_config, **kwargs)
        model.load_state_dict(state)
        return model
