"""Import the read-only reference (`/root/reference/src/multivae`) in THIS container only.

Used exclusively by `tests/golden/make_golden.py` to generate golden vectors.  It never
travels to the GPU box and nothing in `tests/`, `bench.py` or `__graft_entry__.py`
imports it at run time.

The reference depends on `pythae` (PyPI, not installed, no network).  On the hot path
pythae contributes plumbing only: `ModelOutput`/`DatasetOutput`
(attribute-access OrderedDicts), `BaseEncoder`/`BaseDecoder` (bare nn.Module bases),
`BaseConfig` (pydantic dataclass + JSON) and `CPU_Unpickler`.  This file fabricates
stand-ins for those names, and empty attribute-yielding modules for the third-party
packages the reference imports eagerly but does not use on the path
(torchvision, nltk, torchmetrics, pythae.trainers/samplers/normalizing_flows).
"""
import sys

sys.dont_write_bytecode = True  # keep /root/reference untouched

import dataclasses
import importlib.abc
import importlib.machinery
import json
import os
import pickle
import types
from collections import OrderedDict

import torch
import torch.nn as nn
from pydantic.dataclasses import dataclass as pyd_dataclass

REFERENCE_SRC = "/root/reference/src"


class ModelOutput(OrderedDict):
    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return list(self.values())[k]

    def __setattr__(self, name, value):
        super().__setitem__(name, value)
        super().__setattr__(name, value)

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        super().__setattr__(key, value)

    def __init__(self, *a, **kw):
        super().__init__()
        for k, v in dict(*a, **kw).items():
            self[k] = v


class DatasetOutput(ModelOutput):
    pass


class BaseEncoder(nn.Module):
    def __init__(self):
        nn.Module.__init__(self)


class BaseDecoder(nn.Module):
    def __init__(self):
        nn.Module.__init__(self)


class BaseMetric(nn.Module):
    def __init__(self):
        nn.Module.__init__(self)


class BaseDiscriminator(nn.Module):
    def __init__(self):
        nn.Module.__init__(self)


@pyd_dataclass
class BaseConfig:
    name: str = dataclasses.field(init=False, default="BaseConfig")

    def __post_init__(self):
        self.name = self.__class__.__name__

    @classmethod
    def from_dict(cls, d):
        d = dict(d)
        d.pop("name", None)
        return cls(**d)

    @classmethod
    def _dict_from_json(cls, path):  # pythae.config.BaseConfig._dict_from_json (used by multivae's AutoConfig)
        with open(path) as f:
            return json.load(f)

    @classmethod
    def from_json_file(cls, path):
        return cls.from_dict(cls._dict_from_json(path))

    def to_dict(self):
        return dataclasses.asdict(self)

    def to_json_string(self):
        return json.dumps(self.to_dict())

    def save_json(self, dir_path, filename):
        with open(os.path.join(dir_path, f"{filename}.json"), "w") as f:
            f.write(self.to_json_string())


@pyd_dataclass
class BaseAEConfig(BaseConfig):
    input_dim: object = None
    latent_dim: int = 10


class CPU_Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "torch.storage" and name == "_load_from_bytes":
            import io

            return lambda b: torch.load(io.BytesIO(b), map_location="cpu")
        return super().find_class(module, name)


class _Anything(types.ModuleType):
    """A module whose every attribute is a harmless placeholder class."""

    __path__ = []

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        placeholder = type(item, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, item, placeholder)
        return placeholder


_FAKE_PREFIXES = (
    "torchvision",
    "nltk",
    "torchmetrics",
    "pythae.trainers",
    "pythae.samplers",
    "pythae.models.normalizing_flows",
    "pythae.models.nn.benchmarks",
    "pythae.models.nn.default_architectures",
    "pythae.pipelines",
    "hostlist",
    "wandb",
    "mlflow",
    "matplotlib",
    "PIL",
    "imageio",
    "pandas_stub",
)


class _FakeFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if any(fullname == p or fullname.startswith(p + ".") for p in _FAKE_PREFIXES):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Anything(spec.name)

    def exec_module(self, module):
        pass


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    if "multivae" in sys.modules:
        return sys.modules["multivae"]
    _mod("pythae")
    _mod("pythae.config", BaseConfig=BaseConfig)
    _mod("pythae.models")
    _mod("pythae.models.base")
    _mod("pythae.models.base.base_config", BaseConfig=BaseConfig, BaseAEConfig=BaseAEConfig)
    _mod("pythae.models.base.base_utils", ModelOutput=ModelOutput, CPU_Unpickler=CPU_Unpickler)
    _mod("pythae.models.base.base_model", BaseDecoder=BaseDecoder, BaseEncoder=BaseEncoder)
    _mod("pythae.models.nn")
    _mod(
        "pythae.models.nn.base_architectures",
        BaseEncoder=BaseEncoder,
        BaseDecoder=BaseDecoder,
        BaseMetric=BaseMetric,
        BaseDiscriminator=BaseDiscriminator,
    )
    _mod("pythae.data")
    _mod(
        "pythae.data.datasets",
        Dataset=torch.utils.data.Dataset,
        BaseDataset=torch.utils.data.Dataset,
        DatasetOutput=DatasetOutput,
    )
    sys.meta_path.insert(0, _FakeFinder())
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import multivae  # noqa: F401

    return sys.modules["multivae"]
