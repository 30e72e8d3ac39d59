"""Configuration dataclasses (pydantic), JSON round-trip compatible with the reference's layout:
`multivae/models/base/base_config.py:8-75` on top of `pythae.config.BaseConfig` (name field = class name)."""
import dataclasses
import json
import os
from dataclasses import field
from typing import Dict, Literal, Optional, Tuple, Union

from pydantic.dataclasses import dataclass


@dataclass
class BaseConfig:
    name: str = field(init=False, default="BaseConfig")

    def __post_init__(self):
        self.name = self.__class__.__name__

    @classmethod
    def from_dict(cls, config_dict):
        d = dict(config_dict)
        d.pop("name", None)
        # a null in the file means "the default" (fields like `output_dir: str = None` do not validate an explicit None)
        defaults = {f.name: f.default for f in dataclasses.fields(cls)}
        d = {k: v for k, v in d.items() if not (v is None and defaults.get(k, 0) is None)}
        return cls(**d)

    @classmethod
    def from_json_file(cls, json_path):
        with open(json_path) as f:
            d = json.load(f)
        name = d.pop("name", None)
        if name is not None and name != cls.__name__:
            raise ValueError(f"config file is for {name}, not {cls.__name__}")
        return cls.from_dict(d)

    def to_dict(self):
        return dataclasses.asdict(self)

    def to_json_string(self):
        return json.dumps(self.to_dict())

    def save_json(self, dir_path, filename):
        with open(os.path.join(dir_path, f"{filename}.json"), "w", encoding="utf-8") as f:
            f.write(self.to_json_string())


@dataclass
class BaseMultiVAEConfig(BaseConfig):
    n_modalities: int = None
    latent_dim: int = 10
    input_dims: Optional[dict] = None
    uses_likelihood_rescaling: bool = False
    rescale_factors: Optional[dict] = None
    decoders_dist: Union[Dict[str, Literal["normal", "bernoulli", "laplace", "categorical"]], None] = None
    decoder_dist_params: Union[dict, None] = None
    custom_architectures: list = field(default_factory=lambda: [])

    def __post_init__(self):
        super().__post_init__()
        if self.input_dims is not None:
            self.input_dims = {k: tuple(self.input_dims[k]) for k in self.input_dims}


@dataclass
class EnvironmentConfig(BaseConfig):
    python_version: str = "3.8"


@dataclass
class BaseAEConfig(BaseConfig):
    input_dim: Union[Tuple[int, ...], None] = None
    latent_dim: int = 10
    style_dim: int = 0
