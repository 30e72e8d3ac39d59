"""Reload any saved model / configuration from its folder (`multivae/models/auto_model/auto_model.py:41-98`,
`auto_config.py`): the class is picked from the `name` field of `model_config.json`.  Folders written by the
reference load here and folders written by `BaseModel.save` load in the reference (tests/golden/checkpoint_compat.py)."""
import json
import os

from .. import CRMVAE, DMVAE, DMVAEConfig, JMVAE, MMVAE, MVAE, MVTCAE, CRMVAEConfig, JMVAEConfig, MVAEConfig, MMVAEConfig, MMVAEPlus, MMVAEPlusConfig, MoPoE, MoPoEConfig, MVTCAEConfig
from ..base import BaseMultiVAEConfig

_MODELS = {"JMVAEConfig": (JMVAE, JMVAEConfig), "MMVAEConfig": (MMVAE, MMVAEConfig), "MoPoEConfig": (MoPoE, MoPoEConfig),
           "MVTCAEConfig": (MVTCAE, MVTCAEConfig), "MVAEConfig": (MVAE, MVAEConfig), "CRMVAEConfig": (CRMVAE, CRMVAEConfig), "DMVAEConfig": (DMVAE, DMVAEConfig), "MMVAEPlusConfig": (MMVAEPlus, MMVAEPlusConfig)}


def _name(json_path):
    with open(json_path) as f:
        return json.load(f)["name"]


class AutoConfig:
    @classmethod
    def from_json_file(cls, json_path):
        name = _name(json_path)
        if name == "BaseMultiVAEConfig":
            return BaseMultiVAEConfig.from_json_file(json_path)
        if name not in _MODELS:
            raise NameError("Cannot reload automatically the model configuration... The model name in the "
                            f"`model_config.json may be corrupted. Got `{name}`")
        return _MODELS[name][1].from_json_file(json_path)


class AutoModel:
    @classmethod
    def load_from_folder(cls, dir_path: str):
        name = _name(os.path.join(dir_path, "model_config.json"))
        if name not in _MODELS:
            raise NameError("Cannot reload automatically the model... The model name in the `model_config.json may be "
                            f"corrupted. Got {name}")
        return _MODELS[name][0].load_from_folder(dir_path)
