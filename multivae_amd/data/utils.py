"""Batch helpers (`multivae/data/utils.py:7-64`).  Unlike the reference (which only moves data when
`device == "cuda"` as a string, SURVEY.md §8e trap 3) every tensor of the selected keys is moved."""
import torch

from .datasets.base import DatasetOutput


def _to(obj, device):
    if torch.is_tensor(obj):
        return obj.to(device, non_blocking=True)
    if isinstance(obj, dict):
        return {k: _to(v, device) for k, v in obj.items()}
    return obj


def set_inputs_to_device(inputs, device="cpu", keys=None):
    keys = list(inputs.keys()) if keys is None else keys
    out = {k: (_to(inputs[k], device) if k in keys else inputs[k]) for k in inputs.keys()}
    return DatasetOutput(**out)


def get_batch_size(inputs):
    k = list(inputs.data.keys())[0]
    return len(inputs.data[k])


def drop_unused_modalities(inputs):
    """Drop modalities that are unavailable for the whole batch (data/utils.py:53-64).  Costs one host sync."""
    if not hasattr(inputs, "masks"):
        return inputs
    for m in list(inputs.masks.keys()):
        if not bool(torch.any(inputs.masks[m])):
            inputs.data.pop(m)
            inputs.masks.pop(m)
    return inputs
