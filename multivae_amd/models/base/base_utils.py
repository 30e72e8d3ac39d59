"""ModelOutput (attribute-access ordered dict) and small host-side helpers.

Mirrors what the reference takes from pythae (`pythae.models.base.base_utils.ModelOutput`) and
`multivae/models/base/base_utils.py:62-87` (set_decoder_dist validation).  The arithmetic of poe /
kl_divergence / rsample lives in the HIP kernels (multivae_amd/csrc/elbo.hip).
"""
from collections import OrderedDict

from ... import _lib


class ModelOutput(OrderedDict):
    """Ordered dict whose items are also attributes; integer indexing returns the i-th value."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return super().__getitem__(k)
        return list(self.values())[k]

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        super().__setattr__(key, value)

    def __setattr__(self, name, value):
        super().__setitem__(name, value)
        super().__setattr__(name, value)

    def __delitem__(self, key):
        super().__delitem__(key)
        if key in self.__dict__:
            super().__delattr__(key)

    def pop(self, key, *default):
        if key in self.__dict__:
            super().__delattr__(key)
        return super().pop(key, *default)


def decoder_dist_code(dist_name):
    """'normal' | 'laplace' | 'bernoulli' -> MVK_DIST_*; anything else is rejected like the reference
    (`set_decoder_dist` raises ValueError, base_utils.py:84-85).  'categorical' is not on the HIP path yet."""
    if dist_name in _lib.DIST:
        return _lib.DIST[dist_name]
    if dist_name == "categorical":
        raise NotImplementedError("decoder distribution 'categorical' has no HIP kernel yet")
    raise ValueError("The distribution type 'dist' is not supported")
