"""ModelOutput: attribute-access ordered dict (what the reference takes from pythae).  Leaf module: no package
imports, so both `multivae_amd.models` and `multivae_amd.data` can use it without an import cycle."""
from collections import OrderedDict


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
