"""ModelOutput (attribute-access ordered dict) and small host-side helpers.

Mirrors what the reference takes from pythae (`pythae.models.base.base_utils.ModelOutput`) and
`multivae/models/base/base_utils.py:62-87` (set_decoder_dist validation).  The arithmetic of poe /
kl_divergence / rsample lives in the HIP kernels (multivae_amd/csrc/elbo.hip).
"""
from ... import _lib
from ..._output import ModelOutput  # noqa: F401  (re-exported: `multivae.models.base.base_utils.ModelOutput`)


def decoder_dist_code(dist_name):
    """'normal' | 'laplace' | 'bernoulli' | 'categorical' -> MVK_DIST_*; anything else is rejected like the reference
    (`set_decoder_dist` raises ValueError, base_utils.py:84-85).  'categorical' takes tensors: logits [..., n_classes]
    against one-hot (or probability) targets of the same trailing shape."""
    if dist_name in _lib.DIST:
        return _lib.DIST[dist_name]
    raise ValueError("The distribution type 'dist' is not supported")
