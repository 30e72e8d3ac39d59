import os
import sys

import pytest

# The experiment switches of the library are honoured only under MVK_TUNE=1 (multivae_amd/_lib.py, csrc/common.hpp).  The
# suite sets it so that the tests which select a secondary kernel on purpose can (test_small_up_fwd_bwd runs the exact-fp32
# image-layer backward, which is also the kernel of every shape the split-bf16 one does not cover); no switch is set by
# default, so every other test runs the shipped configuration.
os.environ.setdefault("MVK_TUNE", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
