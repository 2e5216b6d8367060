import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import build

    build.build_oracle()
    from oracle import clib

    return clib


@pytest.fixture(scope="session")
def hip_lib():
    """libd3d_hip.so, built if missing (hipcc cross-compiles without a GPU)."""
    from det3d_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        from det3d_amd.csrc import build

        build.build(verbose=False)
    return _lib.lib()
