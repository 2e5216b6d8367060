import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# GPU run order (`pytest -m gpu -x` stops at the first failure, so what proves the kernels goes first): kernel-level
# parity files, then the model-level goldens, then the distributed paths, the optional features (hipGraph tail) and,
# last, the long training-run comparisons of test_precision_gpu (8 of the suite's 12.5 minutes).  Files not listed keep their alphabetical place between the kernel files and the end-to-end ones.
_GPU_ORDER = ["test_voxelize_gpu", "test_spconv_gpu", "test_determinism_gpu", "test_conv2d_gpu", "test_nms_gpu",
              "test_iou3d_gpu", "test_loss_gpu", "test_optim_gpu", "test_pfn_gpu", "test_assign_gpu", "test_pipeline_gpu",
              "test_e2e_gpu", "test_dist_gpu", "test_graph_tail_gpu", "test_precision_gpu"]


def _order_key(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _GPU_ORDER:
        return _GPU_ORDER.index(name)
    return _GPU_ORDER.index("test_e2e_gpu") - 0.5 if name.endswith("_gpu") else -1


def pytest_collection_modifyitems(config, items):
    import torch

    items.sort(key=_order_key)  # stable: the order inside a file is kept
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
