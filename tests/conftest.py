import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=["fp32", "bf16x6"])
def gemm_mode(request):
    """Run a test under both arithmetic modes of the projection GEMMs (exact fp32 MFMA / split-bf16 emulation);
    the SAME tolerances apply to both."""
    from wsi_hgnn_amd import ops
    ops.set_gemm_precision(request.param)
    yield request.param
    ops.set_gemm_precision("fp32")
