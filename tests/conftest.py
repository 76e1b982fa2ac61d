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


@pytest.fixture(params=["fp32", "bf16x6", "fp16x3", "auto"])
def gemm_mode(request):
    """Run a test under every arithmetic mode of the projection GEMMs (exact fp32 MFMA / split-bf16 / scaled split-fp16
    emulation, and 'auto' = the library's per-launch choice between the two emulations); the SAME tolerances apply to all."""
    from wsi_hgnn_amd import ops
    ops.set_gemm_precision(request.param)
    yield request.param
    ops.set_gemm_precision("fp32")
