import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def reference_binning():
    """Bin with the reference's 3-sigma bounding square (exact tile culling off) for the duration of a test: the per-tile
    instance lists and num_rendered are then bit-comparable with the oracle's restatement of rasterizer_impl.cu."""
    from s3gaussian_amd import raster_C
    prev = raster_C.set_exact_cull(False)
    yield
    raster_C.set_exact_cull(prev)
