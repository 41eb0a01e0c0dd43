import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library; GPU tests fail (not skip) when it is missing."""
    import torch
    from faster_rcnn_pytorch_multimodal_amd import _hip
    assert torch.cuda.is_available(), "GPU test selected but no GPU is visible"
    return _hip.load()
