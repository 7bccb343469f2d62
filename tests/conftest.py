import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def nof():
    """The loaded HIP library on a GPU box; GPU tests fail loudly if it is missing."""
    import torch
    from bundlesdf_amd import lib
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    lib.load()
    return lib
