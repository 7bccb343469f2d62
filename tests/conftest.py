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


# Collection order of the GPU suite: the driver runs `pytest -x`, so whatever is collected after the first failure is not evidence.
# Cheap per-operation parity first (seconds each), whole-step / renderer / runner next, the multi-minute full-size and two-process
# data-parallel tests last: a late failure then costs the fewest rows.  Files not listed keep their alphabetical place in the middle.
_ORDER = ['test_gpu_erratum', 'test_gpu_ops', 'test_gpu_tiles', 'test_gpu_rays', 'test_gpu_reference_fixture', 'test_gpu_texture',
          'test_gpu_mesh', 'test_gpu_chain', 'test_gpu_step', 'test_gpu_render', 'test_gpu_runner']
_LAST = ['test_gpu_fullsize', 'test_gpu_dp']


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if mod in _ORDER:
            return (0, _ORDER.index(mod))
        if mod in _LAST:
            return (2, _LAST.index(mod))
        return (1, 0)
    items.sort(key=key)                        # stable: the order inside a file is untouched
