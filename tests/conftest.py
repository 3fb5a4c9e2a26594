import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def pytest_collection_modifyitems(config, items):
    """The two-process HIP-engine test starts fresh child processes that open the GPU themselves: run it before
    any test of this process has initialised the GPU (children of a GPU-initialised parent work too on this
    image -- tests/test_gpu_fuzz.py starts one -- but first is the conservative order)."""
    first = [it for it in items if "test_two_process_hip_engine" in it.nodeid]
    if first:
        rest = [it for it in items if it not in first]
        items[:] = first + rest
