import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Multi-rank GPU tests (test_gpu_partition.py) fork their ranks from a helper process that is started HERE, before
# anything in this process can have touched the GPU: a process that has initialised HIP must not fork+exec (the GPU
# boxes refuse it), while children forked from the clean helper initialise the GPU themselves.
try:
    import multiprocessing.forkserver as _forkserver
    _forkserver.ensure_running()
except Exception:                                      # no forkserver on this platform: those tests skip
    _forkserver = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))
    return load
