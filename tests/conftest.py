import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session")
def lib():
    """The in-tree HIP library; building it needs hipcc but no GPU."""
    from tensoralloy_amd import _lib
    _lib.build()
    return _lib.load()
