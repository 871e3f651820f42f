import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the checker (oracle + CPU twin) and the product library once per session."""
    import oracle
    oracle.build()
    from tests import util
    util.build_twin()
    from rawphotoforge_amd import build as b
    b.build()
    yield


@pytest.fixture(scope="session")
def gpu():
    import rawphotoforge_amd as R
    return R.GpuProcessor(0)
