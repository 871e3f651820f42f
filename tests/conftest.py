import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# modules that load librpf_hip.so (the product); the rest of the CPU suite — oracle known answers, the oracle against
# the committed goldens, the CPU twin of the device arithmetic, gloo sharding — needs gcc/g++ only
_NEEDS_HIP_LIB = {"test_abi", "test_gpu_parity", "test_host_state", "test_legacy_flavour", "test_no_gpu_behaviour",
                  "test_ppm", "test_preview", "test_shard", "test_unpack"}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    for item in items:
        if item.module.__name__.rsplit(".", 1)[-1] in _NEEDS_HIP_LIB:
            item.add_marker(pytest.mark.usefixtures("hip_lib"))


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the checker (oracle + CPU twin of the device arithmetic) once per session."""
    import oracle
    oracle.build()
    from tests import util
    util.build_twin()
    yield


@pytest.fixture(scope="session")
def hip_lib():
    """The product library, built in-tree on first use.  On a machine without ROCm (no hipcc, no prebuilt .so) the
    tests that bind it are skipped with this message instead of erroring; on the GPU box it must exist."""
    from rawphotoforge_amd import build as b
    if not os.path.exists(b.LIB) and not (shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))):
        pytest.skip("librpf_hip.so is not built and hipcc is not installed: tests of the HIP library skipped")
    return b.build()


@pytest.fixture(scope="session")
def gpu(hip_lib):
    import rawphotoforge_amd as R
    return R.GpuProcessor(0)
