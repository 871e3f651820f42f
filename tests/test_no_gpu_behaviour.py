"""Without a HIP device nothing computes: the processor fails with the reference's
GpuInitializationError (gpu_image_processing.rs:48-51), it never falls back to the CPU."""
import ctypes as C

import pytest

import rawphotoforge_amd as R
from rawphotoforge_amd import _native as N


def _have_gpu():
    n = C.c_int(0)
    return N.lib().rpf_device_count(C.byref(n)) == N.OK and n.value > 0


@pytest.mark.skipif(_have_gpu(), reason="a GPU is present")
def test_processor_creation_fails_without_gpu():
    with pytest.raises(R.GpuInitializationError):
        R.GpuProcessor(0)
    assert R.GpuProcessor.get_adapter_string_list() == []


def test_out_of_range_adapter_index():
    with pytest.raises(R.GpuInitializationError):
        R.GpuProcessor(4096)
