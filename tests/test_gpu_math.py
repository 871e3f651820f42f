"""Device functions of rpf_math.h without a bit-identical host twin (they start from gfx950's approximate
v_rsq_f32), checked ON THE GPU over whole ranges of binary32 inputs (run on the GPU box: -m gpu)."""
import ctypes as C

import pytest

from tests.util import build_checker  # lives in a module without a pytest import: __graft_entry__.build() uses it too

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def checker():
    lib = C.CDLL(build_checker())
    lib.gpu_check_sqrt.restype = C.c_int
    lib.gpu_check_sqrt.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    lib.gpu_hue_index.restype = C.c_int
    lib.gpu_hue_index.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
    lib.gpu_check_pow_1p5.restype = C.c_int
    lib.gpu_check_pow_1p5.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    return lib


def test_sqrt_cr_equals_ieee_sqrt_on_every_binary32(checker):
    """sqrt_cr = one Markstein step on v_rsq_f32 inside [2^-96, 2^96), the IEEE sequence elsewhere: every one of
    the 2^32 bit patterns (negative, NaN, inf, zeros, subnormals included) must give the bits of sqrtf."""
    total = 0
    for lo in range(0, 1 << 32, 1 << 30):
        bad, first = C.c_uint64(0), C.c_uint32(0)
        assert checker.gpu_check_sqrt(lo, 1 << 30, C.byref(bad), C.byref(first)) == 0
        assert bad.value == 0, f"{bad.value} mismatches in [{lo:#x}, +2^30), first at bits {first.value:#010x}"
        total += 1 << 30
    assert total == 1 << 32


def test_pow_1p5_equals_the_binary64_formulation_on_its_whole_domain(checker):
    """pow_1p5 on the device (sqrt_cr's Markstein step + two correction terms, five binary64 operations) against
    (float)(x * sqrt(x)) in binary64 — the host build, which tests/test_math_exhaustive.py pins to the oracle — for
    x = 0 and every binary32 in [2^-96, 1]: the vignette's falloff argument is 0 or >= 3.9e-8 (wgsl:170-171)."""
    bad, first = C.c_uint64(0), C.c_uint32(0)
    assert checker.gpu_check_pow_1p5(0, 1, C.byref(bad), C.byref(first)) == 0 and bad.value == 0      # x = +0
    lo, hi = 0x0f800000, 0x3f800000                                                                    # 2^-96 .. 1.0
    assert checker.gpu_check_pow_1p5(lo, hi - lo + 1, C.byref(bad), C.byref(first)) == 0
    assert bad.value == 0, f"{bad.value} mismatches, first at bits {first.value:#010x}"


def test_hue_index_on_the_device_matches_the_oracle(checker):
    """The device build of hue_index differs from the host twin in one place, the hardware reciprocal inside the
    candidate, and the candidate now also decides on its own when it is 3/16 or more from an integer: random directions
    over the chain's magnitudes, and true positions on both sides of that threshold around every boundary, through the
    device code against the oracle's correctly rounded atan2 pipeline."""
    import numpy as np
    import oracle as O
    from tests import util
    tw, orc = util.twin(), O.lib()
    c, s = C.c_double(), C.c_double()
    bounds = np.empty((65536, 2))
    for k in range(65536):
        tw.twin_hue_bound(C.c_uint32(k), C.byref(c), C.byref(s))
        bounds[k] = (c.value, s.value)

    def both(y, x, neg_zero_x=0):
        y, x = np.ascontiguousarray(y, np.float32), np.ascontiguousarray(x, np.float32)
        a, b = np.empty(y.size, np.uint32), np.empty(y.size, np.uint32)
        assert checker.gpu_hue_index(y.ctypes.data, x.ctypes.data, y.size, bounds.ctypes.data, neg_zero_x, a.ctypes.data) == 0
        orc.orc_hue_index_many(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_size_t(y.size),
                               b.ctypes.data_as(C.c_void_p))
        return a, b

    rng = np.random.default_rng(31)
    n = 16_000_000
    th = rng.uniform(-np.pi, np.pi, n)
    r = 10.0 ** rng.uniform(-8, 0.5, n)
    for nz in (0, 1):
        a, b = both(r * np.sin(th), r * np.cos(th), nz)
        assert np.array_equal(a, b), np.nonzero(a != b)[0][:10]
    j = np.arange(65536, dtype=np.float64)
    for radius in (0.8, 1e-3):
        for off in (-0.49, -0.2, -0.19, -0.1875, -0.18, -0.05, -1e-4, 1e-4, 0.05, 0.18, 0.1875, 0.19, 0.2, 0.49):
            ang = (j + off) * (2 * np.pi / 65535.0)
            rr = radius * rng.uniform(0.5, 1.0, ang.size)
            a, b = both(rr * np.sin(ang), rr * np.cos(ang))
            assert np.array_equal(a, b), (radius, off, np.nonzero(a != b)[0][:10])


def test_hue_candidate_error_over_every_binary32_ratio(checker):
    """the sparse decision of hue_index (a candidate 3/16 or more away from the integer it rounds to settles
    its side of the boundary by itself) needs |candidate - true position| + |boundary j - position j| < 0.1875, and that
    was only sampled.  The candidate is a function of q = min / max in [0, 1] and of the octant, so it can be swept: EVERY
    binary32 q in [0, 1] through the device code (1 065 353 217 values), then the rest bounded analytically —
      reciprocal (1 ulp) and the product mn * rcp: the slope of the index scale in q is <= 65535 / (2 pi) = 10430 and
        q / (1 + q^2) <= 1/2, so <= 10430 * 0.5 * 1.5 * 2^-23 = 0.001;
      three reflections (16383.75 - v, 32767.5 - v, 65535 - v), each one rounding of a value < 65536: <= 3 * 2^-9 = 0.006;
      boundary j against position j: 0.007 (measured over all 65536 boundaries, tests/test_math_exhaustive.py)."""
    lib = checker
    lib.gpu_hue_candidate_sweep.restype = C.c_int
    lib.gpu_hue_candidate_sweep.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_float)]
    worst = 0.0
    one = 0x3f800000
    for lo in range(0, one + 1, 1 << 28):
        n = min(1 << 28, one + 1 - lo)
        e = C.c_float(0)
        assert lib.gpu_hue_candidate_sweep(lo, n, C.byref(e)) == 0
        worst = max(worst, e.value)
    total = worst + 0.001 + 0.006 + 0.007
    print(f"hue candidate: max polynomial error {worst:.5f} index units over every binary32 q in [0, 1]; total bound {total:.5f} < 0.1875")
    assert worst < 0.17 and total < 0.1875, (worst, total)
