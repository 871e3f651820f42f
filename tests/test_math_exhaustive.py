"""The transcendental code the kernels run (rpf_math.h, IEEE-only, hence bit-identical on host and
device) against the oracle's correctly rounded builtins — exhaustively where the domain allows."""
import ctypes as C
import struct

import numpy as np

import oracle as O
from tests import util


def _bits(f):
    return struct.unpack("I", struct.pack("f", f))[0]


def _sweep(which, y, lo, hi):
    tw, orc = util.twin(), O.lib()
    b0, b1 = _bits(lo), _bits(hi)
    bad, chunk = [], 1 << 24
    for s in range(b0, b1, chunk):
        n = min(chunk, b1 - s)
        a, b = np.empty(n, np.float32), np.empty(n, np.float32)
        tw.twin_pow_range(C.c_int(which), C.c_uint32(s), C.c_uint32(n), a.ctypes.data_as(C.c_void_p))
        orc.orc_cr_powf_range(C.c_uint32(s), C.c_uint32(n), C.c_float(y), b.ctypes.data_as(C.c_void_p))
        bad += [s + int(i) for i in np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]]
    return b1 - b0, bad


def test_pow_one_third_every_float_in_the_lms_domain():
    # lms = M1 * rgb with rgb = k/65535: 0 or within [2^-21, 1.0000001]  (wgsl:65-67)
    n, bad = _sweep(0, float(np.float32(1.0 / 3.0)), 2.0 ** -21, 1.25)
    assert n > 170_000_000 and bad == []


def test_pow_inv_2p4_every_float_in_the_gamma_domain():
    # linear_to_srgb takes the pow branch above 0.0031308 > 2^-9 (wgsl:95-103); > 1 clamps to 1 afterwards
    n, bad = _sweep(1, float(np.float32(1.0 / 2.4)), 2.0 ** -9, 8.0)
    assert n > 100_000_000 and bad == []


def test_pow_edge_cases():
    tw = util.twin()
    assert tw.twin_pow_cbrt(0.0) == 0.0 and tw.twin_pow_gamma(0.0) == 0.0
    for x in (1e-30, 1e-12, 1.2e-38, 1e20, 3e38):   # normal numbers far outside the swept domains: same algorithm
        for f, c in ((tw.twin_pow_cbrt, np.float32(1 / 3)), (tw.twin_pow_gamma, np.float32(1 / 2.4))):
            assert util.ulp_diff(np.float32(f(x)), np.float32(O.lib().orc_cr_powf(x, float(c)))) <= 1
    # documented: a subnormal argument returns 0 (exponent field 0); the chain cannot produce one where the
    # result is used (rpf_math.h powc)
    assert tw.twin_pow_cbrt(3e-39) == 0.0 and tw.twin_pow_gamma(1e-45) == 0.0
    # a negative argument returns |x|^c (never used by the chain)
    assert tw.twin_pow_cbrt(-0.125) == tw.twin_pow_cbrt(0.125)


def test_sincos_every_reachable_hue():
    tw, orc = util.twin(), O.lib()
    a, b, c, d = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    for k in range(65536):
        tw.twin_sincos_hue(C.c_uint32(k), C.byref(a), C.byref(b))
        orc.orc_hue_sincos(C.c_uint32(k), C.byref(c), C.byref(d))
        assert (a.value, b.value) == (c.value, d.value), k


def test_division_by_constant_every_k():
    tw = util.twin()
    k = np.arange(65536, dtype=np.float32)
    for d in (65535.0, 32767.5):
        want = k / np.float32(d)
        got = np.array([tw.twin_div_u16(int(i), d) for i in range(65536)], np.float32)
        assert np.array_equal(got, want)


def test_atan2_random_and_special():
    tw, orc = util.twin(), O.lib()
    rng = np.random.default_rng(3)
    n = 200_000
    ys = (rng.standard_normal(n) * 10 ** rng.uniform(-9, 1, n)).astype(np.float32)
    xs = (rng.standard_normal(n) * 10 ** rng.uniform(-9, 1, n)).astype(np.float32)
    for y, x in zip(ys.tolist(), xs.tolist()):
        assert tw.twin_atan2(y, x) == orc.orc_cr_atan2f(y, x), (y, x)
    t = float(np.float32(0.41421356))
    for y, x in [(0, 0), (0, -0.0), (-0.0, 0.0), (-0.0, -0.0), (1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1),
                 (1e-20, 1), (1, 1e-20), (3e-39, 1e-39), (float("inf"), 1), (1, float("inf")), (t, 1), (1, t),
                 (1e38, 1e38), (-1e-31, 2e-31)]:
        g, w = tw.twin_atan2(y, x), orc.orc_cr_atan2f(y, x)
        assert g == w and np.signbit(g) == np.signbit(w), (y, x, g, w)


def test_pow_1p5_every_float_of_the_top_binades_and_samples():
    # vignette falloff pow(t, 1.5), t in [0, 1] (wgsl:171)
    n, bad = _sweep(4, 1.5, 2.0 ** -12, 1.0 + 2.0 ** -20)
    assert n > 100_000_000 and bad == []
    tw, orc = util.twin(), O.lib()
    rng = np.random.default_rng(4)
    xs = np.concatenate([(rng.random(100_000) * 10.0 ** rng.uniform(-30, 0, 100_000)).astype(np.float32),
                         np.float32([0, 1, 1e-10, 0.25, 1.2e-38, 1e-40])])
    for x in xs.tolist():
        assert tw.twin_pow_1p5(x) == orc.orc_cr_powf(x, 1.5), x


def test_division_by_two_pi_every_atan2_result():
    # Markstein sequence vs IEEE division for every binary32 in [0, 3.15] (atan2's range; the
    # sequence is odd in x, so the negative half follows)
    tw = util.twin()
    d = np.float32(2.0 * 3.14159265359)
    lo, hi, chunk = 0, _bits(3.15), 1 << 25
    for s in range(lo, hi, chunk):
        n = min(chunk, hi - s)
        got = np.empty(n, np.float32)
        tw.twin_div_two_pi_range(C.c_uint32(s), C.c_uint32(n), got.ctypes.data_as(C.c_void_p))
        x = np.arange(s, s + n, dtype=np.uint32).view(np.float32)
        assert np.array_equal(got.view(np.uint32), (x / d).view(np.uint32)), hex(s)
    for x in (-1.0, -3.1415927, -1e-30, -1e-44):
        tw.twin_div_two_pi.restype = C.c_float
        tw.twin_div_two_pi.argtypes = [C.c_float]
        assert np.float32(tw.twin_div_two_pi(x)) == np.float32(x) / d


# ---- Godot-legacy shader flavour (rust-godot-legacy/photo-editor/src/wgpu_shader.wgsl:89-97, 173-186) ----

def test_pow_2p4_every_float_in_the_srgb_decode_domain():
    # srgb_to_linear takes the pow branch above 0.04045: (c + 0.055) / 1.055 > 0.0904 > 2^-4; up to 2.0 for HDR input
    n, bad = _sweep(2, float(np.float32(2.4)), 2.0 ** -4, 2.0)
    assert n > 40_000_000 and bad == []


def test_pow_4_is_exact():
    # error-free by construction; every float of the top four binades of [0, 1], samples of the rest
    n, bad = _sweep(3, 4.0, 2.0 ** -4, 1.0 + 2.0 ** -20)
    assert n > 33_000_000 and bad == []
    tw, orc = util.twin(), O.lib()
    rng = np.random.default_rng(11)
    xs = np.concatenate([(rng.random(50_000) * 10.0 ** rng.uniform(-9, 0, 50_000)).astype(np.float32),
                         np.float32([0.0, 1.0, 1e-12, 1e-30, 0.5, 0.70710677])])
    for x in xs.tolist():
        assert tw.twin_pow_4(x) == orc.orc_cr_powf(x, 4.0), x


def test_srgb_to_linear_matches_oracle_on_every_16_bit_and_8_bit_code():
    tw, orc = util.twin(), O.lib()
    orc.orc_srgb_to_linear1.restype = C.c_float
    orc.orc_srgb_to_linear1.argtypes = [C.c_float]
    codes = np.concatenate([np.arange(65536, dtype=np.float32) / np.float32(65535.0),
                            np.arange(256, dtype=np.float32) / np.float32(255.0),
                            np.float32([0.04045, np.nextafter(np.float32(0.04045), np.float32(1)), 1.5, 0.0])])
    for c in codes.tolist():
        assert tw.twin_srgb_to_linear(c) == orc.orc_srgb_to_linear1(c), c


# ---- hue index without an arctangent (rpf_math.h hue_index) ----

def _hue_both(y, x):
    tw, orc = util.twin(), O.lib()
    y = np.ascontiguousarray(y, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    a, b = np.empty(y.size, np.uint32), np.empty(y.size, np.uint32)
    tw.twin_hue_index_many(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_size_t(y.size),
                           a.ctypes.data_as(C.c_void_p))
    orc.orc_hue_index_many(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_size_t(y.size),
                           b.ctypes.data_as(C.c_void_p))
    return a, b


def test_hue_index_random_directions_and_magnitudes():
    rng = np.random.default_rng(21)
    n = 6_000_000
    y = (rng.standard_normal(n) * 10.0 ** rng.uniform(-12, 2, n)).astype(np.float32)
    x = (rng.standard_normal(n) * 10.0 ** rng.uniform(-12, 2, n)).astype(np.float32)
    a, b = _hue_both(y, x)
    assert np.array_equal(a, b), np.nonzero(a != b)[0][:10]
    # what the chain produces: |a|, |b| <= ~0.5 with both near zero for greys
    th = rng.uniform(-np.pi, np.pi, n)
    r = 10.0 ** rng.uniform(-8, -0.3, n)
    a, b = _hue_both(r * np.sin(th), r * np.cos(th))
    assert np.array_equal(a, b)


def test_hue_index_next_to_every_boundary():
    # directions a few binary32 steps of the angle either side of each of the 65536 boundaries, at several radii
    tw = util.twin()
    cs = np.empty((65536, 2))
    c, s = C.c_double(), C.c_double()
    for j in range(65536):
        tw.twin_hue_bound(C.c_uint32(j), C.byref(c), C.byref(s))
        cs[j] = (c.value, s.value)
    theta = np.arctan2(cs[:, 1], cs[:, 0])
    assert abs(theta[1] - 2 * np.pi / 65535) < 1e-6 and abs(theta[65534] + 2 * np.pi / 65535) < 1e-6
    assert np.all(np.diff(np.where(theta < 0, theta + 2 * np.pi, theta)[:65535]) > 0)   # counter-clockwise order
    rng = np.random.default_rng(22)
    for radius in (1.0, 0.37, 3e-3, 2.5e-7):
        for steps in (-40, -3, -1, -0.5, -0.1, 0.1, 0.5, 1, 3, 40):
            # (boundary 0 is the angle 0 itself: step in units of 1e-30 there, clear of the documented -1e-45 corner)
            ulp = np.spacing(np.maximum(np.abs(theta), 1e-30).astype(np.float32)).astype(np.float64)
            t = theta + steps * ulp
            rr = radius * rng.uniform(0.5, 1.0, t.size)
            a, b = _hue_both(rr * np.sin(t), rr * np.cos(t))
            assert np.array_equal(a, b), (radius, steps, np.nonzero(a != b)[0][:10])


def test_hue_candidate_error_and_the_sparse_lookup_threshold():
    """hue_index looks a boundary up only when the candidate position is within 3/16 of an integer; everywhere else the
    candidate alone decides.  That needs |candidate - true position| < 3/16: measured here against binary64 (the
    device's v_rcp_f32 instead of the host's division moves it by ~1e-3), and the index itself is checked at true
    positions all around the threshold, on both sides of every boundary."""
    tw = util.twin()
    rng = np.random.default_rng(23)
    n = 8_000_000
    th = rng.uniform(-np.pi, np.pi, n)
    r = 10.0 ** rng.uniform(-8, 0.5, n)
    y, x = (r * np.sin(th)).astype(np.float32), (r * np.cos(th)).astype(np.float32)
    v = np.empty(n, np.float32)
    tw.twin_hue_candidate_many(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                               v.ctypes.data_as(C.c_void_p))
    t = np.arctan2(y.astype(np.float64), x.astype(np.float64)) / (2 * np.pi)
    u = np.where(t < 0, t + 1.0, t) * 65535.0
    err = np.abs(v.astype(np.float64) - u)
    err = np.minimum(err, 65535.0 - err)       # the wrap at angle 0
    assert err.max() < 0.125, err.max()
    # ... and boundary j sits at position j up to the reference pipeline's own roundings
    c, s = C.c_double(), C.c_double()
    pos = np.empty(65536)
    for k in range(65536):
        tw.twin_hue_bound(C.c_uint32(k), C.byref(c), C.byref(s))
        pos[k] = np.arctan2(s.value, c.value)
    pos = np.where(pos < 0, pos + 2 * np.pi, pos) * (65535.0 / (2 * np.pi))
    assert np.abs(pos - np.arange(65536)).max() < 0.01        # 0.125 + 0.01 + the device reciprocal's 1e-3 < 3/16
    # true positions j + off for every boundary j, off around +-3/16 and across the whole cell
    j = np.arange(65536, dtype=np.float64)
    for radius in (0.8, 1e-3):
        for off in (-0.49, -0.3, -0.2, -0.19, -0.1875, -0.18, -0.17, -0.05, 0.05, 0.17, 0.18, 0.1875, 0.19, 0.2, 0.3, 0.49):
            ang = (j + off) * (2 * np.pi / 65535.0)
            rr = radius * rng.uniform(0.5, 1.0, ang.size)
            a, b = _hue_both(rr * np.sin(ang), rr * np.cos(ang))
            assert np.array_equal(a, b), (radius, off, np.nonzero(a != b)[0][:10])


def test_hue_index_special_arguments():
    z, nz = 0.0, -0.0
    cases = [(z, z), (nz, z), (z, nz), (nz, nz), (z, 1), (nz, 1), (z, -1), (nz, -1), (1, z), (1, nz), (-1, z), (-1, nz),
             (1e-30, 1), (-1e-30, 1), (1e-30, -1), (-1e-30, -1), (1, 1e-30), (1, -1e-30), (1, 1), (-1, 1), (1, -1), (-1, -1),
             # the larger magnitude anywhere in the supported [2^-100, 2^100] (rpf_math.h hue_index_core; the chain's
             # |a|, |b| lie in {0} + [2^-75, 8)); the smaller one may be anything, subnormals included
             (3e-30, 1e-30), (-3e-30, 1e-30), (1e-30, -3e-30), (1e30, 1e30), (-1e29, 3e29), (8e-31, 8e-31), (-8e-31, -8e-31),
             (1e-45, 1e-30), (-1e-45, -1e30), (1e-39, 1.0), (1.0, -1e-42),
             (-1e-7, 1), (-1.9e-7, 1), (-2e-7, 1), (6e-8, -1), (-6e-8, -1)]
    y, x = np.float32([c[0] for c in cases]), np.float32([c[1] for c in cases])
    a, b = _hue_both(y, x)
    assert np.array_equal(a, b), [(cases[i], int(a[i]), int(b[i])) for i in np.nonzero(a != b)[0]]
    # the documented exception: a true angle in (-3.5 * 2^-149, 0) — unreachable from the chain
    a, b = _hue_both(np.float32([-1e-45]), np.float32([1.0]))
    assert (int(a[0]), int(b[0])) == (65535, 0)


def test_division_by_three_quarters_every_vignette_argument():
    # (dist - 0.25) / 0.75 (wgsl:171): Markstein sequence vs IEEE division for every binary32 in [0, 2] and [-0.25, -0]
    tw = util.twin()
    d = np.float32(0.75)
    for lo, hi in ((0, _bits(2.0)), (_bits(-0.0), _bits(-0.25) + 1)):
        chunk = 1 << 25
        for s in range(lo, hi, chunk):
            n = min(chunk, hi - s)
            got = np.empty(n, np.float32)
            tw.twin_div_const_range(C.c_float(0.75), C.c_uint32(s), C.c_uint32(n), got.ctypes.data_as(C.c_void_p))
            x = np.arange(s, s + n, dtype=np.uint32).view(np.float32)
            assert np.array_equal(got.view(np.uint32), (x / d).view(np.uint32)), hex(s)


def test_requantised_hue_index_is_the_identity():
    """wgsl:320 on a hue that came out of an identity hue curve: u32(f32(k / 65535) * 65535) == k for every k.  This is
    what lets HUE_DIRECT (cos / sin of the hue read with the index "main" settled) serve launches with user masks whose
    hue curves are all the identity: the index survives every mask's re-quantisation."""
    tw = util.twin()
    k = np.arange(65536, dtype=np.uint32)
    h = np.array([tw.twin_div_u16(C.c_uint32(int(i)), C.c_float(65535.0)) for i in k], np.float32)
    assert np.array_equal(h, (k.astype(np.float32) / np.float32(65535.0)).astype(np.float32))
    assert np.array_equal((h * np.float32(65535.0)).astype(np.float32).astype(np.uint32), k)
