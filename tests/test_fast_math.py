"""RPF_PRECISION_FAST (rawphotoforge_amd/csrc/rpf_fast.h, rpf_chain.h edit_chain_fast): the binary32-only builtins against the
oracle's correctly rounded ones, on the CPU twin.  The pows, the sine and the cosine are IEEE operations only, so what is
checked here are the device's bits; the hue position and the square root use a hardware reciprocal / square root on the
device (tests/test_gpu_fast.py measures those there)."""
import ctypes as C
import struct

import numpy as np

import oracle as O
from tests import util


def _bits(f):
    return struct.unpack("I", struct.pack("f", f))[0]


def _pow_histogram(which, y, lo, hi):
    tw, orc = util.twin(), O.lib()
    b0, b1 = _bits(lo), _bits(hi)
    hist, chunk = np.zeros(4, np.int64), 1 << 24
    for s in range(b0, b1, chunk):
        n = min(chunk, b1 - s)
        a, b = np.empty(n, np.float32), np.empty(n, np.float32)
        tw.twin_fast_pow_range(C.c_int(which), C.c_uint32(s), C.c_uint32(n), a.ctypes.data_as(C.c_void_p))
        orc.orc_cr_powf_range(C.c_uint32(s), C.c_uint32(n), C.c_float(y), b.ctypes.data_as(C.c_void_p))
        hist += np.bincount(np.minimum(util.ulp_diff(a, b), 3), minlength=4)
    return b1 - b0, hist


def test_fast_cube_root_every_float_in_the_lms_domain():
    # the same domain as the exact function's sweep (tests/test_math_exhaustive.py): within 1 ulp of the correctly rounded
    # value everywhere, and correctly rounded itself on all but a fraction of a percent
    n, hist = _pow_histogram(0, float(np.float32(1.0 / 3.0)), 2.0 ** -21, 1.25)
    assert n > 170_000_000 and hist[2] == 0 and hist[3] == 0
    assert hist[1] < 0.005 * n, hist


def test_fast_srgb_upper_branch_every_float_in_the_gamma_domain():
    """srgb_upper_fast = 1.055 pow(x, f32(1 / 2.4)) - 0.055 as one table evaluation, on every binary32 of [2^-9, 8) (the branch
    is taken above 0.0031308; above 1 the chain clamps): within 1 ulp of the branch's true value (binary64 here, rounded
    once).  What the reference's own three roundings give (the oracle's correctly rounded pow, a binary32 product, a binary32
    difference) is further from that true value than this function is — near the threshold the difference 0.078 - 0.055
    cancels and the three roundings amount to several ulp of the result — so against it: equal or 1 ulp away for 97 % of the
    inputs, 3 ulp or more for 0.3 %."""
    tw, orc = util.twin(), O.lib()
    c = float(np.float32(1.0 / 2.4))
    b0, b1 = _bits(2.0 ** -9), _bits(8.0)
    hist, hist3, chunk = np.zeros(4, np.int64), np.zeros(4, np.int64), 1 << 24
    for s in range(b0, b1, chunk):
        n = min(chunk, b1 - s)
        got, pw = np.empty(n, np.float32), np.empty(n, np.float32)
        tw.twin_fast_pow_range(C.c_int(1), C.c_uint32(s), C.c_uint32(n), got.ctypes.data_as(C.c_void_p))
        orc.orc_cr_powf_range(C.c_uint32(s), C.c_uint32(n), C.c_float(c), pw.ctypes.data_as(C.c_void_p))
        x = np.arange(s, s + n, dtype=np.uint32).view(np.float32).astype(np.float64)
        true = (float(np.float32(1.055)) * x ** c - float(np.float32(0.055))).astype(np.float32)
        three = np.float32(1.055) * pw - np.float32(0.055)
        hist += np.bincount(np.minimum(util.ulp_diff(got, true), 3), minlength=4)
        hist3 += np.bincount(np.minimum(util.ulp_diff(got, three), 3), minlength=4)
    n = b1 - b0
    assert n > 100_000_000 and hist[2] == 0 and hist[3] == 0 and hist[1] < 0.01 * n, hist
    assert hist3[0] + hist3[1] > 0.96 * n and hist3[3] < 0.005 * n, hist3


def test_fast_pow_outside_the_tables():
    tw = util.twin()
    tw.twin_fast_pow_cbrt.restype = tw.twin_fast_srgb_upper.restype = C.c_float
    tw.twin_fast_pow_cbrt.argtypes = tw.twin_fast_srgb_upper.argtypes = [C.c_float]
    # zero, subnormals and everything below 2^-21 (the chain's cube-root arguments are 0 or >= 7e-7): 0
    for x in (0.0, 1e-45, 1e-39, 1e-10, 4.7e-7, -0.0):
        assert tw.twin_fast_pow_cbrt(x) == 0.0, x
    # the sRGB upper branch at and above 8: anything >= 1 will do (the chain clamps to 1, wgsl:335)
    for x in (8.0, 9.5, 31.0, 100.0, 1e4):
        assert tw.twin_fast_srgb_upper(x) >= 1.0, x
    # below 2^-9 its result is unspecified but finite (the chain takes the linear branch, wgsl:99-102)
    for x in (0.0, 1e-30, 1e-3, -1.0, -1e-5):
        assert np.isfinite(tw.twin_fast_srgb_upper(x)), x


def test_fast_sincos_every_reachable_hue():
    tw, orc = util.twin(), O.lib()
    cs, sn = np.empty(65536, np.float32), np.empty(65536, np.float32)
    tw.twin_fast_sincos_all(cs.ctypes.data_as(C.c_void_p), sn.ctypes.data_as(C.c_void_p))
    wc, ws = np.empty(65536, np.float32), np.empty(65536, np.float32)
    a, b = C.c_float(), C.c_float()
    for k in range(65536):
        orc.orc_hue_sincos(C.c_uint32(k), C.byref(a), C.byref(b))
        wc[k], ws[k] = a.value, b.value
    dc, ds = util.ulp_diff(cs, wc), util.ulp_diff(sn, ws)
    assert dc.max() <= 1 and ds.max() <= 1, (int(dc.max()), int(ds.max()))
    assert (dc == 0).mean() > 0.8 and (ds == 0).mean() > 0.8
    # signs of the zeros and of the tiny values next to the axes
    assert cs[0] == 1.0 and sn[0] == 0.0 and not np.signbit(sn[0])
    assert np.array_equal(np.signbit(cs), np.signbit(wc)) and np.array_equal(np.signbit(sn), np.signbit(ws))


def test_hue_index_survives_the_round_trip_through_the_hue():
    # a user mask re-derives its curve index from the hue the previous mask left, u32(f32(k / 65535) * 65535)
    # (wgpu_shader.wgsl:320): k again for every k — edit_chain_fast carries the integer instead
    k = np.arange(65536, dtype=np.float32)
    h = k / np.float32(65535.0)
    assert np.array_equal((h * np.float32(65535.0)).astype(np.uint32), np.arange(65536, dtype=np.uint32))


def test_fast_hue_index_flip_rate():
    """The hue index from the binary32 polynomial alone: the oracle's index, or — within a few thousandths of an index unit
    of a boundary — its neighbour.  (Host form: an IEEE division where the device multiplies by v_rcp_f32.)"""
    tw, orc = util.twin(), O.lib()
    rng = np.random.default_rng(5)
    n = 4_000_000
    th, rad = rng.uniform(0, 2 * np.pi, n), 10.0 ** rng.uniform(-4, 0, n)
    x, y = (rad * np.cos(th)).astype(np.float32), (rad * np.sin(th)).astype(np.float32)
    v, k = np.empty(n, np.float32), np.empty(n, np.uint32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    tw.twin_fast_hue_position_many(P(y), P(x), C.c_size_t(n), P(v))
    orc.orc_hue_index_many(P(y), P(x), C.c_size_t(n), P(k))
    assert v.min() >= 0.0 and v.max() <= 65535.0
    d = np.trunc(v).astype(np.int64) - k.astype(np.int64)
    d = np.where(d > 32768, d - 65536, np.where(d < -32768, d + 65536, d))
    assert np.abs(d).max() <= 1
    assert (d != 0).mean() < 0.002, (d != 0).mean()
    # the axes and the diagonals, both signs of zero
    for yy, xx, want in ((0.0, 1.0, 0), (0.0, 0.0, 0), (-0.0, 1.0, 0), (1.0, 0.0, 16383), (0.0, -1.0, 32767), (-1.0, 0.0, 49151)):
        a, b, o = np.float32([yy]), np.float32([xx]), np.empty(1, np.float32)
        tw.twin_fast_hue_position_many(P(a), P(b), C.c_size_t(1), P(o))
        assert int(np.trunc(o[0])) == want, (yy, xx, o[0])


def test_fast_hue_fraction():
    """hue_fraction_fast: the fractional part of the hue-index position (what the rotation form of the chain turns (a, b)
    through), against binary64: within 0.003 index units (3e-7 rad) all round the circle, in [0, 1)."""
    tw = util.twin()
    rng = np.random.default_rng(8)
    n = 2_000_000
    th, rad = rng.uniform(0, 2 * np.pi, n), 10.0 ** rng.uniform(-4, 0, n)
    x, y = (rad * np.cos(th)).astype(np.float32), (rad * np.sin(th)).astype(np.float32)
    f = np.empty(n, np.float32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    tw.twin_fast_hue_fraction_many(P(y), P(x), C.c_size_t(n), P(f))
    assert f.min() >= 0.0 and f.max() < 1.0
    pos = np.arctan2(y.astype(np.float64), x.astype(np.float64)) / np.float64(np.float32(2.0 * 3.14159265359))
    pos = np.where(pos < 0, pos + 1, pos) * 65535.0
    d = np.abs(f - (pos - np.floor(pos)))
    d = np.minimum(d, 1.0 - d)   # the fraction wraps at an index boundary
    assert d.max() < 0.003 and d.mean() < 0.001, (d.max(), d.mean())
    for yy, xx in ((0.0, 0.0), (0.0, 1.0), (-0.0, 1.0), (0.0, -1.0)):   # no direction / the axes: no NaN, in range
        a, b, o = np.float32([yy]), np.float32([xx]), np.empty(1, np.float32)
        tw.twin_fast_hue_fraction_many(P(a), P(b), C.c_size_t(1), P(o))
        assert 0.0 <= o[0] < 1.0


def _chain_stats(params, n, seed, mode=True):
    rng = np.random.default_rng(seed)
    rgb = rng.random((n, 3), dtype=np.float32)
    img = np.concatenate([rgb, np.ones((n, 1), np.float32)], 1).reshape(1, n, 4)
    want = O.apply(img, params).reshape(n, 4)[:, :3]
    got = util.twin_apply(rgb, params, fast=mode)[:, :3]
    return util.ulp_diff(got, want), np.abs(got.astype(np.float64) - want)


def test_fast_chain_against_the_oracle():
    """edit_chain_fast on uniformly random pixels, three edit sets (D and S take the rotation form, HUE_ROT; G the builtin
    route), and D again through the builtin route: what RPF_PRECISION_FAST promises in include/rpf.h — median at most 2 ulp
    per channel, at least 85 % of the pixels within 1e-6 and 99 % within 1e-5 of the oracle, the rest (where the hue index
    steps: the hue moves by one step of the index scale) within 4e-3 for smooth curves."""
    s = util.params_S()
    s.vignette = 0  # the vignette is applied in front of the chain, by the kernel
    for params, flip_bound, mode in (([util.params_D()], 4e-3, True), ([s], 4e-3, True), ([util.params_G()], 3e-2, True),   # G: steeper curves
                                     ([util.params_D()], 4e-3, "builtins")):
        ulp, ad = _chain_stats(params, 400_000, 11, mode)
        assert np.median(ulp) <= 2
        assert (ad.max(1) <= 1e-6).mean() >= 0.85, (ad.max(1) <= 1e-6).mean()
        assert (ad.max(1) <= 1e-5).mean() >= 0.99, (ad.max(1) <= 1e-5).mean()
        assert ad.max() <= flip_bound, ad.max()


def test_fast_chain_on_grey_pixels():
    """Grey pixels have (almost) no chroma, so a hue index that flips cannot show: what is left of the difference to the
    oracle are the cube roots, the gamma and the contracted matrix products — below 2e-6 everywhere."""
    n = 100_000
    rng = np.random.default_rng(3)
    rgb = np.repeat(rng.random((n, 1), dtype=np.float32), 3, 1)
    img = np.concatenate([rgb, np.ones((n, 1), np.float32)], 1).reshape(1, n, 4)
    p = [util.params_D()]
    want = O.apply(img, p).reshape(n, 4)[:, :3]
    got = util.twin_apply(rgb, p, fast=True)[:, :3]
    assert np.abs(got.astype(np.float64) - want).max() < 2e-6
