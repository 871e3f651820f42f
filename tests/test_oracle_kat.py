"""Known-answer tests that pin the oracle (the reference has no tests of its own,
so these are the analytic cases derivable from its source)."""
import hashlib
import os

import numpy as np
import pytest

import oracle as O
from tests import util


def _px(rgb, params=None, **kw):
    img = np.array([[list(rgb) + [1.0]]], np.float32)
    return O.apply(img, params or [O.EditParameters()], **kw)[0, 0]


def test_black_pixel_stays_black():
    # KAT 1: (0,0,0) with black = 0 -> (0,0,0,1); atan2(+0,+0) = 0, pow(0, 1/3) = 0
    assert _px((0, 0, 0)).tolist() == [0.0, 0.0, 0.0, 1.0]
    p = O.EditParameters(exposure=3.0, contrast=50, shadow=100, highlight=-100, white=40)
    assert _px((0, 0, 0), [p])[:3].tolist() != [np.nan] * 3


def test_defaults_are_not_identity():
    # KAT 2: brightness LUT floors to the 16-bit grid and sat/light gain is 32767/32767.5 (wgsl:293-330)
    out = _px((0.5, 0.5, 0.5))
    ident = O.lib().orc_linear_to_srgb1(0.5)
    assert out[0] != ident and abs(out[0] - ident) < 1e-4
    assert np.float32(32767) / np.float32(32767.5) == np.float32(0.99998474)


def test_lut_entries_are_clamped_at_fetch():
    # KAT 3: wgsl:188 clamp(curve[i], 0, 65535)
    p, q = O.EditParameters(), O.EditParameters()
    p.brightness_tone_curve = np.full(65536, 70000, np.int32)
    q.brightness_tone_curve = np.full(65536, 65535, np.int32)
    assert _px((0.2, 0.4, 0.6), [p]).tolist() == _px((0.2, 0.4, 0.6), [q]).tolist()
    p.brightness_tone_curve = np.full(65536, -5, np.int32)
    q.brightness_tone_curve = np.zeros(65536, np.int32)
    assert _px((0.2, 0.4, 0.6), [p]).tolist() == _px((0.2, 0.4, 0.6), [q]).tolist()


def test_mask_not_exactly_one_skips_both_loops():
    # KAT 4: wgsl:281, :314 `mask_value != 1.0 -> continue`
    img = np.array([[[0.3, 0.5, 0.7, 1.0], [0.3, 0.5, 0.7, 1.0]]], np.float32)
    main, extra = O.EditParameters(), O.EditParameters(exposure=1.0, contrast=30)
    extra.saturation_tone_curve = np.full(65536, 50000, np.int32)
    masks = np.array([[[1.0, 1.0]], [[1.0, 0.99999994]]], np.float32)
    out = O.apply(img, [main, extra], masks=masks)
    base = O.apply(img, [main])
    assert out[0, 1].tolist() == base[0, 1].tolist()
    assert out[0, 0].tolist() != base[0, 0].tolist()


def test_negative_linear_rgb_lands_on_zero_not_nan():
    # KAT 5: saturated colours leave the sRGB gamut after the OKLCH round trip; <= 0.0031308 branch
    p = O.EditParameters()
    p.saturation_tone_curve = np.full(65536, 65535, np.int32)
    out = _px((0.0, 0.0, 1.0), [p])
    assert np.isfinite(out).all() and out.min() == 0.0 and out.max() <= 1.0


def test_pchip_errors():
    # KAT 6: interpolation.rs:22-32, 48-52
    with pytest.raises(O.PchipError) as e:
        O.pchip_interpolate([0], [0], [0, 1])
    assert e.value.kind == "NotEnoughPoints"
    with pytest.raises(O.PchipError) as e:
        O.pchip_interpolate([0, 5, 5, 9], [0, 1, 2, 3], [0])
    assert e.value.kind == "NotStrictlyIncreasing" and e.value.index == 1
    with pytest.raises(O.PchipError) as e:
        O.pchip_interpolate([0, 1, 2], [0, 1], [0])
    assert e.value.kind == "MismatchedLengths"


def test_pchip_clamps_outside_range():
    # KAT 7: interpolation.rs:82-89
    out = O.pchip_interpolate([100, 200, 300], [7, 50, 9], [-5, 0, 100, 300, 301, 65535])
    assert out.tolist() == [7, 7, 7, 9, 9, 9]


def test_pchip_two_point_tables():
    # KAT 8: the UI's "identity" and "constant 32767" control-point curves are NOT the defaults
    ident = O.curve_from_control_points([0, 65535], [0, 65535])
    i = np.arange(65536)
    assert int((ident == i).sum()) == 49741 and int((ident == i - 1).sum()) == 15795
    const = O.curve_from_control_points([0, 65535], [32767, 32767])
    vals, counts = np.unique(const, return_counts=True)
    assert vals.tolist() == [32766, 32767] and counts.tolist() == [2532, 63004]
    # exact numpy-float32 cross-check of the 2-point evaluation (t = x/65535, cubic Hermite)
    x = i.astype(np.float32)
    h = np.float32(65535.0)
    t = x / h
    t2 = t * t
    t3 = t2 * t
    one, two, three = np.float32(1), np.float32(2), np.float32(3)
    h00 = two * t3 - three * t2 + one
    h10 = t3 - two * t2 + t
    h01 = -two * t3 + three * t2
    h11 = t3 - t2
    y = h00 * np.float32(0) + h10 * h * one + h01 * h + h11 * h * one
    y[0], y[-1] = 0, 65535
    assert np.array_equal(np.clip(y.astype(np.int32), 0, 65535), ident)


def test_param_conversion():
    # gpu_image_processing.rs:235-247
    g = O.convert_params(O.EditParameters(exposure=1.5, contrast=20, shadow=-30, highlight=40, black=5, white=-5,
                                          wb_temperature=15, wb_tint=-10, vignette=-30, lens_distortion=7))
    f = np.float32
    assert g.r_gain == f(1) + f(0.5) * (f(15) / f(100))
    assert g.g_gain == f(1) - f(0.25) * (f(-10) / f(100))
    assert g.b_gain == f(1) - f(0.5) * (f(15) / f(100))
    assert (g.vignette, g.lens_distortion, g.exposure) == (-30.0, 7.0, 1.5)
    assert g.contrast == f(20) / f(100) and g.white == f(-5) / f(100)


def test_mask_binarise():
    # lib.rs:483-486 with the default mask_range 0.0: everything >= 0 becomes 1.0
    m = np.array([-1.0, -0.0, 0.0, 0.3, 1.0], np.float32)
    assert O.binarize_mask(m).tolist() == [0, 1, 1, 1, 1]
    assert O.binarize_mask(m, 0.5).tolist() == [0, 0, 0, 0, 1]


def test_correctly_rounded_builtins_against_mpmath():
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    rng = np.random.default_rng(11)
    third = float(np.float32(1.0 / 3.0))
    for x in rng.random(300).astype(np.float32):
        want = np.float32(float(mp.mpf(float(x)) ** mp.mpf(third)))  # mpf -> float is correctly rounded twice; check vs f64 route
        got = O.lib().orc_cr_powf(float(x), third)
        exact = mp.mpf(float(x)) ** mp.mpf(third)
        lo, hi = np.nextafter(np.float32(got), np.float32(-np.inf)), np.nextafter(np.float32(got), np.float32(np.inf))
        assert abs(exact - mp.mpf(float(got))) <= min(abs(exact - mp.mpf(float(lo))), abs(exact - mp.mpf(float(hi))))
        del want


def test_vignette_uses_global_coordinates_and_tiles_agree():
    img = np.random.default_rng(2).random((16, 12, 4), dtype=np.float32)
    p = O.EditParameters(vignette=-60)
    whole = O.apply(img, [p])
    top = O.apply(img[:6], [p], frame_hw=(16, 12), y0=0)
    bot = O.apply(img[6:], [p], frame_hw=(16, 12), y0=6)
    assert np.array_equal(np.concatenate([top, bot]), whole)
    assert not np.array_equal(whole, O.apply(img, [O.EditParameters()]))


def test_demosaic_flat_field_and_patterns():
    # a flat white-balanced field demosaics to the same value at every site (MHC kernels sum to 1)
    rp = O.RawParams(wb=(1, 1, 1, 1), cam2rgb=(1, 0, 0, 0, 1, 0, 0, 0, 1))
    cfa = np.full((12, 10), 8192, np.uint16)
    out = O.demosaic(cfa, rp)
    v = np.float32(8192 - 512) * (np.float32(1) / np.float32(16383 - 512))
    assert np.allclose(out[..., :3], v, rtol=0, atol=1e-6) and (out[..., 3] == 1).all()
    # the four CFA layouts are shifts of one another
    rng = np.random.default_rng(3)
    big = rng.integers(512, 16384, (20, 22), dtype=np.uint16)
    base = O.demosaic(big, O.RawParams(cfa_pattern=O.CFA_RGGB))
    g = O.demosaic(big[:, 1:], O.RawParams(cfa_pattern=O.CFA_GRBG, wb=(2.0, 1.0, 1.0, 1.5)))
    assert np.array_equal(g[4:-4, 4:-4], base[4:-4, 5:-4])
    b = O.demosaic(big[1:, 1:], O.RawParams(cfa_pattern=O.CFA_BGGR))
    assert np.array_equal(b[4:-4, 4:-4], base[5:-4, 5:-4])


def test_demosaic_row_tiles_equal_whole_frame():
    cfa = util.bayer_uniform(37, 29)
    rp = util.raw_params()
    whole = O.demosaic(cfa, rp)
    parts = [O.demosaic(cfa, rp, y0=0, rows=10), O.demosaic(cfa, rp, y0=10, rows=13), O.demosaic(cfa, rp, y0=23)]
    assert np.array_equal(np.concatenate(parts), whole)


def test_oracle_matches_committed_golden():
    g = np.load(os.path.join(util.GOLDEN, "chain_golden.npz"))
    rp = util.raw_params()
    for tag, gen in (("U", util.bayer_uniform), ("N", util.bayer_natural)):
        assert np.array_equal(gen(64, 64), g[f"cfa64_{tag}"])
        assert np.array_equal(O.demosaic(gen(64, 64), rp), g[f"demosaic64_{tag}"])
        for pname, pf in (("D", util.params_D), ("S", util.params_S)):
            assert np.array_equal(O.develop(gen(64, 64), rp, [pf()]), g[f"develop64_{tag}_{pname}"])
            big = O.develop(gen(256, 256), rp, [pf()])
            assert np.array_equal(np.frombuffer(hashlib.sha256(big.tobytes()).digest(), np.uint8),
                                  g[f"sha256_develop256_{tag}_{pname}"])
