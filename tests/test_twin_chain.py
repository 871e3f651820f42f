"""The kernels' per-pixel arithmetic (rpf_chain.h) and the product's host-side parameter packing
(rpf_host.h: make_mask_params) compiled for the CPU, against the oracle: bit-exact."""
import numpy as np

import oracle as O
from tests import util


def _pixels(n, seed):
    rng = np.random.default_rng(seed)
    rgb = rng.random((n, 3), dtype=np.float32) * np.float32(1.3) - np.float32(0.1)
    rgb[:2000] = rng.random((2000, 1), dtype=np.float32)  # greys: a, b ~ 0, hue ill-conditioned
    rgb[2000:2100] = 0
    rgb[2100:2200] = 1
    rgb[2200:2300] = np.float32([1, 0, 0])
    rgb[2300:2400] = np.float32([0, 0, 1])
    return rgb


def _check(masks_params, n=400_000, seed=5):
    rgb = _pixels(n, seed)
    img = np.concatenate([rgb, np.ones((n, 1), np.float32)], 1).reshape(1, n, 4)
    want = O.apply(img, masks_params).reshape(n, 4)
    got = util.twin_apply(rgb, masks_params)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_defaults():
    _check([util.params_D()])


def test_stress_set():
    p = util.params_S()
    p.vignette = 0  # the vignette is applied before the chain, by the kernel
    _check([p])


def test_general_curves():
    _check([util.params_G()])


def test_two_active_masks():
    s = util.params_S()
    s.vignette = 0
    _check([s, util.params_G()], n=200_000)


def test_tone_corner_values():
    for kw in (dict(exposure=10.0), dict(exposure=-10.0), dict(contrast=100), dict(contrast=-100),
               dict(shadow=100, highlight=-100), dict(black=100, white=-100), dict(black=-100, white=100),
               dict(wb_temperature=100, wb_tint=100), dict(wb_temperature=-100, wb_tint=-100)):
        _check([O.EditParameters(**kw)], n=50_000, seed=9)


def test_random_parameter_space():
    rng = np.random.default_rng(2026)
    for _ in range(30):
        n_masks = int(rng.integers(1, 4))
        _check([util.random_params(rng) for _ in range(n_masks)], n=20_000, seed=int(rng.integers(1 << 30)))


def test_fused_hue_table_equals_the_per_pixel_evaluation():
    """One mask: the chain with that mask's hue / saturation / lightness curves, their quotients and sincos_hue read from the
    fused hue table (edit_chain's HUE_MODE 2 — what the fused develop kernels run) against the oracle, bit for bit: set G,
    an identity hue curve with the other two left, constant gains, and random single-mask parameter sets."""
    import copy
    g = util.params_G()
    cases = [g]
    p = copy.deepcopy(g); p.hue_tone_curve = list(range(65536)); cases.append(p)
    p = copy.deepcopy(g); p.saturation_tone_curve = [40000] * 65536; p.lightness_tone_curve = [30000] * 65536; cases.append(p)
    rng = np.random.default_rng(99)
    cases += [util.random_params(rng, trivial_prob=0.2) for _ in range(12)]
    for i, p in enumerate(cases):
        n = 200_000 if i < 3 else 20_000
        rgb = _pixels(n, 40 + i)
        img = np.concatenate([rgb, np.ones((n, 1), np.float32)], 1).reshape(1, n, 4)
        want = O.apply(img, [p]).reshape(n, 4)
        got = util.twin_apply(rgb, [p], fused=True)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), i


def test_demosaic_formulas_with_carried_powers_of_two_equal_the_specification():
    """rpf_chain.h evaluates the four Malvar-He-Cutler values with the division by 8 carried into the terms and every
    'power of two times a term, plus a term' as one fma (19 instead of 29 instructions per pixel pair).  DESIGN.md §3 /
    orc_demosaic state them with one binary32 rounding per operation in source order: the two must agree bit for
    bit — on normalised samples as the kernels see them (n * scale, clamped), on their sums, and on zeros."""
    import ctypes as C
    rng = np.random.default_rng(12)
    n = 2_000_000
    f = np.float32
    scale = f(2.0) / f(16383 - 512)
    def samples(k):
        return (rng.integers(0, 15872, size=(k, n)).astype(f) * scale)
    v = samples(13)                       # the 13 taps of a 5x5 cross neighbourhood
    v[:, :1000] = 0                       # black areas
    v[:, 1000:2000] = v[0, 1000:2000]     # flat areas (exact cancellation)
    c = v[0]
    s1, e1, d = v[1] + v[2], v[3] + v[4], (v[5] + v[6]) + (v[7] + v[8])
    v2, h2 = v[9] + v[10], v[11] + v[12]
    want_cross = ((f(4) * c + f(2) * (s1 + e1)) - (v2 + h2)) * f(0.125)
    want_diag = ((f(6) * c + f(2) * d) - f(1.5) * (v2 + h2)) * f(0.125)
    want_h = ((((f(5) * c + f(4) * e1) - d) - h2) + f(0.5) * v2) * f(0.125)
    want_v = ((((f(5) * c + f(4) * s1) - d) - v2) + f(0.5) * h2) * f(0.125)
    outs = [np.empty(n, f) for _ in range(4)]
    P = lambda a: np.ascontiguousarray(a, f).ctypes.data_as(C.POINTER(C.c_float))
    util.twin().twin_mhc(C.c_size_t(n), P(c), P(s1), P(e1), P(d), P(v2), P(h2), *[o.ctypes.data_as(C.POINTER(C.c_float)) for o in outs])
    for got, want, name in zip(outs, (want_cross, want_diag, want_h, want_v), ("cross", "diag", "horiz", "vert")):
        assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want, f).view(np.uint32)), name


def test_coverage_planes_any_number_of_masks():
    """The reference has no limit on masks (lib.rs:481-499).  The chain under real coverage, with the product's layout — one
    16-bit plane per 16 masks, the masks past the first word through edit_chain's WideMasks (k_apply<SINGLE = 7>) — against
    the oracle's f32 mask planes: bit-exact on both sides of every plane boundary."""
    for n_masks in (2, 15, 16, 17, 31, 32, 33, 50):
        rng = np.random.default_rng(700 + n_masks)
        n = 6_000
        rgb = _pixels(n, 800 + n_masks)
        params = [util.random_params(rng, trivial_prob=0.4) for _ in range(n_masks)]
        for p in params:
            p.vignette = 0
            p.lens_distortion = 0
        masks = np.ones((n_masks, 1, n), np.float32)
        for m in range(1, n_masks):
            masks[m, 0] = (rng.random(n) > (0.97 if m % 7 == 3 else 0.6)).astype(np.float32)
        if n_masks > 20:
            masks[20] = 0.0   # a mask that covers nothing
            masks[19] = 1.0   # and one that covers everything
        img = np.concatenate([rgb, np.ones((n, 1), np.float32)], 1).reshape(1, n, 4)
        want = O.apply(img, params, masks=masks).reshape(n, 4)
        got = util.twin_apply_masked(rgb, params, masks)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), n_masks
