"""A second, independently written restatement of the reference's pixel chain (rust/photo-editor/src/wgpu_shader.wgsl:166-337 without
the lens resample; gpu_image_processing.rs:232-254 for the parameters) in numpy binary32 array arithmetic, against the oracle's
scalar C.  It shares the oracle's READING of the WGSL — no contraction, every operation rounded to binary32 in source order, builtins
correctly rounded — because that reading is this build's definition of parity; it shares none of its code: arrays instead of a
pixel loop, numpy's binary32 operators instead of gcc's, the builtins through numpy's extended precision (x87 long double: 64-bit
significand, then ONE rounding to binary32) instead of the oracle's own correctly rounded routines.  A slip in either transcription —
an operand order, a matrix column, a clamp, which luma the black / white stage sees, which hue indexes a later mask's curves —
shows up as a differing bit."""
import numpy as np

import oracle as O
from tests import util

F = np.float32
LD = np.longdouble


def _cr(fn, *a):   # a builtin, correctly rounded for all practical purposes: evaluated in long double, rounded once
    return fn(*[np.asarray(x, F).astype(LD) for x in a]).astype(F)


def _pow(x, y):
    return _cr(np.power, x, np.full_like(np.asarray(x, F), y))


def _mat(cols, v):   # WGSL mat3x3 (column major) * vec3: column 0 * x + column 1 * y + column 2 * z, per component, left to right
    x, y, z = v
    return [F(cols[0][k]) * x + F(cols[1][k]) * y + F(cols[2][k]) * z for k in range(3)]


M1 = ((0.4122214708, 0.2119034982, 0.0883024619), (0.5363325363, 0.6806995451, 0.2817188376), (0.0514459929, 0.1073969566, 0.6299787005))
M1_INV = ((4.0767416621, -1.2684380046, -0.0041960863), (-3.3077115913, 2.6097574011, -0.7034186147), (0.2309699292, -0.3413193965, 1.7076147010))
M2 = ((0.2104542553, 1.9779984951, 0.0259040371), (0.7936177850, -2.4285922050, 0.7827717662), (-0.0040720468, 0.4505937099, -0.8086757660))
M2_INV = ((1.0, 1.0, 1.0), (0.3963377774, -0.1055613458, -0.0894841775), (0.2158037573, -0.0638541728, -1.2914855480))


def _clamp(v, lo, hi):
    return np.minimum(np.maximum(v, F(lo)), F(hi))


def _lut(curve, idx):   # lut_fetch, wgsl:184-194
    return np.clip(np.asarray(curve, np.int64)[idx], 0, 65535).astype(np.uint32)


def chain_numpy(rgba, params, masks=None):
    h, w = rgba.shape[:2]
    rgb = [np.ascontiguousarray(rgba[..., k], F) for k in range(3)]
    n = len(params)
    if masks is None:
        masks = np.ones((n, h, w), F)
    main = params[0]
    # vignette, wgsl:166-178 ("main" only, :276)
    vs = (-F(main.vignette) / F(100.0)) * F(2.0)
    if vs != 0:
        yy, xx = np.mgrid[0:h, 0:w]
        cx = (xx.astype(F) / F(w) - F(0.5)) * F(1.5)
        cy = (yy.astype(F) / F(h) - F(0.5)) * F(1.5)
        dist = _cr(np.sqrt, cx * cx + cy * cy)
        fall = _pow(_clamp((dist - F(0.25)) / F(0.75), 0, 1), 1.5)
        gain = _clamp(F(1.0) - vs * fall, 0, 4)
        rgb = [c * gain for c in rgb]
    # loop 1, wgsl:279-308
    for m, p in enumerate(params):
        on = masks[m] == F(1.0)
        t = F(p.wb_temperature) / F(100.0)                                 # gpu_image_processing.rs:236-246
        gains = (F(1.0) + F(0.5) * t, F(1.0) - F(0.25) * (F(p.wb_tint) / F(100.0)), F(1.0) - F(0.5) * t)
        contrast, shadow, highlight = F(p.contrast) / F(100.0), F(p.shadow) / F(100.0), F(p.highlight) / F(100.0)
        black, white = F(p.black) / F(100.0), F(p.white) / F(100.0)
        c = [rgb[k] * gains[k] for k in range(3)]
        e = _pow(np.asarray([2.0], F), F(p.exposure))[0]                    # tone(), wgsl:200-259
        c = [v * e for v in c]
        luma = c[0] * F(0.2126) + c[1] * F(0.7152) + c[2] * F(0.0722)
        smask, hmask = _clamp(F(1.0) - luma, 0, 1), _clamp(luma, 0, 1)
        sg = F(1.0) + shadow * smask
        c = [v * sg for v in c]
        hg = F(1.0) + highlight * hmask
        c = [v * hg for v in c]
        if black != 0:
            tt = _clamp(luma, 0, 1)
            c = [v + black * _pow(F(1.0) - tt, 2.0) for v in c]
        if white != 0:
            tt = _clamp(luma, 0, 1)
            c = [v + white * _pow(tt, 2.0) for v in c]
        if contrast != 0:
            cc = F(1.0) + contrast
            c = [(v - F(0.5)) * cc + F(0.5) for v in c]
        c = [_clamp(v, 0, 1) for v in c]
        new = [_lut(p.brightness_tone_curve, (v * F(65535)).astype(np.uint32)).astype(F) / F(65535.0) for v in c]
        rgb = [np.where(on, new[k], rgb[k]) for k in range(3)]
    # linear_srgb_to_oklch, wgsl:64-75
    lms = _mat(M1, rgb)
    lms_c = [_pow(np.maximum(v, F(0.0)), F(1.0 / 3.0)) for v in lms]
    L, a, b = _mat(M2, lms_c)
    C = _cr(np.sqrt, a * a + b * b)
    hh = _cr(np.arctan2, b, a) / F(2.0 * 3.14159265359)
    hh = np.where(hh < 0, hh + F(1.0), hh)
    # loop 2, wgsl:312-331
    for m, p in enumerate(params):
        on = masks[m] == F(1.0)
        hu = (hh * F(65535)).astype(np.uint32)
        nh = _lut(p.hue_tone_curve, hu).astype(F) / F(65535.0)
        nc = C * (_lut(p.saturation_tone_curve, hu).astype(F) / F(32767.5))
        nl = L * (_lut(p.lightness_tone_curve, hu).astype(F) / F(32767.5))
        hh, C, L = np.where(on, nh, hh), np.where(on, nc, C), np.where(on, nl, L)
    # oklch_to_linear_srgb, wgsl:77-84
    ang = hh * F(2.0) * F(3.14159265359)
    lab = [L, C * _cr(np.cos, ang), C * _cr(np.sin, ang)]
    q = _mat(M2_INV, lab)
    lin = _mat(M1_INV, [v * v * v for v in q])
    # linear_to_srgb + clamp, wgsl:95-103, 335-336
    with np.errstate(invalid="ignore"):
        out = [np.where(v <= F(0.0031308), v * F(12.92), F(1.055) * _pow(v, F(1.0 / 2.4)) - F(0.055)) for v in lin]
    out = [_clamp(v, 0, 1) for v in out]
    return np.stack(out + [np.ones((h, w), F)], -1)


def _image(h, w, seed):
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 3), dtype=F) * F(1.25) - F(0.1)
    img[0, : w // 2] = rng.random((w // 2, 1), dtype=F)   # greys: a, b ~ 0
    img[1, :8] = 0
    img[1, 8:16] = 1
    return np.concatenate([img, np.ones((h, w, 1), F)], -1)


def _compare(rgba, params, masks=None):
    want = O.apply(rgba, params, masks=masks)
    got = chain_numpy(rgba, params, masks)
    bad = np.argwhere(want.view(np.uint32) != got.view(np.uint32))
    assert bad.size == 0, (len(bad), bad[:5], want[tuple(bad[0][:2])], got[tuple(bad[0][:2])])


def test_defaults_stress_and_general_sets():
    rgba = _image(96, 160, 1)
    for p in (util.params_D(), util.params_S(), util.params_G()):
        _compare(rgba, [p])


def test_random_parameters_and_masks():
    rng = np.random.default_rng(77)
    for case in range(12):
        h, w = int(rng.integers(20, 70)), int(rng.integers(20, 120))
        rgba = _image(h, w, 100 + case)
        n = int(rng.integers(1, 5))
        params = [util.random_params(rng) for _ in range(n)]
        for p in params:
            p.lens_distortion = 0
        masks = np.ones((n, h, w), F)
        for m in range(1, n):
            masks[m] = (rng.random((h, w)) > 0.5).astype(F)
        _compare(rgba, params, masks)
