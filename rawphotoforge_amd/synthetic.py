"""Synthetic inputs and edit-parameter sets of SURVEY.md §8(d), shared by bench.py, the smoke test
and the test-suite.  Pure numpy; nothing here touches the oracle or the GPU."""
from __future__ import annotations

import numpy as np

BLACK, WHITE = 512, 16383
CURVE_RESOLUTION = 65536


def bayer_uniform(h: int, w: int, seed: int = 20260630) -> np.ndarray:
    """(U): i.i.d. uniform integers(black, white + 1) — worst case for curve-gather locality."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(BLACK, WHITE + 1, size=(h, w), dtype=np.uint16)


def bayer_natural(h: int, w: int, seed: int = 20260631) -> np.ndarray:
    """(N): smooth gradient + 3 low-frequency sinusoids per colour plane + 2 % noise, mosaicked RGGB."""
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)  # float64: integer output identical across CPUs
    planes = []
    for _ in range(3):
        p = 0.15 + 0.5 * (0.6 * xx / max(w - 1, 1) + 0.4 * yy / max(h - 1, 1))
        for _ in range(3):
            fx, fy, ph = rng.uniform(0.5, 3.0), rng.uniform(0.5, 3.0), rng.uniform(0, 2 * np.pi)
            p += 0.08 * np.sin(2 * np.pi * (fx * xx / w + fy * yy / h) + ph)
        planes.append(p)
    cy, cx = np.mgrid[0:h, 0:w]
    cell = (cx & 1) + 2 * (cy & 1)
    mos = np.where(cell == 0, planes[0], np.where(cell == 3, planes[2], planes[1]))
    mos = mos + rng.normal(0.0, 0.02, size=(h, w))
    return np.clip(BLACK + mos * (WHITE - BLACK), 0, 65535).astype(np.uint16)


# RAW front-half parameters: WB gains (R, G, G, B) and camera -> sRGB rows summing to 1
RAW = dict(cfa_pattern=0, black=(512.0,) * 4, white=16383.0, wb=(2.0, 1.0, 1.0, 1.5),
           cam2rgb=(1.60, -0.40, -0.20, -0.30, 1.50, -0.20, 0.00, -0.50, 1.50))


def _general_brightness(seed=7):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(-100, 70000, CURVE_RESOLUTION).astype(np.int32)


# A curve is ("points", xs, ys) -> PCHIP, ("const", v) or ("array", int32[65536]); absent = reference default.
PARAM_SETS = {
    "D": dict(),                                      # reference defaults, lib.rs:44-64
    "S": dict(exposure=0.5, contrast=20, shadow=30, highlight=-20, black=5, white=-5, wb_temperature=15,
              wb_tint=-10, vignette=-30,
              brightness=("points", [0, 16384, 49152, 65535], [0, 20000, 52000, 65535]),
              saturation=("const", 40000)),
    "G": dict(exposure=-0.3, contrast=-15, shadow=-10, highlight=25, wb_temperature=-40, wb_tint=35,
              brightness=("array", _general_brightness()),
              hue=("points", [0, 20000, 40000, 65535], [3000, 15000, 48000, 60000]),
              saturation=("points", [0, 30000, 65535], [20000, 45000, 30000]),
              lightness=("points", [0, 30000, 65535], [30000, 36000, 33000])),
    # single-stage sets for cost attribution (bench.py --params V / T): the defaults plus one stage
    "V": dict(vignette=-30),
    "T": dict(exposure=0.5, contrast=20, shadow=30, highlight=-20, black=5, white=-5, wb_temperature=15, wb_tint=-10),
    "B": dict(brightness=("points", [0, 16384, 49152, 65535], [0, 20000, 52000, 65535])),   # S's brightness curve alone
    "C": dict(saturation=("const", 40000)),                                                  # S's saturation curve alone
}


def configure(ed, spec: dict, mask_name=None, main: bool = True) -> None:
    """Drive a PhotoEditor's setters from a parameter set (as the UIs do, web/main.ts:809-826)."""
    g = spec.get
    ed.set_whitebalance(g("wb_temperature", 0), g("wb_tint", 0), mask_name)
    ed.set_tone(g("exposure", 0.0), g("contrast", 0), g("shadow", 0), g("highlight", 0), g("black", 0),
                g("white", 0), mask_name)
    if main:
        ed.set_vignette(g("vignette", 0))
        ed.set_lens_distortion_correction(g("lens_distortion", 0))
    setters = {"brightness": ed.set_brightness_tone_curve, "hue": ed.set_oklch_hue_curve,
               "saturation": ed.set_oklch_saturation_curve, "lightness": ed.set_oklch_lightness_curve}
    defaults = {"brightness": np.arange(CURVE_RESOLUTION, dtype=np.int32),
                "hue": np.arange(CURVE_RESOLUTION, dtype=np.int32),
                "saturation": np.full(CURVE_RESOLUTION, 32767, np.int32),
                "lightness": np.full(CURVE_RESOLUTION, 32767, np.int32)}
    for name, setter in setters.items():
        c = spec.get(name)
        if c is None:
            setter(curve=defaults[name], mask_name=mask_name)
        elif c[0] == "points":
            setter(control_points_x=c[1], control_points_y=c[2], mask_name=mask_name)
        elif c[0] == "const":
            setter(curve=np.full(CURVE_RESOLUTION, c[1], np.int32), mask_name=mask_name)
        else:
            setter(curve=c[1], mask_name=mask_name)


def pack_msb(cfa: np.ndarray, bits: int) -> np.ndarray:
    """MSB-first bit-packing of a (rows, width) u16 plane, rows padded to whole bytes (synthetic packed-RAW input for
    bench.py; the layout rpf_dev_unpack_bayer / rpf_dev_develop_packed read)."""
    rows, w = cfa.shape
    b = np.unpackbits(np.ascontiguousarray(cfa).astype(">u2").view(np.uint8).reshape(rows, w, 2), axis=2)[:, :, 16 - bits:]
    return np.packbits(b.reshape(rows, w * bits), axis=1)


def pack_mipi(cfa: np.ndarray, bits: int) -> np.ndarray:
    """MIPI CSI-2 RAW10 / RAW12 / RAW14 packing of a (rows, width) u16 plane (width a multiple of the group size):
    groups of 4 / 2 / 4 pixels in 5 / 3 / 7 bytes, high bytes first, then the low bits LSB-first in pixel order
    (RPF_PACK_MIPI of include/rpf.h); rows padded to a multiple of 8 bytes."""
    rows, w = cfa.shape
    n, low = (2 if bits == 12 else 4), bits - 8
    if w % n:
        raise ValueError("width must be a multiple of the CSI-2 group size")
    v = np.ascontiguousarray(cfa).astype(np.uint32).reshape(rows, w // n, n)
    hi = (v >> low).astype(np.uint8)
    lowbits = np.unpackbits((v & ((1 << low) - 1)).astype(np.uint8)[..., None], axis=3, bitorder="little")[..., :low]
    lo = np.packbits(lowbits.reshape(rows, w // n, n * low), axis=2, bitorder="little")
    out = np.concatenate([hi, lo], axis=2).reshape(rows, -1)
    pad = (-out.shape[1]) % 8
    return np.pad(out, ((0, 0), (0, pad))) if pad else out
