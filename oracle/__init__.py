"""CPU oracle — TEST INFRASTRUCTURE ONLY.

ctypes front end of ``oracle/rpf_oracle.c`` (a plain-C restatement of the
reference's per-pixel chain; see that file's header for the reference
file:line citations and the parity status).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  ``rawphotoforge_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "rpf_oracle.c")
_LIB = os.path.join(_HERE, "_build", "librpf_oracle.so")

CURVE_RESOLUTION = 65536

CFA_RGGB, CFA_BGGR, CFA_GRBG, CFA_GBRG = 0, 1, 2, 3


def build(force: bool = False) -> str:
    """Compile the C restatement (strict IEEE, no contraction)."""
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= os.path.getmtime(_SRC):
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-shared",
           "-o", _LIB, _SRC, "-lquadmath", "-lm"]
    subprocess.run(cmd, check=True)
    return _LIB


class GpuParams(C.Structure):
    """gpu_image_processing.rs:18-32 (11 x f32)."""
    _fields_ = [(n, C.c_float) for n in (
        "r_gain", "g_gain", "b_gain", "vignette", "lens_distortion", "exposure",
        "contrast", "shadow", "highlight", "black", "white")]


class EditParamsC(C.Structure):
    _fields_ = [("exposure", C.c_float)] + [(n, C.c_int32) for n in (
        "contrast", "shadow", "highlight", "black", "white", "wb_temperature", "wb_tint",
        "vignette", "lens_distortion")] + [("mask_range", C.c_float)]


class RawParamsC(C.Structure):
    _fields_ = [("cfa_pattern", C.c_uint32), ("black", C.c_float * 4), ("white", C.c_float),
                ("wb", C.c_float * 4), ("cam2rgb", C.c_float * 9)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        for name in ("orc_cr_powf", "orc_cr_atan2f"):
            getattr(_lib, name).restype = C.c_float
            getattr(_lib, name).argtypes = [C.c_float, C.c_float]
        for name in ("orc_cr_sinf", "orc_cr_cosf", "orc_srgb_to_linear1", "orc_linear_to_srgb1"):
            getattr(_lib, name).restype = C.c_float
            getattr(_lib, name).argtypes = [C.c_float]
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


@dataclass
class EditParameters:
    """lib.rs:19-64 — one mask's edit parameters, reference defaults."""
    exposure: float = 0.0
    contrast: int = 0
    shadow: int = 0
    highlight: int = 0
    black: int = 0
    white: int = 0
    wb_temperature: int = 0
    wb_tint: int = 0
    vignette: int = 0
    lens_distortion: int = 0
    mask_range: float = 0.0
    brightness_tone_curve: np.ndarray = field(default_factory=lambda: np.arange(CURVE_RESOLUTION, dtype=np.int32))
    hue_tone_curve: np.ndarray = field(default_factory=lambda: np.arange(CURVE_RESOLUTION, dtype=np.int32))
    saturation_tone_curve: np.ndarray = field(default_factory=lambda: np.full(CURVE_RESOLUTION, 32767, np.int32))
    lightness_tone_curve: np.ndarray = field(default_factory=lambda: np.full(CURVE_RESOLUTION, 32767, np.int32))

    def to_c(self) -> EditParamsC:
        return EditParamsC(self.exposure, self.contrast, self.shadow, self.highlight, self.black, self.white,
                           self.wb_temperature, self.wb_tint, self.vignette, self.lens_distortion, self.mask_range)


def convert_params(p: EditParameters) -> GpuParams:
    g = GpuParams()
    c = p.to_c()
    lib().orc_convert_params(C.byref(c), C.byref(g))
    return g


@dataclass
class RawParams:
    """This build's RAW front-half parameters (not in the reference source)."""
    cfa_pattern: int = CFA_RGGB
    black: tuple = (512.0, 512.0, 512.0, 512.0)
    white: float = 16383.0
    wb: tuple = (2.0, 1.0, 1.0, 1.5)          # R, G(red row), G(blue row), B   
    cam2rgb: tuple = (1.60, -0.40, -0.20, -0.30, 1.50, -0.20, 0.00, -0.50, 1.50)

    def to_c(self) -> RawParamsC:
        r = RawParamsC()
        r.cfa_pattern = self.cfa_pattern
        r.black[:] = [float(v) for v in self.black]
        r.white = float(self.white)
        r.wb[:] = [float(v) for v in self.wb]
        r.cam2rgb[:] = [float(v) for v in self.cam2rgb]
        return r


PCHIP_ERRORS = {1: "MismatchedLengths", 2: "NotEnoughPoints", 3: "NotStrictlyIncreasing"}


class PchipError(ValueError):
    def __init__(self, kind, index=None):
        super().__init__(kind if index is None else f"{kind}(index={index})")
        self.kind, self.index = kind, index


def pchip_interpolate(x_pts, y_pts, x_eval) -> np.ndarray:
    """interpolation.rs:11-123 on i32 arrays."""
    x = np.ascontiguousarray(x_pts, np.int32)
    y = np.ascontiguousarray(y_pts, np.int32)
    e = np.ascontiguousarray(x_eval, np.int32)
    out = np.zeros(e.shape[0], np.int32)
    bad = C.c_int32(-1)
    rc = lib().orc_pchip_interpolate_i32(_p(x, C.c_int32), C.c_size_t(x.size), _p(y, C.c_int32), C.c_size_t(y.size),
                                         _p(e, C.c_int32), C.c_size_t(e.size), _p(out, C.c_int32), C.byref(bad))
    if rc:
        raise PchipError(PCHIP_ERRORS[rc], bad.value if rc == 3 else None)
    return out


def curve_from_control_points(xs, ys) -> np.ndarray:
    """lib.rs:336-339: PCHIP over 0..65535 then clamp(0, 65535)."""
    x = np.ascontiguousarray(xs, np.int32)
    y = np.ascontiguousarray(ys, np.int32)
    out = np.zeros(CURVE_RESOLUTION, np.int32)
    bad = C.c_int32(-1)
    rc = lib().orc_curve_from_control_points(_p(x, C.c_int32), C.c_size_t(x.size), _p(y, C.c_int32),
                                             C.c_size_t(y.size), _p(out, C.c_int32), C.byref(bad))
    if rc:
        raise PchipError(PCHIP_ERRORS[rc], bad.value if rc == 3 else None)
    return out


def binarize_mask(mask, mask_range=0.0) -> np.ndarray:
    m = np.ascontiguousarray(mask, np.float32)
    out = np.empty_like(m)
    lib().orc_binarize_mask(_p(m, C.c_float), _p(out, C.c_float), C.c_size_t(m.size), C.c_float(mask_range))
    return out


def _pack(masks_params):
    n = len(masks_params)
    gp = (GpuParams * max(n, 1))()
    luts = [np.empty((max(n, 1), CURVE_RESOLUTION), np.int32) for _ in range(4)]
    for i, p in enumerate(masks_params):
        gp[i] = convert_params(p)
        for j, nm in enumerate(("brightness_tone_curve", "hue_tone_curve", "saturation_tone_curve",
                                "lightness_tone_curve")):
            luts[j][i] = np.asarray(getattr(p, nm), np.int32)
    return gp, luts


def apply(image_rgba, masks_params, masks=None, frame_hw=None, y0=0, legacy=None) -> np.ndarray:
    """GpuProcessor::apply_adjustments + wgsl main (gpu_image_processing.rs:214-400).

    image_rgba : (rows, W, 4) f32 tile.  masks_params : list[EditParameters], [0] is "main".
    masks : None (every mask all-ones) or (N, rows, W) f32, already binarised.
    frame_hw : (H, W) of the full frame when image_rgba is a row tile starting at y0.
    legacy : None = the active shader; "encoded" = the Godot-legacy shader (sRGB-encoded input linearised, older
             vignette; rust-godot-legacy/photo-editor/src/wgpu_shader.wgsl); "linear" = its vignette only.
    """
    img = np.ascontiguousarray(image_rgba, np.float32)
    rows, w = img.shape[:2]
    fh, fw = frame_hw if frame_hw is not None else (rows, w)
    assert fw == w
    gp, luts = _pack(masks_params)
    out = np.empty_like(img)
    mptr = None
    if masks is not None:
        mk = np.ascontiguousarray(masks, np.float32)
        assert mk.shape == (len(masks_params), rows, w)
        mptr = _p(mk, C.c_float)
    args = [_p(img, C.c_float), C.c_uint32(fw), C.c_uint32(fh), C.c_uint32(y0), C.c_uint32(rows),
            mptr, C.c_uint32(len(masks_params)), gp, *[_p(l, C.c_int32) for l in luts], _p(out, C.c_float)]
    if legacy is None:
        lib().orc_apply(*args)
    else:
        lib().orc_apply_legacy(*args, C.c_int({"encoded": 1, "linear": 0}[legacy]))
    return out


def demosaic(cfa, raw: RawParams, y0=0, rows=None) -> np.ndarray:
    """This build's front half on rows [y0, y0+rows) of the whole-frame CFA (H, W) u16."""
    c = np.ascontiguousarray(cfa, np.uint16)
    h, w = c.shape
    rows = h - y0 if rows is None else rows
    out = np.empty((rows, w, 4), np.float32)
    rp = raw.to_c()
    lib().orc_demosaic(_p(c, C.c_uint16), C.c_uint32(w), C.c_uint32(h), C.c_uint32(y0), C.c_uint32(rows),
                       C.byref(rp), _p(out, C.c_float))
    return out


def develop(cfa, raw: RawParams, masks_params, masks=None, y0=0, rows=None, out=None) -> np.ndarray:
    """Full RAW -> display chain: demosaic then apply.  `out`: optional preallocated (rows, W, 4) f32."""
    c = np.ascontiguousarray(cfa, np.uint16)
    h, w = c.shape
    rows = h - y0 if rows is None else rows
    gp, luts = _pack(masks_params)
    if out is None:
        out = np.empty((rows, w, 4), np.float32)
    assert out.shape == (rows, w, 4) and out.dtype == np.float32 and out.flags.c_contiguous
    rp = raw.to_c()
    mptr = None
    if masks is not None:
        mk = np.ascontiguousarray(masks, np.float32)
        assert mk.shape == (len(masks_params), rows, w)
        mptr = _p(mk, C.c_float)
    lib().orc_develop(_p(c, C.c_uint16), C.c_uint32(w), C.c_uint32(h), C.c_uint32(y0), C.c_uint32(rows),
                      C.byref(rp), mptr, C.c_uint32(len(masks_params)), gp, *[_p(l, C.c_int32) for l in luts],
                      _p(out, C.c_float))
    return out


def rgba_to_rgb_u8(rgba) -> np.ndarray:
    a = np.ascontiguousarray(rgba, np.float32)
    out = np.empty(a.shape[:-1] + (3,), np.uint8)
    lib().orc_rgba_to_rgb_u8(_p(a, C.c_float), C.c_size_t(a.size // 4), _p(out, C.c_uint8))
    return out


def legacy_apply_gamma(img, gamma=(2.222, 4.5 / 255.0)) -> np.ndarray:
    """python-legacy editor.py:47-76."""
    a = np.ascontiguousarray(img, np.float32)
    out = np.empty_like(a)
    lib().orc_legacy_apply_gamma(_p(a, C.c_float), C.c_size_t(a.size), C.c_double(gamma[0]), C.c_double(gamma[1]),
                                 _p(out, C.c_float))
    return out


def legacy_oklch_to_rgb(L, Cc, H) -> np.ndarray:
    """python-legacy editor.py:121-150 on float64 arrays."""
    l = np.ascontiguousarray(L, np.float64)
    c = np.ascontiguousarray(Cc, np.float64)
    h = np.ascontiguousarray(H, np.float64)
    out = np.empty(l.shape + (3,), np.float64)
    lib().orc_legacy_oklch_to_rgb(_p(l, C.c_double), _p(c, C.c_double), _p(h, C.c_double), C.c_size_t(l.size),
                                  _p(out, C.c_double))
    return out


def legacy_develop(cfa, raw: RawParams, gamma=(2.222, 4.5)) -> np.ndarray:
    """BASELINE config 1: front half + apply_gamma((2.222, 4.5)) (raw_photo_forge.py:1727)."""
    c = np.ascontiguousarray(cfa, np.uint16)
    h, w = c.shape
    out = np.empty((h, w, 4), np.float32)
    rp = raw.to_c()
    lib().orc_legacy_develop(_p(c, C.c_uint16), C.c_uint32(w), C.c_uint32(h), C.byref(rp), C.c_double(gamma[0]),
                             C.c_double(gamma[1]), _p(out, C.c_float))
    return out


def set_threads(n: int) -> None:
    lib().orc_set_threads(C.c_int(n))


def max_threads() -> int:
    return lib().orc_max_threads()


PACK_MSB_FIRST, PACK_LSB_FIRST, PACK_MIPI = 0, 1, 2


def packed_row_stride(width: int, bits: int, align: int = 4, order: int = 0) -> int:
    if order == PACK_MIPI:   # whole groups of 4 (RAW10 / RAW14) or 2 (RAW12) pixels
        npx = 2 if bits == 12 else 4
        n = (width + npx - 1) // npx * (npx + npx * (bits - 8) // 8)
    else:
        n = (width * bits + 7) // 8
    return (n + align - 1) // align * align


def pack_bayer(cfa, bits: int, row_stride=None, order=PACK_MSB_FIRST) -> np.ndarray:
    """Bit-packing of a (rows, width) u16 plane (test helper; inverse of unpack_bayer)."""
    a = np.ascontiguousarray(cfa, np.uint16)
    rows, w = a.shape
    stride = row_stride or packed_row_stride(w, bits, order=order)
    out = np.zeros((rows, stride), np.uint8)
    fn = {PACK_LSB_FIRST: lib().orc_pack_bayer_lsb, PACK_MIPI: lib().orc_pack_bayer_mipi}.get(order, lib().orc_pack_bayer)
    fn(_p(a, C.c_uint16), C.c_uint32(w), C.c_uint32(rows), C.c_uint32(bits), C.c_size_t(stride), _p(out, C.c_uint8))
    return out


def unpack_bayer(packed, width: int, bits: int, order=PACK_MSB_FIRST) -> np.ndarray:
    """This build's packed-RAW unpack definition (orc_unpack_bayer)."""
    p = np.ascontiguousarray(packed, np.uint8)
    rows, stride = p.shape
    out = np.empty((rows, width), np.uint16)
    fn = {PACK_LSB_FIRST: lib().orc_unpack_bayer_lsb, PACK_MIPI: lib().orc_unpack_bayer_mipi}.get(order, lib().orc_unpack_bayer)
    fn(_p(p, C.c_uint8), C.c_size_t(stride), C.c_uint32(width), C.c_uint32(rows), C.c_uint32(bits), _p(out, C.c_uint16))
    return out


def legacy_tone_lut(exposure=0.0, contrast=0, shadow=0, highlight=0, black=0, white=0) -> np.ndarray:
    """python-legacy editor.py:712-778 _create_tone_lut_from_params (float32[65536])."""
    out = np.empty(CURVE_RESOLUTION, np.float32)
    lib().orc_legacy_tone_lut(C.c_double(exposure), C.c_int(contrast), C.c_int(shadow), C.c_int(highlight),
                              C.c_int(black), C.c_int(white), _p(out, C.c_float))
    return out


def preview_size(sw: int, sh: int, target: int):
    """web/main.ts:971-977 -> (dst_width, dst_height)."""
    dw, dh = C.c_uint32(), C.c_uint32()
    lib().orc_preview_size(C.c_uint32(sw), C.c_uint32(sh), C.c_uint32(target), C.byref(dw), C.byref(dh))
    return dw.value, dh.value


def resize_long_edge(img, target: int) -> np.ndarray:
    """web/main.ts:958-1026 resizeFloat32RGBLongEdge on an (H, W, 3|4) f32 image."""
    a = np.ascontiguousarray(img, np.float32)
    sh, sw, nch = a.shape
    dw, dh = preview_size(sw, sh, target)
    out = np.empty((dh, dw, nch), np.float32)
    lib().orc_resize_bilinear(_p(a, C.c_float), C.c_uint32(sw), C.c_uint32(sh), C.c_uint32(nch), _p(out, C.c_float),
                              C.c_uint32(dw), C.c_uint32(dh))
    return out


def rgb_int_to_rgba(img, lut=None) -> np.ndarray:
    """image.rs:443 to_rgb32f (+ alpha 1) of a decoded (H, W, 3) uint8 / uint16 image; `lut` (256 / 65536 floats)
    replaces the v / MAX quotient (the caller's colour-space step, image.rs:431-441)."""
    a = np.ascontiguousarray(img)
    if a.dtype not in (np.uint8, np.uint16) or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected (H, W, 3) uint8 or uint16")
    n = 256 if a.dtype == np.uint8 else 65536
    t = None
    if lut is not None:
        t = np.ascontiguousarray(lut, np.float32)
        if t.size != n:
            raise ValueError(f"lut must hold {n} floats")
    out = np.empty(a.shape[:2] + (4,), np.float32)
    fn = lib().orc_rgb_u8_to_rgba if a.dtype == np.uint8 else lib().orc_rgb_u16_to_rgba
    fn(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.shape[0] * a.shape[1]),
       t.ctypes.data_as(C.c_void_p) if t is not None else None, out.ctypes.data_as(C.c_void_p))
    return out


def ppm16_header(file_bytes):
    """web-ts/core/image.ts:146-173 -> (status, width, height, data_offset); status 0 ok, 1 not P6, 2 size line,
    3 maxval, 4 truncated."""
    a = np.frombuffer(bytes(file_bytes), np.uint8)
    w, h, off = C.c_uint32(), C.c_uint32(), C.c_size_t()
    rc = lib().orc_ppm16_header(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.byref(w), C.byref(h), C.byref(off))
    return rc, w.value, h.value, off.value


def ppm16_load(file_bytes) -> np.ndarray:
    """loadPpm: (H, W, 4) f32 image of a 16-bit P6 file."""
    rc, w, h, off = ppm16_header(file_bytes)
    if rc:
        raise ValueError(f"ppm16 header status {rc}")
    a = np.frombuffer(bytes(file_bytes), np.uint8)
    out = np.empty((h, w, 4), np.float32)
    lib().orc_rgb16be_to_rgba(C.c_void_p(a.ctypes.data + off), C.c_size_t(w * h), out.ctypes.data_as(C.c_void_p))
    return out
