"""Shared helpers of the test-suite: synthetic inputs, parameter sets,
ULP distance, the CPU twin loader.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWIN_SRC = os.path.join(ROOT, "tests", "twin", "rpf_twin.cpp")
TWIN_LIB = os.path.join(ROOT, "tests", "twin", "librpf_twin.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def build_twin(force=False) -> str:
    deps = [TWIN_SRC] + [os.path.join(ROOT, "rawphotoforge_amd", "csrc", f)
                         for f in ("rpf_math.h", "rpf_fast.h", "rpf_chain.h", "rpf_host.h", "rpf_tables.cpp")]
    if not force and os.path.exists(TWIN_LIB) and all(os.path.getmtime(TWIN_LIB) >= os.path.getmtime(d) for d in deps):
        return TWIN_LIB
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-shared",
                    "-I/opt/rocm/include", "-o", TWIN_LIB, TWIN_SRC,
                    os.path.join(ROOT, "rawphotoforge_amd", "csrc", "rpf_tables.cpp"), "-lm"], check=True)
    return TWIN_LIB


# Device functions checked ON THE GPU over whole binary32 ranges (tests/test_gpu_math.py): the checker's build lives here,
# in a module with no pytest import, because __graft_entry__.build() compiles it too (hipcc cross-compiles without a GPU).
CHECKER_SRC = os.path.join(ROOT, "tests", "gpu_math", "gpu_math_check.hip")
CHECKER_LIB = os.path.join(ROOT, "tests", "gpu_math", "librpf_gpu_math_check.so")
CHECKER_HDR = os.path.join(ROOT, "rawphotoforge_amd", "csrc", "rpf_math.h")
CHECKER_HDR2 = os.path.join(ROOT, "rawphotoforge_amd", "csrc", "rpf_fast.h")


def build_checker(force=False) -> str:
    if force or not os.path.exists(CHECKER_LIB) or os.path.getmtime(CHECKER_LIB) < max(os.path.getmtime(CHECKER_SRC),
                                                                                      os.path.getmtime(CHECKER_HDR), os.path.getmtime(CHECKER_HDR2)):
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-x", "hip", "--offload-arch=gfx950", "-O3",
                        "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", CHECKER_SRC, "-o", CHECKER_LIB], check=True)
    return CHECKER_LIB


# The test-only RCCL stand-in (tests/rccl_standin): several processes on one GPU as the ranks of the library's communicators.
STANDIN_SRC = os.path.join(ROOT, "tests", "rccl_standin", "rccl_standin.cpp")
STANDIN_LIB = os.path.join(ROOT, "tests", "rccl_standin", "librccl_standin.so")


def build_standin(force=False) -> str:
    if force or not os.path.exists(STANDIN_LIB) or os.path.getmtime(STANDIN_LIB) < os.path.getmtime(STANDIN_SRC):
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "-fPIC", "-shared", STANDIN_SRC,
                        "-o", STANDIN_LIB, "-lrt"], check=True)
    return STANDIN_LIB


_twin = None


def twin():
    global _twin
    if _twin is None:
        _twin = C.CDLL(build_twin())
        for n in ("twin_pow_4", "twin_srgb_to_linear"):
            getattr(_twin, n).restype = C.c_float
            getattr(_twin, n).argtypes = [C.c_float]
        _twin.twin_vignette_gain.restype = C.c_float
        _twin.twin_vignette_gain.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float]
        for n in ("twin_pow_cbrt", "twin_pow_gamma", "twin_pow_1p5", "twin_hue_angle"):
            getattr(_twin, n).restype = C.c_float
        _twin.twin_pow_cbrt.argtypes = [C.c_float]
        _twin.twin_pow_gamma.argtypes = [C.c_float]
        _twin.twin_pow_1p5.argtypes = [C.c_float]
        _twin.twin_atan2.restype = C.c_float
        _twin.twin_atan2.argtypes = [C.c_float, C.c_float]
        _twin.twin_div_u16.restype = C.c_float
        _twin.twin_div_u16.argtypes = [C.c_uint32, C.c_float]
    return _twin


def twin_apply(rgb, masks_params, fused=False, fast=False) -> np.ndarray:
    """Device arithmetic (rpf_chain.h) + the product's host parameter packing, on the CPU.  fused=True (one mask): the
    chain as the fused develop kernels run it when the mask's hue / saturation / lightness curves are folded into the
    fused hue table (edit_chain's HUE_MODE 2).  fast=True: edit_chain_fast (RPF_PRECISION_FAST) with the host forms of
    its builtins."""
    n = len(masks_params)
    a = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
    ep = (O.EditParamsC * n)(*[p.to_c() for p in masks_params])
    luts = [np.stack([np.asarray(getattr(p, nm), np.int32) for p in masks_params])
            for nm in ("brightness_tone_curve", "hue_tone_curve", "saturation_tone_curve", "lightness_tone_curve")]
    out = np.empty((a.shape[0], 4), np.float32)
    twin().twin_apply_mode(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.shape[0]), ep, C.c_uint32(n),
                           *[l.ctypes.data_as(C.c_void_p) for l in luts], out.ctypes.data_as(C.c_void_p),
                           C.c_int((3 if fast == "builtins" else 2) if fast else (1 if fused else 0)))
    return out


def bench_line(stdout: str) -> dict:
    """THE JSON line of a bench.py run from the text of a pipe that several ranks may have written to (another rank's message
    can land on the same line in front of it): exactly one object starting with "metric", parsed from where it starts."""
    import json
    import re
    starts = [m.start() for m in re.finditer(re.escape('{"metric"'), stdout)]
    assert len(starts) == 1, stdout[-2000:]
    return json.JSONDecoder().raw_decode(stdout[starts[0]:])[0]


def mask_bit_planes(masks) -> np.ndarray:
    """Binarised masks (num_masks, ..., "main" first) -> the product's coverage planes (rpf_launch.h DeviceEdit::mask_bits):
    one u16 per pixel and 16 masks; plane 0 bit j = user mask j + 1, plane g >= 1 bit j = mask 16 g + j."""
    m = np.asarray(masks)
    n = m.shape[0]
    flat = (m.reshape(n, -1) == 1.0)
    planes = np.zeros(((n + 15) // 16, flat.shape[1]), np.uint16)
    for k in range(1, n):
        g, j = (0, k - 1) if k < 16 else (k // 16, k % 16)
        planes[g] |= (flat[k].astype(np.uint16) << np.uint16(j))
    return planes


def twin_apply_masked(rgb, masks_params, masks) -> np.ndarray:
    """twin_apply under real coverage (any number of masks): the exact chain with the product's coverage-plane layout."""
    n = len(masks_params)
    a = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
    ep = (O.EditParamsC * n)(*[p.to_c() for p in masks_params])
    luts = [np.stack([np.asarray(getattr(p, nm), np.int32) for p in masks_params])
            for nm in ("brightness_tone_curve", "hue_tone_curve", "saturation_tone_curve", "lightness_tone_curve")]
    planes = np.ascontiguousarray(mask_bit_planes(masks))
    assert planes.shape[1] == a.shape[0]
    out = np.empty((a.shape[0], 4), np.float32)
    twin().twin_apply_masked(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.shape[0]), ep, C.c_uint32(n),
                             *[l.ctypes.data_as(C.c_void_p) for l in luts], planes.ctypes.data_as(C.c_void_p),
                             out.ctypes.data_as(C.c_void_p))
    return out


def ulp_diff(a, b) -> np.ndarray:
    """Distance in units in the last place between two float32 arrays (monotone integer mapping)."""
    def key(x):
        i = np.ascontiguousarray(x, np.float32).view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)
    return np.abs(key(a) - key(b))


# ---- synthetic inputs and parameter sets, SURVEY.md §8(d) --------------------------------------
from rawphotoforge_amd import synthetic as syn  # noqa: E402  (pure numpy; shared with bench.py)

BLACK, WHITE = syn.BLACK, syn.WHITE
bayer_uniform = syn.bayer_uniform
bayer_natural = syn.bayer_natural


def raw_params(pattern=O.CFA_RGGB) -> O.RawParams:
    r = dict(syn.RAW)
    r["cfa_pattern"] = pattern
    return O.RawParams(**r)


def oracle_params(spec: dict) -> O.EditParameters:
    """The oracle's EditParameters for a synthetic.PARAM_SETS entry (curves through the ORACLE's PCHIP)."""
    scalars = {k: v for k, v in spec.items() if k not in ("brightness", "hue", "saturation", "lightness")}
    p = O.EditParameters(**scalars)
    for name in ("brightness", "hue", "saturation", "lightness"):
        c = spec.get(name)
        if c is None:
            continue
        if c[0] == "points":
            v = O.curve_from_control_points(c[1], c[2])
        elif c[0] == "const":
            v = np.full(O.CURVE_RESOLUTION, c[1], np.int32)
        else:
            v = np.asarray(c[1], np.int32)
        setattr(p, name + "_tone_curve", v)
    return p


def params_D() -> O.EditParameters:
    """Set D: reference defaults (lib.rs:44-64)."""
    return oracle_params(syn.PARAM_SETS["D"])


def params_S() -> O.EditParameters:
    """Set S (stress), SURVEY.md §8(d)."""
    return oracle_params(syn.PARAM_SETS["S"])


def params_G() -> O.EditParameters:
    """General: every curve non-trivial (hue/sat/light from PCHIP, brightness arbitrary incl. out-of-range)."""
    return oracle_params(syn.PARAM_SETS["G"])


def configure_editor(ed, p: O.EditParameters, mask_name=None, main=True):
    """Drive the product's setters from an oracle EditParameters (like web/main.ts:809-826)."""
    ed.set_whitebalance(p.wb_temperature, p.wb_tint, mask_name)
    ed.set_tone(p.exposure, p.contrast, p.shadow, p.highlight, p.black, p.white, mask_name)
    if main:
        ed.set_vignette(p.vignette)
        ed.set_lens_distortion_correction(p.lens_distortion)
    ed.set_brightness_tone_curve(curve=p.brightness_tone_curve, mask_name=mask_name)
    ed.set_oklch_hue_curve(curve=p.hue_tone_curve, mask_name=mask_name)
    ed.set_oklch_saturation_curve(curve=p.saturation_tone_curve, mask_name=mask_name)
    ed.set_oklch_lightness_curve(curve=p.lightness_tone_curve, mask_name=mask_name)


def to_raw(rp: O.RawParams):
    import rawphotoforge_amd as R
    return R.RawParams(rp.cfa_pattern, rp.black, rp.white, rp.wb, rp.cam2rgb)


def random_params(rng, trivial_prob=0.4) -> O.EditParameters:
    """A random point of the operator's parameter space, including the flag combinations the kernels
    specialise on (identity / constant curves, zero sliders)."""
    def slider():
        return 0 if rng.random() < trivial_prob else int(rng.integers(-100, 101))
    p = O.EditParameters(exposure=0.0 if rng.random() < trivial_prob else float(np.float32(rng.uniform(-3, 3))),
                         contrast=slider(), shadow=slider(), highlight=slider(), black=slider(), white=slider(),
                         wb_temperature=slider(), wb_tint=slider())

    def curve(kind):
        u = rng.random()
        if u < trivial_prob:
            return None
        if u < 0.6:   # PCHIP through random control points
            n = int(rng.integers(2, 7))
            xs = np.sort(rng.choice(65536, n, replace=False)).astype(np.int32)
            xs[0], xs[-1] = 0, 65535
            xs = np.unique(xs)
            ys = rng.integers(0, 65536, xs.size).astype(np.int32)
            return O.curve_from_control_points(xs, ys)
        if u < 0.8:
            return np.full(65536, int(rng.integers(0, 65536)), np.int32)
        return rng.integers(-2000, 70000, 65536).astype(np.int32)   # arbitrary, incl. out-of-range entries
    for name in ("brightness", "hue", "saturation", "lightness"):
        c = curve(name)
        if c is not None:
            setattr(p, name + "_tone_curve", c)
    return p
