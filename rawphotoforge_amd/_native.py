"""ctypes binding of librpf_hip.so (include/rpf.h).  Fails loudly when the HIP library is
missing: there is no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RPF_HIP_LIB: load another build of the same library (A/B measurements of compiler options)
LIB_PATH = os.environ.get("RPF_HIP_LIB") or os.path.join(_HERE, "librpf_hip.so")

CURVE_RESOLUTION = 65536
FUSED_MASKS = 16  # RPF_FUSED_MASKS: "main" + 15 user masks run the fused kernels; an editor may hold more

# rpf_status
OK = 0
ERR_FAILED_TO_READ_IMAGE = 1
ERR_FAILED_TO_SAVE_IMAGE = 2
ERR_READ_IMAGE_UNSUPPORTED_FORMAT = 3
ERR_SAVE_IMAGE_UNSUPPORTED_FORMAT = 4
ERR_GPU_COMPUTE = 5
ERR_GPU_INITIALIZATION = 6
ERR_MASK_NOT_FOUND = 7
ERR_INTERP_MISMATCHED_LENGTHS = 16
ERR_INTERP_NOT_ENOUGH_POINTS = 17
ERR_INTERP_NOT_STRICTLY_INCREASING = 18
ERR_INTERP_MISSING_CURVE_OR_CONTROL_POINTS = 19
ERR_INTERP_INVALID_CURVE_LENGTH = 20
ERR_INTERP_MISSING_CONTROL_POINTS = 21
ERR_INTERP_EMPTY_CONTROL_POINTS = 22
ERR_INVALID_ARGUMENT = 32

OUT_RGBA_F32, OUT_RGB_F32, OUT_RGB_U8 = 0, 1, 2
CFA_RGGB, CFA_BGGR, CFA_GRBG, CFA_GBRG = 0, 1, 2, 3
FLAVOUR_ACTIVE, FLAVOUR_GODOT_LEGACY = 0, 1
PRECISION_EXACT, PRECISION_FAST = 0, 1
PACK_MSB_FIRST, PACK_LSB_FIRST, PACK_MIPI = 0, 1, 2
CURVE_BRIGHTNESS, CURVE_HUE, CURVE_SATURATION, CURVE_LIGHTNESS = 0, 1, 2, 3
SHARD_ID_BYTES = 128
SHARD_NO_EXCHANGE = 1
SHARD_TIMED = 0x100


class RawParams(C.Structure):
    """rpf_raw_params"""
    _fields_ = [("cfa_pattern", C.c_uint32), ("black", C.c_float * 4), ("white", C.c_float),
                ("wb", C.c_float * 4), ("cam2rgb", C.c_float * 9)]


class GpuEditParameters(C.Structure):
    """rpf_gpu_edit_parameters == GpuEditParameters (gpu_image_processing.rs:18-32)"""
    _fields_ = [(n, C.c_float) for n in ("r_gain", "g_gain", "b_gain", "vignette", "lens_distortion", "exposure",
                                         "contrast", "shadow", "highlight", "black", "white")]


class Tile(C.Structure):
    """rpf_tile"""
    _fields_ = [("frame_width", C.c_uint32), ("frame_height", C.c_uint32), ("y0", C.c_uint32), ("rows", C.c_uint32)]


# every symbol include/rpf.h declares: name -> (restype, argtypes)
_P = C.POINTER
_vp, _sz, _i32p, _f32p, _cs = C.c_void_p, C.c_size_t, _P(C.c_int32), _P(C.c_float), C.c_char_p
_curve_args = [_vp, _i32p, _sz, _i32p, _sz, _i32p, _sz, _cs]
SYMBOLS = {
    "rpf_last_error": (C.c_char_p, []),
    "rpf_version": (C.c_char_p, []),
    "rpf_device_count": (C.c_int, [_P(C.c_int)]),
    "rpf_device_name": (C.c_int, [C.c_int, C.c_char_p, _sz]),
    "rpf_processor_create": (C.c_int, [C.c_int, _P(_vp)]),
    "rpf_processor_destroy": (None, [_vp]),
    "rpf_processor_device": (C.c_int, [_vp, _P(C.c_int)]),
    "rpf_processor_compute_units": (C.c_int, [_vp, _P(C.c_int)]),
    "rpf_editor_create_from_rgb_f32": (C.c_int, [_vp, _f32p, C.c_uint32, C.c_uint32, _P(_vp)]),
    "rpf_editor_create_from_bayer_u16": (C.c_int, [_vp, _P(C.c_uint16), C.c_uint32, C.c_uint32, _P(RawParams),
                                                   _P(_vp)]),
    "rpf_editor_create_from_rgb_u8": (C.c_int, [_vp, _P(C.c_uint8), C.c_uint32, C.c_uint32, _f32p, _P(_vp)]),
    "rpf_editor_create_from_rgb_u16": (C.c_int, [_vp, _P(C.c_uint16), C.c_uint32, C.c_uint32, _f32p, _P(_vp)]),
    "rpf_ppm16_parse_header": (C.c_int, [_P(C.c_uint8), _sz, _P(C.c_uint32), _P(C.c_uint32), _P(_sz)]),
    "rpf_editor_create_from_ppm16": (C.c_int, [_vp, _P(C.c_uint8), _sz, _P(_vp)]),
    "rpf_editor_create_headless": (C.c_int, [C.c_uint32, C.c_uint32, _P(_vp)]),
    "rpf_editor_create_for_tile": (C.c_int, [_vp, _P(Tile), _P(_vp)]),
    "rpf_editor_destroy": (None, [_vp]),
    "rpf_width": (C.c_int, [_vp, _P(C.c_uint32)]),
    "rpf_height": (C.c_int, [_vp, _P(C.c_uint32)]),
    "rpf_set_whitebalance": (C.c_int, [_vp, C.c_int32, C.c_int32, _cs]),
    "rpf_set_vignette": (C.c_int, [_vp, C.c_int32]),
    "rpf_set_lens_distortion_correction": (C.c_int, [_vp, C.c_int32]),
    "rpf_set_tone": (C.c_int, [_vp, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _cs]),
    "rpf_set_brightness_tone_curve": (C.c_int, _curve_args),
    "rpf_set_oklch_hue_curve": (C.c_int, _curve_args),
    "rpf_set_oklch_saturation_curve": (C.c_int, _curve_args),
    "rpf_set_oklch_lightness_curve": (C.c_int, _curve_args),
    "rpf_add_mask": (C.c_int, [_vp, _cs, _f32p]),
    "rpf_remove_mask": (C.c_int, [_vp, _cs]),
    "rpf_set_mask_range": (C.c_int, [_vp, C.c_float]),
    "rpf_reset": (C.c_int, [_vp]),
    "rpf_num_masks": (C.c_int, [_vp, _P(C.c_uint32)]),
    "rpf_preview_size": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, _P(C.c_uint32), _P(C.c_uint32)]),
    "rpf_editor_create_preview": (C.c_int, [_vp, C.c_uint32, _P(_vp)]),
    "rpf_set_shader_flavour": (C.c_int, [_vp, C.c_int]),
    "rpf_set_precision": (C.c_int, [_vp, C.c_int]),
    "rpf_get_precision": (C.c_int, [_vp, _P(C.c_int)]),
    "rpf_apply_adjustments": (C.c_int, [_vp]),
    "rpf_editor_sync": (C.c_int, [_vp]),
    "rpf_get_rgba_f32": (C.c_int, [_vp, _f32p]),
    "rpf_get_rgb_f32": (C.c_int, [_vp, _f32p]),
    "rpf_get_rgb_u8": (C.c_int, [_vp, _P(C.c_uint8)]),
    "rpf_editor_get_gpu_params": (C.c_int, [_vp, _cs, _P(GpuEditParameters)]),
    "rpf_editor_get_curve": (C.c_int, [_vp, C.c_int, _cs, _i32p]),
    "rpf_pchip_interpolate_i32": (C.c_int, [_i32p, _sz, _i32p, _sz, _i32p, _sz, _i32p, _i32p]),
    "rpf_dev_prefetch": (C.c_int, [_vp, _vp, _sz, _vp]),
    "rpf_dev_unpack_bayer": (C.c_int, [_vp, _vp, _sz, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp]),
    "rpf_dev_rgb_u8_to_rgba_f32": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "rpf_dev_rgb_u16_to_rgba_f32": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "rpf_dev_rgb16be_to_rgba_f32": (C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "rpf_dev_demosaic": (C.c_int, [_vp, _vp, _P(RawParams), _P(Tile), _vp, _vp]),
    "rpf_dev_develop": (C.c_int, [_vp, _vp, _P(RawParams), _vp, C.c_int, _vp]),
    "rpf_dev_develop_packed": (C.c_int, [_vp, _vp, _sz, C.c_uint32, C.c_uint32, _P(RawParams), _vp, C.c_int, _vp]),
    "rpf_dev_apply": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "rpf_set_kernel_variant": (C.c_int, [_vp, C.c_int]),
    "rpf_shard_unique_id": (C.c_int, [_P(C.c_uint8)]),
    "rpf_shard_create": (C.c_int, [_vp, _P(C.c_uint8), C.c_int, C.c_int, _P(_vp)]),
    "rpf_shard_destroy": (None, [_vp]),
    "rpf_shard_set_neighbours": (C.c_int, [_vp, C.c_int, C.c_int]),
    "rpf_shard_info": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int), _P(C.c_int), _P(C.c_int), C.c_char_p, _sz]),
    "rpf_shard_exchange": (C.c_int, [_vp, _vp, _sz, C.c_uint32, _vp]),
    "rpf_shard_develop": (C.c_int, [_vp, _vp, _vp, _sz, C.c_uint32, C.c_uint32, _P(RawParams), _vp, C.c_int, _vp,
                                    C.c_uint32]),
    "rpf_shard_develop_async": (C.c_int, [_vp, _vp, _vp, _sz, C.c_uint32, C.c_uint32, _P(RawParams), _vp, C.c_int, _vp,
                                    C.c_uint32]),
    "rpf_shard_flush": (C.c_int, [_vp]),
    "rpf_shard_drain": (C.c_int, [_vp, _vp]),
    "rpf_shard_set_timeout": (C.c_int, [_vp, C.c_int]),
    "rpf_shard_timing": (C.c_int, [_vp, _P(C.c_uint32), _P(C.c_double)]),
}

_lib = None


def _share_hip_runtime_with_torch() -> None:
    """One process, one HIP runtime.  A ROCm PyTorch wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7, requested by torch under the name "libamdhip64.so"), while librpf_hip.so asks the
    loader for "libamdhip64.so.7" — loaded in that order the process ends up with two runtimes and the
    second one sees no GPU.  If torch is installed but not imported yet, map ITS copy first (by path, no
    torch import): librpf_hip.so then resolves to it by SONAME and a later `import torch` finds it loaded."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # torch's runtime is already mapped; ours resolves to it by SONAME
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> C.CDLL:
    """Load librpf_hip.so; raise if it was not built (python -m rawphotoforge_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built and there is no CPU fallback. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` or `python rawphotoforge_amd/build.py`.")
        _share_hip_runtime_with_torch()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError here = header/library drift: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error() -> str:
    return lib().rpf_last_error().decode("utf-8", "replace")
