"""rawphotoforge_amd — MI355X-native RAW development / edit hot path (gfx950 HIP kernels behind a C ABI).

The product is ``librpf_hip.so`` (``include/rpf.h``); this package is its ctypes mirror of the
reference's ``photo-editor`` operator surface plus the row-tile sharding helpers.  Importing the
package never touches the GPU; the first call into the library does.  There is no CPU fallback.
"""
from ._native import (CFA_BGGR, CFA_GBRG, CFA_GRBG, CFA_RGGB, CURVE_BRIGHTNESS, CURVE_HUE, CURVE_LIGHTNESS,
                      CURVE_RESOLUTION, CURVE_SATURATION, FLAVOUR_ACTIVE, FLAVOUR_GODOT_LEGACY, FUSED_MASKS, PRECISION_EXACT, PRECISION_FAST, PACK_LSB_FIRST, PACK_MIPI, PACK_MSB_FIRST, OUT_RGB_F32, OUT_RGB_U8, OUT_RGBA_F32,
                      SHARD_NO_EXCHANGE, SHARD_TIMED)
from .editor import (EmptyControlPoints, FailedToReadImage, GpuComputeError, GpuInitializationError, GpuProcessor,
                     InterpolationError, InvalidArgument, InvalidCurveLength, MaskNotFound, MismatchedLengths,
                     MissingControlPoints, MissingCurveOrControlPoints, NotEnoughPoints, NotStrictlyIncreasing,
                     PhotoEditor, PhotoEditorError, RawParams, Shard, dev_demosaic, dev_rgb16be_to_rgba_f32,
                     dev_prefetch, dev_rgb_int_to_rgba_f32, dev_unpack_bayer, pchip_interpolate, ppm16_parse_header)

__all__ = [n for n in dir() if not n.startswith("_")]
