"""Host-side mirror of the reference's operator surface over the C ABI.

Same names, argument meaning and error behaviour as ``rust/photo-editor/src/lib.rs``
(``PhotoEditor``, lib.rs:86-516) and ``gpu_image_processing.rs`` (``GpuProcessor``, :42-432),
so tests read like tests of the reference.  All arithmetic happens in librpf_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _native as N


# ---- errors.rs ---------------------------------------------------------------------------------
class PhotoEditorError(Exception):
    """errors.rs:7-49"""
    code = -1


class FailedToReadImage(PhotoEditorError):
    code = N.ERR_FAILED_TO_READ_IMAGE


class GpuComputeError(PhotoEditorError):
    code = N.ERR_GPU_COMPUTE


class GpuInitializationError(PhotoEditorError):
    code = N.ERR_GPU_INITIALIZATION


class MaskNotFound(PhotoEditorError):
    code = N.ERR_MASK_NOT_FOUND


class InterpolationError(PhotoEditorError):
    """errors.rs:105-130 (wrapped as PhotoEditorError::Interpolation)"""


class MismatchedLengths(InterpolationError):
    code = N.ERR_INTERP_MISMATCHED_LENGTHS


class NotEnoughPoints(InterpolationError):
    code = N.ERR_INTERP_NOT_ENOUGH_POINTS


class NotStrictlyIncreasing(InterpolationError):
    code = N.ERR_INTERP_NOT_STRICTLY_INCREASING

    def __init__(self, msg, index=None):
        super().__init__(msg)
        self.index = index


class MissingCurveOrControlPoints(InterpolationError):
    code = N.ERR_INTERP_MISSING_CURVE_OR_CONTROL_POINTS


class InvalidCurveLength(InterpolationError):
    code = N.ERR_INTERP_INVALID_CURVE_LENGTH


class MissingControlPoints(InterpolationError):
    code = N.ERR_INTERP_MISSING_CONTROL_POINTS


class EmptyControlPoints(InterpolationError):
    code = N.ERR_INTERP_EMPTY_CONTROL_POINTS


class InvalidArgument(PhotoEditorError):
    code = N.ERR_INVALID_ARGUMENT


_BY_CODE = {c.code: c for c in (FailedToReadImage, GpuComputeError, GpuInitializationError, MaskNotFound,
                                MismatchedLengths, NotEnoughPoints, NotStrictlyIncreasing,
                                MissingCurveOrControlPoints, InvalidCurveLength, MissingControlPoints,
                                EmptyControlPoints, InvalidArgument)}


def _check(rc: int) -> None:
    if rc != N.OK:
        raise _BY_CODE.get(rc, PhotoEditorError)(N.last_error())


def _i32(a) -> Optional[np.ndarray]:
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _name(mask_name: Optional[str]):
    return None if mask_name is None else mask_name.encode("utf-8")


def pchip_interpolate(x_pts, y_pts, x_eval) -> np.ndarray:
    """interpolation::pchip_interpolate (interpolation.rs:11-123) on i32 arrays."""
    x, y, e = _i32(x_pts), _i32(y_pts), _i32(x_eval)
    out = np.zeros(e.size, np.int32)
    bad = C.c_int32(-1)
    rc = N.lib().rpf_pchip_interpolate_i32(_ptr(x, C.c_int32), x.size, _ptr(y, C.c_int32), y.size,
                                           _ptr(e, C.c_int32), e.size, _ptr(out, C.c_int32), C.byref(bad))
    if rc == N.ERR_INTERP_NOT_STRICTLY_INCREASING:
        raise NotStrictlyIncreasing(N.last_error(), bad.value)
    _check(rc)
    return out


class RawParams:
    """This build's RAW front-half parameters (rpf_raw_params): black/wb per RGGB cell position."""

    def __init__(self, cfa_pattern=N.CFA_RGGB, black=(512.0,) * 4, white=16383.0, wb=(2.0, 1.0, 1.0, 1.5),
                 cam2rgb=(1.60, -0.40, -0.20, -0.30, 1.50, -0.20, 0.00, -0.50, 1.50)):
        self.cfa_pattern, self.black, self.white, self.wb, self.cam2rgb = cfa_pattern, black, white, wb, cam2rgb

    def to_c(self) -> N.RawParams:
        r = N.RawParams()
        r.cfa_pattern = int(self.cfa_pattern)
        r.black[:] = [float(v) for v in self.black]
        r.white = float(self.white)
        r.wb[:] = [float(v) for v in self.wb]
        r.cam2rgb[:] = [float(v) for v in self.cam2rgb]
        return r


class GpuProcessor:
    """GpuProcessor (gpu_image_processing.rs:34-432): one HIP device."""

    def __init__(self, adapter_index: int = 0):
        h = C.c_void_p()
        _check(N.lib().rpf_processor_create(int(adapter_index), C.byref(h)))
        self._h = h

    @staticmethod
    def get_adapter_string_list() -> list:
        """gpu_image_processing.rs:412-426"""
        n = C.c_int(0)
        if N.lib().rpf_device_count(C.byref(n)) != N.OK:
            return []
        out = []
        for i in range(n.value):
            buf = C.create_string_buffer(256)
            if N.lib().rpf_device_name(i, buf, 256) == N.OK:
                out.append(buf.value.decode())
        return out

    @property
    def compute_units(self) -> int:
        n = C.c_int(0)
        _check(N.lib().rpf_processor_compute_units(self._h, C.byref(n)))
        return n.value

    @property
    def device(self) -> int:
        n = C.c_int(0)
        _check(N.lib().rpf_processor_device(self._h, C.byref(n)))
        return n.value

    def set_kernel_variant(self, variant: int) -> None:
        _check(N.lib().rpf_set_kernel_variant(self._h, int(variant)))

    def close(self) -> None:
        if getattr(self, "_h", None):
            N.lib().rpf_processor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PhotoEditor:
    """PhotoEditor (lib.rs:77-516).  Construct with new_from_rgb_f32 / new_from_bayer_u16 /
    headless / for_tile."""

    def __init__(self, handle, processor: Optional[GpuProcessor]):
        self._h = handle
        self._proc = processor  # keep the processor alive (Arc<GpuProcessor>, lib.rs:83)

    # -- constructors ---------------------------------------------------------------------------
    @classmethod
    def new_from_rgb_f32(cls, gpu_processor: GpuProcessor, image_vec, height: int, width: int) -> "PhotoEditor":
        """lib.rs:125-166 — note (height, width) order as in the Rust signature."""
        a = np.ascontiguousarray(image_vec, dtype=np.float32).reshape(-1)
        if a.size != height * width * 3:
            raise FailedToReadImage("Failed to read image: expected height*width*3 values")
        h = C.c_void_p()
        _check(N.lib().rpf_editor_create_from_rgb_f32(gpu_processor._h, _ptr(a, C.c_float), width, height,
                                                      C.byref(h)))
        return cls(h, gpu_processor)

    @classmethod
    def new_from_rgb_int(cls, gpu_processor: GpuProcessor, image, lut=None) -> "PhotoEditor":
        """A decoded (H, W, 3) uint8 / uint16 image (read_image behind the decoder, image.rs:412-447): v / 255 resp.
        v / 65535 on the device, or lut[v] (256 / 65536 floats: the caller's colour-space step, image.rs:431-441)."""
        a = np.ascontiguousarray(image)
        if a.dtype not in (np.uint8, np.uint16) or a.ndim != 3 or a.shape[2] != 3:
            raise FailedToReadImage("Failed to read image: expected (H, W, 3) uint8 or uint16")
        wide = a.dtype == np.uint16
        t = None
        if lut is not None:
            t = np.ascontiguousarray(lut, dtype=np.float32).reshape(-1)
            if t.size != (65536 if wide else 256):
                raise InvalidArgument("the table must hold one float per sample value")
        h = C.c_void_p()
        fn = N.lib().rpf_editor_create_from_rgb_u16 if wide else N.lib().rpf_editor_create_from_rgb_u8
        _check(fn(gpu_processor._h, _ptr(a, C.c_uint16 if wide else C.c_uint8), a.shape[1], a.shape[0],
                  _ptr(t, C.c_float) if t is not None else None, C.byref(h)))
        return cls(h, gpu_processor)

    @classmethod
    def new_from_ppm16(cls, gpu_processor: GpuProcessor, file_bytes) -> "PhotoEditor":
        """A 16-bit P6 PPM file (web-ts/core/image.ts:146-195 loadPpm); samples become f32 on the device."""
        a = np.frombuffer(bytes(file_bytes), dtype=np.uint8)
        h = C.c_void_p()
        _check(N.lib().rpf_editor_create_from_ppm16(gpu_processor._h, _ptr(a, C.c_uint8), a.size, C.byref(h)))
        return cls(h, gpu_processor)

    @classmethod
    def new_from_bayer_u16(cls, gpu_processor: GpuProcessor, cfa, raw: RawParams) -> "PhotoEditor":
        """RAW ingest (the step the Godot-legacy crate hands to rawler, image.rs:509-557)."""
        a = np.ascontiguousarray(cfa, dtype=np.uint16)
        if a.ndim != 2:
            raise FailedToReadImage("Failed to read image: CFA must be (height, width)")
        rp = raw.to_c()
        h = C.c_void_p()
        _check(N.lib().rpf_editor_create_from_bayer_u16(gpu_processor._h, _ptr(a, C.c_uint16), a.shape[1],
                                                        a.shape[0], C.byref(rp), C.byref(h)))
        return cls(h, gpu_processor)

    @classmethod
    def headless(cls, width: int, height: int) -> "PhotoEditor":
        """Operator state only, no device (rpf_editor_create_headless)."""
        h = C.c_void_p()
        _check(N.lib().rpf_editor_create_headless(width, height, C.byref(h)))
        return cls(h, None)

    @classmethod
    def for_tile(cls, gpu_processor: GpuProcessor, frame_width: int, frame_height: int, y0: int,
                 rows: int) -> "PhotoEditor":
        """Edit state + device curves/masks for a row tile; pixels stay with the caller (rpf_dev_*)."""
        t = N.Tile(frame_width, frame_height, y0, rows)
        h = C.c_void_p()
        _check(N.lib().rpf_editor_create_for_tile(gpu_processor._h, C.byref(t), C.byref(h)))
        return cls(h, gpu_processor)

    def create_preview(self, target_long_edge: int) -> "PhotoEditor":
        """A new editor over this editor's original image resized to `target_long_edge` (web/main.ts:652-695,
        resizeFloat32RGBLongEdge :958-1026), computed on the device."""
        h = C.c_void_p()
        _check(N.lib().rpf_editor_create_preview(self._h, int(target_long_edge), C.byref(h)))
        return PhotoEditor(h, self._proc)

    # -- geometry -------------------------------------------------------------------------------
    @property
    def width(self) -> int:
        v = C.c_uint32(0)
        _check(N.lib().rpf_width(self._h, C.byref(v)))
        return v.value

    @property
    def height(self) -> int:
        v = C.c_uint32(0)
        _check(N.lib().rpf_height(self._h, C.byref(v)))
        return v.value

    # -- setters (lib.rs:255-479) ---------------------------------------------------------------
    def set_whitebalance(self, temperature: int, tint: int, mask_name: Optional[str] = None) -> None:
        _check(N.lib().rpf_set_whitebalance(self._h, temperature, tint, _name(mask_name)))

    def set_vignette(self, value: int) -> None:
        _check(N.lib().rpf_set_vignette(self._h, value))

    def set_lens_distortion_correction(self, value: int) -> None:
        _check(N.lib().rpf_set_lens_distortion_correction(self._h, value))

    def set_tone(self, exposure: float, contrast: int, shadow: int, highlight: int, black: int, white: int,
                 mask_name: Optional[str] = None) -> None:
        _check(N.lib().rpf_set_tone(self._h, exposure, contrast, shadow, highlight, black, white, _name(mask_name)))

    def _set_curve(self, fn, curve, xs, ys, mask_name):
        c, x, y = _i32(curve), _i32(xs), _i32(ys)
        _check(fn(self._h, _ptr(c, C.c_int32), 0 if c is None else c.size, _ptr(x, C.c_int32),
                  0 if x is None else x.size, _ptr(y, C.c_int32), 0 if y is None else y.size, _name(mask_name)))

    def set_brightness_tone_curve(self, curve=None, control_points_x=None, control_points_y=None,
                                  mask_name: Optional[str] = None) -> None:
        self._set_curve(N.lib().rpf_set_brightness_tone_curve, curve, control_points_x, control_points_y, mask_name)

    def set_oklch_hue_curve(self, curve=None, control_points_x=None, control_points_y=None,
                            mask_name: Optional[str] = None) -> None:
        self._set_curve(N.lib().rpf_set_oklch_hue_curve, curve, control_points_x, control_points_y, mask_name)

    def set_oklch_saturation_curve(self, curve=None, control_points_x=None, control_points_y=None,
                                   mask_name: Optional[str] = None) -> None:
        self._set_curve(N.lib().rpf_set_oklch_saturation_curve, curve, control_points_x, control_points_y, mask_name)

    def set_oklch_lightness_curve(self, curve=None, control_points_x=None, control_points_y=None,
                                  mask_name: Optional[str] = None) -> None:
        self._set_curve(N.lib().rpf_set_oklch_lightness_curve, curve, control_points_x, control_points_y, mask_name)

    # -- masks (lib.rs:481-505) -----------------------------------------------------------------
    def add_mask(self, name: str, mask_data) -> None:
        m = np.ascontiguousarray(mask_data, dtype=np.float32)
        if m.size != self.width * self.height:
            raise InvalidArgument("mask must be height x width")
        _check(N.lib().rpf_add_mask(self._h, name.encode(), _ptr(m, C.c_float)))

    def remove_mask(self, name: str) -> None:
        _check(N.lib().rpf_remove_mask(self._h, name.encode()))

    def set_mask_range(self, mask_range: float) -> None:
        _check(N.lib().rpf_set_mask_range(self._h, mask_range))

    def reset(self) -> None:
        _check(N.lib().rpf_reset(self._h))

    def set_shader_flavour(self, flavour: int) -> None:
        """FLAVOUR_ACTIVE (rust/photo-editor shader) or FLAVOUR_GODOT_LEGACY (rust-godot-legacy shader:
        sRGB-encoded input linearised in the kernel, the older vignette)."""
        _check(N.lib().rpf_set_shader_flavour(self._h, flavour))

    def set_precision(self, precision: int) -> None:
        """PRECISION_EXACT (default: every builtin correctly rounded, bit-identical to the parity oracle) or PRECISION_FAST
        (binary32-only builtins within 1 ulp, contracted matrix products; include/rpf.h rpf_set_precision)."""
        _check(N.lib().rpf_set_precision(self._h, precision))

    @property
    def precision(self) -> int:
        v = C.c_int(0)
        _check(N.lib().rpf_get_precision(self._h, C.byref(v)))
        return v.value

    @property
    def num_masks(self) -> int:
        v = C.c_uint32(0)
        _check(N.lib().rpf_num_masks(self._h, C.byref(v)))
        return v.value

    # -- apply + readback (lib.rs:507-515, image.rs:202-383) ------------------------------------
    def apply_adjustments(self) -> None:
        _check(N.lib().rpf_apply_adjustments(self._h))

    def sync(self) -> None:
        _check(N.lib().rpf_editor_sync(self._h))

    # `out`: an array of the right size and dtype to read into (a caller that reads frame after frame should reuse
    # one: first-touch page faults of a fresh 384-MB buffer cost more than the transfer)
    def _out(self, out, n, dtype):
        if out is None:
            return np.empty(n, dtype)
        if out.dtype != dtype or out.size != n or not out.flags.c_contiguous:
            raise InvalidArgument(f"out must be a C-contiguous {np.dtype(dtype).name} array of {n} elements")
        return out

    def to_flat_vec_rgba(self, out=None) -> np.ndarray:
        out = self._out(out, self.width * self.height * 4, np.float32)
        _check(N.lib().rpf_get_rgba_f32(self._h, _ptr(out, C.c_float)))
        return out

    def to_flat_vec_rgb(self, out=None) -> np.ndarray:
        out = self._out(out, self.width * self.height * 3, np.float32)
        _check(N.lib().rpf_get_rgb_f32(self._h, _ptr(out, C.c_float)))
        return out

    def to_rgb_array3_u8(self, out=None) -> np.ndarray:
        out = self._out(out, self.width * self.height * 3, np.uint8)
        _check(N.lib().rpf_get_rgb_u8(self._h, _ptr(out, C.c_uint8)))
        return out.reshape(self.height, self.width, 3)

    def rgba(self) -> np.ndarray:
        return self.to_flat_vec_rgba().reshape(self.height, self.width, 4)

    # -- introspection --------------------------------------------------------------------------
    def gpu_params(self, mask_name: Optional[str] = None) -> N.GpuEditParameters:
        g = N.GpuEditParameters()
        _check(N.lib().rpf_editor_get_gpu_params(self._h, _name(mask_name), C.byref(g)))
        return g

    def curve(self, which: int, mask_name: Optional[str] = None) -> np.ndarray:
        out = np.empty(N.CURVE_RESOLUTION, np.int32)
        _check(N.lib().rpf_editor_get_curve(self._h, which, _name(mask_name), _ptr(out, C.c_int32)))
        return out

    # -- device-pointer layer (rpf_dev_*) -------------------------------------------------------
    def dev_develop(self, d_cfa_ptr: int, raw: RawParams, d_out_ptr: int, out_format: int = N.OUT_RGBA_F32,
                    stream: int = 0) -> None:
        rp = raw if isinstance(raw, N.RawParams) else raw.to_c()  # a prebuilt rpf_raw_params skips the conversion
        _check(N.lib().rpf_dev_develop(self._h, C.c_void_p(d_cfa_ptr), C.byref(rp), C.c_void_p(d_out_ptr),
                                       out_format, C.c_void_p(stream)))

    def dev_develop_packed(self, d_packed_ptr: int, row_stride: int, bits: int, raw: RawParams, d_out_ptr: int,
                           out_format: int = N.OUT_RGBA_F32, stream: int = 0, order: int = N.PACK_MSB_FIRST) -> None:
        """Fused unpack + develop: the band's (rows + 4) rows bit-packed (10/12/14/16-bit, MSB- or LSB-first)."""
        rp = raw if isinstance(raw, N.RawParams) else raw.to_c()
        _check(N.lib().rpf_dev_develop_packed(self._h, C.c_void_p(d_packed_ptr), row_stride, bits, order, C.byref(rp),
                                              C.c_void_p(d_out_ptr), out_format, C.c_void_p(stream)))

    def dev_apply(self, d_in_ptr: int, d_out_ptr: int, out_format: int = N.OUT_RGBA_F32, stream: int = 0) -> None:
        _check(N.lib().rpf_dev_apply(self._h, C.c_void_p(d_in_ptr), C.c_void_p(d_out_ptr), out_format,
                                     C.c_void_p(stream)))

    def close(self) -> None:
        if getattr(self, "_h", None):
            N.lib().rpf_editor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Shard:
    """One rank's end of the row-tile sharding of a frame (include/rpf.h rpf_shard_*): an RCCL communicator over the
    ranks of the node and (for develop_async) a thread of its own inside librpf_hip.so.  `unique_id()` on one rank, the 128 bytes to
    the others by any channel, then `Shard(gpu, id, rank, world)` on every rank (collective)."""

    def __init__(self, gpu_processor: GpuProcessor, unique_id: Optional[bytes], rank: int = 0, world: int = 1):
        h = C.c_void_p()
        idbuf = None
        if unique_id is not None:
            if len(unique_id) != N.SHARD_ID_BYTES:
                raise InvalidArgument("the shard id is %d bytes" % N.SHARD_ID_BYTES)
            idbuf = (C.c_uint8 * N.SHARD_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _check(N.lib().rpf_shard_create(gpu_processor._h, idbuf, rank, world, C.byref(h)))
        self._h, self._proc = h, gpu_processor

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * N.SHARD_ID_BYTES)()
        _check(N.lib().rpf_shard_unique_id(buf))
        return bytes(buf)

    def set_neighbours(self, up: int, down: int) -> None:
        _check(N.lib().rpf_shard_set_neighbours(self._h, up, down))

    def info(self) -> dict:
        r, w, u, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        bus = C.create_string_buffer(64)
        _check(N.lib().rpf_shard_info(self._h, C.byref(r), C.byref(w), C.byref(u), C.byref(d), bus, 64))
        return {"rank": r.value, "world": w.value, "up": u.value, "down": d.value, "pci_bus_id": bus.value.decode()}

    def exchange(self, d_band_ptr: int, row_bytes: int, rows: int, stream: int = 0) -> None:
        _check(N.lib().rpf_shard_exchange(self._h, C.c_void_p(d_band_ptr), row_bytes, rows, C.c_void_p(stream)))

    def develop(self, editor: "PhotoEditor", d_band_ptr: int, raw, d_out_ptr: int, out_format: int = N.OUT_RGBA_F32,
                stream: int = 0, flags: int = 0, packed=None) -> None:
        """One frame step of this rank: the halo exchange, then the band's launch, enqueued by ONE library call.
        packed = (bits, row_stride, order) for bit-packed band rows."""
        rp = raw if isinstance(raw, N.RawParams) else raw.to_c()
        bits, stride, order = packed if packed else (0, 0, 0)
        _check(N.lib().rpf_shard_develop(self._h, editor._h, C.c_void_p(d_band_ptr), stride, bits, order, C.byref(rp),
                                         C.c_void_p(d_out_ptr), out_format, C.c_void_p(stream), flags))

    def develop_async(self, editor: "PhotoEditor", d_band_ptr: int, raw, d_out_ptr: int, out_format: int = N.OUT_RGBA_F32,
                      stream: int = 0, flags: int = 0, packed=None) -> None:
        """The same step handed to the shard's own thread (rpf_shard_develop_async): returns at once; flags
        SHARD_NO_EXCHANGE / SHARD_TIMED.  Two shards, two streams, frames alternating: see include/rpf.h."""
        rp = raw if isinstance(raw, N.RawParams) else raw.to_c()
        bits, stride, order = packed if packed else (0, 0, 0)
        _check(N.lib().rpf_shard_develop_async(self._h, editor._h, C.c_void_p(d_band_ptr), stride, bits, order, C.byref(rp),
                                               C.c_void_p(d_out_ptr), out_format, C.c_void_p(stream), flags))

    def flush(self) -> None:
        """Block (bounded: set_timeout) until every step handed to develop_async is enqueued on its stream; raises its first
        failure, or GpuComputeError naming rank, neighbours and the stuck step when the deadline passes."""
        _check(N.lib().rpf_shard_flush(self._h))

    def drain(self, stream: int = 0) -> None:
        """flush(), then a bounded, polling wait for `stream` to drain (rpf_shard_drain)."""
        _check(N.lib().rpf_shard_drain(self._h, C.c_void_p(stream)))

    def set_timeout(self, timeout_ms: int) -> None:
        """The deadline of every wait of this shard in ms (0 = unbounded; default RPF_WAIT_TIMEOUT_MS or 30 s)."""
        _check(N.lib().rpf_shard_set_timeout(self._h, int(timeout_ms)))

    def timing(self):
        """(number, summed milliseconds) of the SHARD_TIMED launches since the last call (waits for them)."""
        n, ms = C.c_uint32(), C.c_double()
        _check(N.lib().rpf_shard_timing(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            N.lib().rpf_shard_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dev_demosaic(processor: GpuProcessor, d_cfa_ptr: int, raw: RawParams, frame_width: int, frame_height: int,
                 y0: int, rows: int, d_rgba_ptr: int, stream: int = 0) -> None:
    rp = raw.to_c()
    t = N.Tile(frame_width, frame_height, y0, rows)
    _check(N.lib().rpf_dev_demosaic(processor._h, C.c_void_p(d_cfa_ptr), C.byref(rp), C.byref(t),
                                    C.c_void_p(d_rgba_ptr), C.c_void_p(stream)))


def dev_prefetch(processor: GpuProcessor, d_ptr: int, nbytes: int, stream: int = 0) -> None:
    """Streaming read of a freshly written device buffer: lands it in the Infinity Cache (rpf_dev_prefetch)."""
    _check(N.lib().rpf_dev_prefetch(processor._h, C.c_void_p(d_ptr), nbytes, C.c_void_p(stream)))


def ppm16_parse_header(file_bytes):
    """(width, height, data_offset) of a 16-bit P6 PPM, with loadPpm's header grammar (image.ts:146-170)."""
    a = np.frombuffer(bytes(file_bytes), dtype=np.uint8)
    w, h, off = C.c_uint32(), C.c_uint32(), C.c_size_t()
    _check(N.lib().rpf_ppm16_parse_header(_ptr(a, C.c_uint8), a.size, C.byref(w), C.byref(h), C.byref(off)))
    return w.value, h.value, off.value


def dev_rgb_int_to_rgba_f32(processor: GpuProcessor, d_samples_ptr: int, bits: int, npix: int, d_rgba_ptr: int,
                            d_lut_ptr: int = 0, stream: int = 0) -> None:
    """Decoded 8- / 16-bit RGB triplets in device memory -> RGBA32F (rpf_dev_rgb_u8 / _u16_to_rgba_f32)."""
    if bits not in (8, 16):
        raise InvalidArgument("bits must be 8 or 16")
    fn = N.lib().rpf_dev_rgb_u8_to_rgba_f32 if bits == 8 else N.lib().rpf_dev_rgb_u16_to_rgba_f32
    _check(fn(processor._h, C.c_void_p(d_samples_ptr), npix, C.c_void_p(d_lut_ptr or None), C.c_void_p(d_rgba_ptr),
              C.c_void_p(stream)))


def dev_rgb16be_to_rgba_f32(processor: GpuProcessor, d_samples_ptr: int, npix: int, d_rgba_ptr: int, stream: int = 0) -> None:
    """Big-endian u16 RGB triplets -> RGBA32F (v / 65535, alpha 1) on the device."""
    _check(N.lib().rpf_dev_rgb16be_to_rgba_f32(processor._h, C.c_void_p(d_samples_ptr), npix, C.c_void_p(d_rgba_ptr),
                                               C.c_void_p(stream)))


def dev_unpack_bayer(processor: GpuProcessor, d_packed_ptr: int, row_stride: int, width: int, rows: int, bits: int,
                     d_cfa_ptr: int, stream: int = 0, order: int = N.PACK_MSB_FIRST) -> None:
    """Packed RAW (10/12/14/16-bit MSB- or LSB-first, or MIPI CSI-2 RAW10/12/14) -> u16 CFA plane on the device
    (rpf_dev_unpack_bayer)."""
    _check(N.lib().rpf_dev_unpack_bayer(processor._h, C.c_void_p(d_packed_ptr), row_stride, width, rows, bits, order,
                                        C.c_void_p(d_cfa_ptr), C.c_void_p(stream)))
