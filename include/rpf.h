/*
 * rpf.h — C ABI of the MI355X-native RAW development / edit hot path.
 *
 * Drop-in boundary for the operator surface of the reference's `photo-editor`
 * crate (all citations relative to /root/reference):
 *
 *   rpf_processor_*            GpuProcessor::new / get_adapter_string_list
 *                              rust/photo-editor/src/gpu_image_processing.rs:43-200, 412-426
 *   rpf_editor_create_from_rgb_f32
 *                              PhotoEditor::new_from_rgb_f32   rust/photo-editor/src/lib.rs:125-166
 *   rpf_editor_create_from_bayer_u16
 *                              the RAW ingest the Godot-legacy crate delegates to rawler
 *                              rust-godot-legacy/photo-editor/src/image.rs:509-557 (arithmetic NOT in the
 *                              reference source; this build's own front half)
 *   rpf_set_whitebalance       PhotoEditor::set_whitebalance               lib.rs:255-265
 *   rpf_set_vignette           PhotoEditor::set_vignette                   lib.rs:267-270
 *   rpf_set_lens_distortion_correction
 *                              PhotoEditor::set_lens_distortion_correction lib.rs:272-275
 *   rpf_set_tone               PhotoEditor::set_tone                       lib.rs:278-298
 *   rpf_set_brightness_tone_curve / rpf_set_oklch_{hue,saturation,lightness}_curve
 *                              PhotoEditor::set_*_curve                    lib.rs:300-479
 *   rpf_add_mask / rpf_remove_mask / rpf_reset
 *                              lib.rs:481-505, 227-235
 *   rpf_apply_adjustments      PhotoEditor::apply_adjustments -> GpuProcessor::apply_adjustments -> wgsl main
 *                              lib.rs:507-515, gpu_image_processing.rs:214-400, wgpu_shader.wgsl:265-337
 *   rpf_get_rgb_f32 / rpf_get_rgba_f32 / rpf_get_rgb_u8
 *                              Image::to_flat_vec_rgb / to_flat_vec_rgba / to_u8_rgbimage
 *                              rust/photo-editor/src/image.rs:202-358, 375-383
 *   rpf_pchip_interpolate_i32  interpolation::pchip_interpolate            interpolation.rs:11-123
 *
 * plus a device-pointer layer (rpf_dev_*) used by the benchmark harness and by the
 * row-tile sharding across GPUs: the caller owns device memory (e.g. a torch
 * tensor) and the library only launches kernels on it.
 *
 * Conventions
 *   - every function returns an rpf_status (0 = OK); rpf_last_error() returns a
 *     thread-local message for the last failure on the calling thread.  Nothing
 *     throws or aborts across this boundary.
 *   - host pointers are borrowed for the duration of the call only.
 *   - an editor is externally synchronised (one thread at a time, like `&mut self`);
 *     a processor may be shared by several editors.  It is reference counted like the
 *     reference's Arc<GpuProcessor> (lib.rs:83): rpf_processor_destroy releases the caller's
 *     handle, and the device tables live until the last editor created on it is destroyed.
 *     (After rpf_processor_destroy the handle must not be passed to rpf_* again.)
 *   - rpf_apply_adjustments enqueues on the editor's HIP stream and returns;
 *     rpf_get_* / rpf_editor_sync wait (reference: submit returns before the GPU
 *     finishes, readback blocks; gpu_image_processing.rs:391-399, image.rs:249-259).
 *   - there is NO CPU fallback: without a usable HIP device rpf_processor_create
 *     fails with RPF_ERR_GPU_INITIALIZATION and nothing can be computed.
 */
#ifndef RPF_H
#define RPF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPF_CURVE_RESOLUTION 65536 /* lib.rs:17 */
/* An editor holds any number of masks (lib.rs:481-499 has no limit).  Up to this many — "main" + 15 user masks — run the
 * fused kernels from a Bayer frame: their coverage is ONE 16-bit word per pixel and their parameters travel with the launch.
 * With more, rpf_apply_adjustments runs the demosaic-only kernel and the RGBA pass (one 16-bit coverage plane per 16 masks,
 * parameters in device memory; the same bits) and so do rpf_dev_develop and a sharded step (rpf_shard_develop*) on a u16 band:
 * the band's linear intermediate belongs to the editor, which therefore serves one stream at a time then.  Packed bands
 * (rpf_dev_develop_packed, packed sharded steps) return RPF_ERR_INVALID_ARGUMENT with more than 16 masks: unpack first. */
#define RPF_FUSED_MASKS 16

typedef enum rpf_status {
    RPF_OK = 0,
    /* PhotoEditorError, errors.rs:7-49 */
    RPF_ERR_FAILED_TO_READ_IMAGE = 1,
    RPF_ERR_FAILED_TO_SAVE_IMAGE = 2,
    RPF_ERR_READ_IMAGE_UNSUPPORTED_FORMAT = 3,
    RPF_ERR_SAVE_IMAGE_UNSUPPORTED_FORMAT = 4,
    RPF_ERR_GPU_COMPUTE = 5,
    RPF_ERR_GPU_INITIALIZATION = 6,
    RPF_ERR_MASK_NOT_FOUND = 7,
    /* InterpolationError, errors.rs:105-130 (wrapped by PhotoEditorError::Interpolation) */
    RPF_ERR_INTERP_MISMATCHED_LENGTHS = 16,
    RPF_ERR_INTERP_NOT_ENOUGH_POINTS = 17,
    RPF_ERR_INTERP_NOT_STRICTLY_INCREASING = 18,
    RPF_ERR_INTERP_MISSING_CURVE_OR_CONTROL_POINTS = 19,
    RPF_ERR_INTERP_INVALID_CURVE_LENGTH = 20,
    RPF_ERR_INTERP_MISSING_CONTROL_POINTS = 21,
    RPF_ERR_INTERP_EMPTY_CONTROL_POINTS = 22,
    /* this ABI's own */
    RPF_ERR_INVALID_ARGUMENT = 32
} rpf_status;

typedef struct rpf_processor rpf_processor;
typedef struct rpf_editor rpf_editor;

/* CFA layouts: colour of (even row, even column), then its row neighbour. */
enum { RPF_CFA_RGGB = 0, RPF_CFA_BGGR = 1, RPF_CFA_GRBG = 2, RPF_CFA_GBRG = 3 };

/* RAW front-half parameters.  black/wb are indexed by RGGB cell position
 * R, G(red row), G(blue row), B.  cam2rgb is row major: out_i = sum_j cam2rgb[3i+j] cam_j. */
typedef struct rpf_raw_params {
    uint32_t cfa_pattern;
    float black[4];
    float white;
    float wb[4];
    float cam2rgb[9];
} rpf_raw_params;

/* GpuEditParameters, gpu_image_processing.rs:18-32 (introspection only) */
typedef struct rpf_gpu_edit_parameters {
    float r_gain, g_gain, b_gain;
    float vignette, lens_distortion;
    float exposure, contrast, shadow, highlight, black, white;
} rpf_gpu_edit_parameters;

/* Which curve, for rpf_editor_get_curve */
enum { RPF_CURVE_BRIGHTNESS = 0, RPF_CURVE_HUE = 1, RPF_CURVE_SATURATION = 2, RPF_CURVE_LIGHTNESS = 3 };

/* Output of the device-pointer layer */
enum { RPF_OUT_RGBA_F32 = 0, RPF_OUT_RGB_F32 = 1, RPF_OUT_RGB_U8 = 2 };

const char* rpf_last_error(void);
const char* rpf_version(void);

/* ---- processor: GpuProcessor ------------------------------------------------------------ */
int rpf_device_count(int* count);
int rpf_device_name(int device, char* buf, size_t buf_len); /* get_adapter_string_list */
int rpf_processor_create(int device, rpf_processor** out);  /* GpuProcessor::new(adapter_index) */
void rpf_processor_destroy(rpf_processor* p);                /* drop(Arc<GpuProcessor>): editors keep it alive */
int rpf_processor_device(const rpf_processor* p, int* device);
int rpf_processor_compute_units(const rpf_processor* p, int* cus);

/* ---- editor: PhotoEditor ----------------------------------------------------------------- */
/* Interleaved RGB f32, row major, top-left origin, width*height*3 values (note: the Rust
 * signature is (vec, height, width); the wasm wrapper is (data, width, height) — this ABI
 * follows the wasm order). */
int rpf_editor_create_from_rgb_f32(rpf_processor* p, const float* rgb, uint32_t width, uint32_t height,
                                   rpf_editor** out);
/* Decoded 8- / 16-bit interleaved RGB, what read_image holds behind the decoder (rust/photo-editor/src/image.rs:412-447):
 * to_rgb32f (:443) is v / 255 resp. v / 65535 — one binary32 division — and the upload pads alpha = 1 (:135-139); both
 * happen on the device here.  lut (256 resp. 65536 floats, or NULL): applied INSTEAD of the quotient, out = lut[v].  It is
 * how the caller renders the step between decode and conversion, apply_color_space(SRGB_LINEAR) for JPEG / PNG / WebP
 * (:431-441), whose arithmetic is inside image 0.25.10 / moxcms 0.8.1 and not in the reference tree (NULL = the TIFF
 * branch, :432).  An empty image gives RPF_ERR_FAILED_TO_READ_IMAGE. */
int rpf_editor_create_from_rgb_u8(rpf_processor* p, const uint8_t* rgb, uint32_t width, uint32_t height,
                                  const float* lut256, rpf_editor** out);
int rpf_editor_create_from_rgb_u16(rpf_processor* p, const uint16_t* rgb, uint32_t width, uint32_t height,
                                   const float* lut65536, rpf_editor** out);
/* u16 Bayer plane, pitch = width.  The editor keeps the CFA on the device; apply runs the
 * fused demosaic -> ... -> gamma kernel. */
int rpf_editor_create_from_bayer_u16(rpf_processor* p, const uint16_t* cfa, uint32_t width, uint32_t height,
                                     const rpf_raw_params* raw, rpf_editor** out);
/* 16-bit binary PPM ("P6\n<w> <h>\n65535\n" + big-endian RGB samples), the development front-end's test input
 * (web-ts/core/image.ts:146-195 loadPpm): sample / 65535 -> f32, alpha 1, converted on the device.  Same header
 * grammar as the reference (three newline-terminated lines, no comments); a magic other than P6, a maxval other
 * than 65535, or fewer than width*height*6 payload bytes give RPF_ERR_FAILED_TO_READ_IMAGE. */
int rpf_ppm16_parse_header(const uint8_t* file, size_t len, uint32_t* width, uint32_t* height, size_t* data_offset);
int rpf_editor_create_from_ppm16(rpf_processor* p, const uint8_t* file, size_t len, rpf_editor** out);
/* Operator state only (setters, curves, masks bookkeeping) with no device: for exercising the
 * state machine where no GPU exists.  apply / get_* fail with RPF_ERR_GPU_COMPUTE. */
int rpf_editor_create_headless(uint32_t width, uint32_t height, rpf_editor** out);
void rpf_editor_destroy(rpf_editor* e);

int rpf_width(const rpf_editor* e, uint32_t* width);
int rpf_height(const rpf_editor* e, uint32_t* height);

/* mask_name NULL = "main" (Option<&str> None) */
int rpf_set_whitebalance(rpf_editor* e, int32_t temperature, int32_t tint, const char* mask_name);
int rpf_set_vignette(rpf_editor* e, int32_t value);
int rpf_set_lens_distortion_correction(rpf_editor* e, int32_t value);
int rpf_set_tone(rpf_editor* e, float exposure, int32_t contrast, int32_t shadow, int32_t highlight, int32_t black,
                 int32_t white, const char* mask_name);

/* curve: NULL or curve_len values (must be 65536).  xs / ys: NULL or n_x / n_y control points. */
int rpf_set_brightness_tone_curve(rpf_editor* e, const int32_t* curve, size_t curve_len, const int32_t* xs,
                                  size_t n_x, const int32_t* ys, size_t n_y, const char* mask_name);
int rpf_set_oklch_hue_curve(rpf_editor* e, const int32_t* curve, size_t curve_len, const int32_t* xs, size_t n_x,
                            const int32_t* ys, size_t n_y, const char* mask_name);
int rpf_set_oklch_saturation_curve(rpf_editor* e, const int32_t* curve, size_t curve_len, const int32_t* xs,
                                   size_t n_x, const int32_t* ys, size_t n_y, const char* mask_name);
int rpf_set_oklch_lightness_curve(rpf_editor* e, const int32_t* curve, size_t curve_len, const int32_t* xs,
                                  size_t n_x, const int32_t* ys, size_t n_y, const char* mask_name);

/* mask: height*width f32, binarised v >= main.mask_range ? 1 : 0 (lib.rs:481-487).  No limit on their number (RPF_FUSED_MASKS). */
int rpf_add_mask(rpf_editor* e, const char* name, const float* mask);
int rpf_remove_mask(rpf_editor* e, const char* name); /* "main" is kept (lib.rs:501-505) */
int rpf_set_mask_range(rpf_editor* e, float mask_range); /* EditParameters.mask_range is a pub field */
int rpf_reset(rpf_editor* e);
int rpf_num_masks(const rpf_editor* e, uint32_t* n);

/* Preview pyramid (web/main.ts:652-695): a new editor over the source editor's ORIGINAL image resized so that its
 * long edge is target_long_edge, with the bilinear arithmetic of resizeFloat32RGBLongEdge (web/main.ts:958-1026),
 * done on the device (the reference reads the frame back, resizes on the CPU and uploads again). */
int rpf_preview_size(uint32_t src_width, uint32_t src_height, uint32_t target_long_edge, uint32_t* width, uint32_t* height);
int rpf_editor_create_preview(rpf_editor* src, uint32_t target_long_edge, rpf_editor** out);

/* Which of the reference's two shader generations apply runs.  ACTIVE = rust/photo-editor/src/wgpu_shader.wgsl.
 * GODOT_LEGACY = rust-godot-legacy/photo-editor/src/wgpu_shader.wgsl: the stored image is sRGB-encoded and is
 * linearised on the sampled input (:89-97, :281-283), and the vignette uses coordinates * 2.0, falloff exponent 4.0
 * and a [0, 2] gain clamp (:173-186).  Everything else is identical.  A Bayer editor's frame is linear already, so
 * for it only the vignette changes (and apply takes the two-pass route: the fused kernel is the active shader). */
enum { RPF_FLAVOUR_ACTIVE = 0, RPF_FLAVOUR_GODOT_LEGACY = 1 };
int rpf_set_shader_flavour(rpf_editor* e, int flavour);

/* How the builtins of the edit chain (pow, sqrt, atan2, cos, sin: wgpu_shader.wgsl:67, 70, 72, 78-79, 98) are evaluated.
 * WGSL leaves their accuracy to the backend; the reference's result is therefore defined only up to that.
 * RPF_PRECISION_EXACT (default): every builtin correctly rounded, every other operation one IEEE binary32 operation in
 *   source order — bit-identical to the parity oracle on every input (what all parity claims of this library refer to).
 * RPF_PRECISION_FAST: binary32 only.  pow, sqrt, cos, sin within 1 ulp of the correctly rounded value; the hue index from a
 *   binary32 polynomial (differs from the exact index, by one, within 0.005 index units of a boundary: about 0.1 % of
 *   directions); a * b + c contracted to an fma in the 3x3 colour matrices and the gamma's affine step; and where no mask has
 *   a hue curve (the reference default) the chroma / hue round trip — length, atan2, index, cos, sin — is replaced by the
 *   rotation of (a, b) through the index-quantisation residue that it amounts to (rpf_chain.h edit_chain_fast).  Everything
 *   up to and including the brightness curve stays bit-identical to EXACT.  Measured against the oracle on uniformly random
 *   frames: median 1-2 ulp per channel, 92-98 % of the pixels within 1e-6 and 99.6 % within 1e-5; the rest sit where the hue
 *   index steps (the hue moves by one step of the index scale there, in the reference too: within 2e-3 for the default
 *   curves; a steep per-hue curve scales that) (tests/test_fast_math.py, tests/test_gpu_fast.py).  One caveat is the
 *   reference's own: the hue of an ACHROMATIC pixel (grey, or clipped to white: a, b are rounding noise) is decided by that
 *   noise.  Its chroma does not care, but a per-hue LIGHTNESS curve that varies with the hue then hands such a pixel the gain
 *   of whatever index the noise picks — EXACT reproduces the oracle's pick bit for bit, FAST (like any backend whose pow
 *   rounds differently) may pick another entry of that curve (tools/soak_parity.py documents and excludes the case).  A hint: launches that
 *   have no binary32 specialisation (the lens resample, the Godot-legacy flavour, user masks or packed input with an output
 *   layout other than RGBA32F, user masks over packed input) compute the exact chain; a u16 Bayer band and the RGBA pass,
 *   each with or without user masks, and a packed band with only "main" have one.  Not versioned edit
 *   state: takes effect at the next apply / rpf_dev_* call.  No reference counterpart (wgpu has one precision, the driver's). */
enum { RPF_PRECISION_EXACT = 0, RPF_PRECISION_FAST = 1 };
int rpf_set_precision(rpf_editor* e, int precision);
int rpf_get_precision(const rpf_editor* e, int* precision);

int rpf_apply_adjustments(rpf_editor* e);
int rpf_editor_sync(rpf_editor* e);

/* Readback of the current image (the original until the first apply, like PhotoEditor.image). */
int rpf_get_rgba_f32(rpf_editor* e, float* out); /* width*height*4 */
int rpf_get_rgb_f32(rpf_editor* e, float* out);  /* width*height*3 */
int rpf_get_rgb_u8(rpf_editor* e, uint8_t* out); /* (v.clamp(0,1)*255) as u8, image.rs:375-383 */

/* Introspection used by the parity tests */
int rpf_editor_get_gpu_params(const rpf_editor* e, const char* mask_name, rpf_gpu_edit_parameters* out);
int rpf_editor_get_curve(const rpf_editor* e, int which, const char* mask_name, int32_t* out65536);

/* ---- stateless host helpers --------------------------------------------------------------- */
/* interpolation.rs:11-123 on i32 arrays; bad_index receives the index for NOT_STRICTLY_INCREASING. */
int rpf_pchip_interpolate_i32(const int32_t* x_pts, size_t n_x, const int32_t* y_pts, size_t n_y,
                              const int32_t* x_eval, size_t n_eval, int32_t* y_eval, int32_t* bad_index);

/* ---- device-pointer layer ------------------------------------------------------------------
 * Row tiles: a frame of frame_width x frame_height is cut into bands of rows [y0, y0+rows).
 * d_cfa points at a (rows + 4) x frame_width u16 buffer whose row r holds frame row y0 - 2 + r
 * (two halo rows above and below the band).  Halo rows outside the frame are never read: the
 * frame edge is mirrored (reflect-101) inside the kernel.  y0 and rows need not be even.
 * d_out: rows x frame_width pixels in `out_format`.  stream: a hipStream_t (NULL = default).
 */
typedef struct rpf_tile {
    uint32_t frame_width, frame_height;
    uint32_t y0, rows;
} rpf_tile;

/* Streaming read of a device buffer that was just written by someone else (an upload, a decoder, a halo exchange):
 * lands it in the GPU's 256-MiB Infinity Cache.  Optional and value-neutral — results never depend on it; a frame that
 * comes from HBM costs the fused kernel more than this pass does.  No reference counterpart (wgpu leaves
 * residency to the driver). */
int rpf_dev_prefetch(rpf_processor* p, const void* d_buf, size_t bytes, void* stream);

/* Packed RAW -> u16 CFA plane (integer, bit-exact).  order RPF_PACK_MSB_FIRST: sample i of a row occupies bits
 * [i*bits, (i+1)*bits) counted from the most significant bit of byte 0 (a big-endian bit stream);
 * RPF_PACK_LSB_FIRST: the same bit positions counted from the least significant bit of byte 0, least significant
 * sample bit first (the row read as one little-endian integer);
 * RPF_PACK_MIPI: MIPI CSI-2 RAW10 / RAW12 / RAW14 — groups of 4 / 2 / 4 pixels in 5 / 3 / 7 bytes, the pixels' high
 * bytes first, then their bits - 8 low bits LSB-first in pixel order; a row holds whole groups (bits 16 is refused).
 * Rows start every row_stride bytes.
 * bits in {10, 12, 14, 16}.  d_cfa_out has pitch `width` (write it at row 2 of a band buffer).  In the reference
 * this step is inside rawler / LibRaw (rust-godot-legacy/photo-editor/src/image.rs:516-523). */
enum { RPF_PACK_MSB_FIRST = 0, RPF_PACK_LSB_FIRST = 1, RPF_PACK_MIPI = 2 };
int rpf_dev_unpack_bayer(rpf_processor* p, const uint8_t* d_packed, size_t row_stride, uint32_t width, uint32_t rows,
                         uint32_t bits, uint32_t order, uint16_t* d_cfa_out, void* stream);
/* npix decoded RGB triplets -> RGBA32F on the device: v / 255 resp. v / 65535 (image.rs:443), or d_lut[v] when a table
 * (256 resp. 65536 floats in device memory) is given; alpha 1.  See rpf_editor_create_from_rgb_u8. */
int rpf_dev_rgb_u8_to_rgba_f32(rpf_processor* p, const uint8_t* d_rgb, size_t npix, const float* d_lut256,
                               float* d_rgba_out, void* stream);
int rpf_dev_rgb_u16_to_rgba_f32(rpf_processor* p, const uint16_t* d_rgb, size_t npix, const float* d_lut65536,
                                float* d_rgba_out, void* stream);
/* npix big-endian 16-bit RGB triplets (2-byte aligned) -> RGBA32F, v / 65535, alpha 1 (image.ts:172-189) */
int rpf_dev_rgb16be_to_rgba_f32(rpf_processor* p, const uint8_t* d_rgb16be, size_t npix, float* d_rgba_out,
                                void* stream);
/* black level -> WB -> 5x5 demosaic -> camera matrix; linear RGBA32F out */
int rpf_dev_demosaic(rpf_processor* p, const uint16_t* d_cfa, const rpf_raw_params* raw, const rpf_tile* tile,
                     float* d_rgba_out, void* stream);
/* fused RAW -> display chain with the editor's current edit state (masks/curves/params).
 * The editor must have been created for the tile's size (rows x frame_width) or headless-with-device
 * via rpf_editor_create_for_tile. */
int rpf_dev_develop(rpf_editor* e, const uint16_t* d_cfa, const rpf_raw_params* raw, void* d_out, int out_format,
                    void* stream);
/* The same with the integer unpack folded into the kernel's row loads: d_packed holds the band's (rows + 4) rows in the
 * packed layout of rpf_dev_unpack_bayer (any of its orders and widths; RPF_PACK_MIPI rows need row_stride >= 8 and, like every row,
 * row_stride readable bytes: the kernel reads a CSI-2 group through an 8-byte window),
 * row r = frame row y0 - 2 + r, rows row_stride bytes apart.  Same output bits as rpf_dev_unpack_bayer followed by rpf_dev_develop, one pass less. */
int rpf_dev_develop_packed(rpf_editor* e, const uint8_t* d_packed, size_t row_stride, uint32_t bits, uint32_t order,
                           const rpf_raw_params* raw, void* d_out, int out_format, void* stream);
/* the reference-equivalent back half: RGBA32F tile in -> out (wgsl main) */
int rpf_dev_apply(rpf_editor* e, const float* d_rgba_in, void* d_out, int out_format, void* stream);

/* An editor that owns edit state + device LUTs/masks for a row tile but no image. */
int rpf_editor_create_for_tile(rpf_processor* p, const rpf_tile* tile, rpf_editor** out);

/* ---- row-tile sharding of ONE frame over the GPUs of a node -----------------------------------
 * Replaces the single dispatch of GpuProcessor::apply_adjustments (gpu_image_processing.rs:385-391; callers
 * photo-editor-web/src/lib.rs:198, photo-editor-godot/src/lib.rs:257-271) when a frame is cut into contiguous row bands, one
 * per process and GPU: with lens_distortion = 0 a pixel needs its own sample and a 5x5 CFA neighbourhood, so a band needs
 * the two Bayer rows above and below it from its neighbours and nothing else.  An rpf_shard is one rank's end of that
 * exchange: an RCCL communicator (librccl is loaded at rpf_shard_create, not linked: a single-GPU host never touches it)
 * and (from the first rpf_shard_develop_async on) a thread of its own that makes the RCCL calls.  One call per frame,
 * rpf_shard_develop, enqueues the whole step on `stream` and returns: ncclGroupStart, ncclSend / ncclRecv of the 2-row halos
 * with the ranks above and below, ncclGroupEnd, then the fused kernel over the band.  (Rounds 2-3 also carried two schedules
 * that hide the exchange behind cross-stream dependencies — interior / edge rows on a side stream, and the exchange one
 * frame ahead; measured on one MI355X both lose to what follows at every band size, DESIGN.md section 6, and round 4 removed
 * them.)  The band buffer has the layout of rpf_dev_develop: (rows + 4) rows, row r = frame row y0 - 2 + r; the exchange
 * writes rows 0, 1 and rows + 2, rows + 3 (never at a frame edge: the kernel mirrors there) and sends rows 2, 3 and rows,
 * rows + 1.  The editor is the band's (rpf_editor_create_for_tile with the band's y0 / rows; masks are band-sized).
 * lens_distortion != 0 does not shard (whole-frame resample): refused, as by rpf_dev_develop.
 *
 * rpf_shard_unique_id: ncclGetUniqueId — call on ONE rank and hand the 128 bytes to the others (any channel: a file, a
 * socket, MPI, torch.distributed).  rpf_shard_create is collective (ncclCommInitRank): every rank of `world` calls it with
 * the same id.  world = 1 is valid and needs no id (NULL): no communicator, rpf_shard_develop is a plain launch unless
 * neighbours are set.  rpf_shard_set_neighbours overrides the ranks above / below (default rank - 1 / rank + 1, -1 = frame
 * edge); up = down = rank makes a rank exchange with itself (its first rows land in its own top halo, its last rows in its
 * bottom halo): the one-GPU rehearsal of the whole step, RCCL included. */
#define RPF_SHARD_ID_BYTES 128
typedef struct rpf_shard rpf_shard;
enum { RPF_SHARD_NO_EXCHANGE = 1u }; /* halo rows are already in place (the caller exchanged or pre-filled them) */
enum { RPF_SHARD_TIMED = 0x100u };           /* rpf_shard_develop_async: time this step's launch (rpf_shard_timing) */
int rpf_shard_unique_id(uint8_t id[RPF_SHARD_ID_BYTES]);
int rpf_shard_create(rpf_processor* p, const uint8_t* id, int rank, int world, rpf_shard** out);
void rpf_shard_destroy(rpf_shard* s);
int rpf_shard_set_neighbours(rpf_shard* s, int up, int down);
/* rank / world / neighbours as the communicator sees them, and the PCI bus id of the shard's device ("0000:05:00.0") */
int rpf_shard_info(const rpf_shard* s, int* rank, int* world, int* up, int* down, char* pci_bus_id, size_t len);
/* the halo exchange alone, on `stream` (row_bytes = bytes of one band row; rows = band rows without the halos) */
int rpf_shard_exchange(rpf_shard* s, void* d_band, size_t row_bytes, uint32_t rows, void* stream);
/* One frame step of this rank.  bits = 0: d_band is the u16 band (row_stride ignored); bits in {10, 12, 14, 16}: packed
 * rows as in rpf_dev_develop_packed (their halo rows are packed rows too). */
int rpf_shard_develop(rpf_shard* s, rpf_editor* e, void* d_band, size_t row_stride, uint32_t bits, uint32_t order,
                      const rpf_raw_params* raw, void* d_out, int out_format, void* stream, uint32_t flags);
/* The same step (exchange on `stream`, then one launch on `stream`), handed to the shard's own thread: the call checks its
 * arguments, uploads pending edit state, copies what the launch needs and returns; the thread makes the RCCL calls (~16 us
 * of host time) and the launch, in the order the steps were handed over.  Meant for a stream of frames through TWO shards —
 * two communicators (rpf_shard_create twice, in the same order on every rank), two streams, two band buffers, frames
 * alternating between them: the two step pipelines share nothing, so one frame's exchange, launch latency and tail run
 * beside the other frame's kernel with no cross-stream dependency at all (which cost this runtime 4-13 us per step each:
 * DESIGN.md section 6).  Measured on one MI355X with the rank as its own neighbour: a 6000 x 500 band 25.8 us per step —
 * the kernel alone — against 39 for rpf_shard_develop on one stream and 35 for the exchange one frame ahead.  Give the two
 * streams DIFFERENT priorities: with equal ones the pipelines can lock in phase (both exchanges at once, then both kernels
 * sharing the CUs: 37 us per step); a priority makes that state unstable and they settle in antiphase.
 * flags: RPF_SHARD_NO_EXCHANGE, RPF_SHARD_TIMED (the launch's own start / completion signals are a pair of timing events,
 * summed up by rpf_shard_timing).  A step takes the edit state as it is at the call (by value).  Setters, rpf_add_mask and
 * rpf_editor_destroy called while steps are still queued are safe: whatever would free or overwrite a device table first
 * waits until the shard's thread has enqueued every step that references it, then for the device.  Output and band
 * buffers are the caller's and must stay valid until the step's stream has passed it.
 * rpf_shard_flush: blocks until every step handed over has been ENQUEUED on its stream (then hipStreamSynchronize or an
 * event of the caller's is meaningful) and reports the first failure of such a step.  EVERY wait of this layer is bounded
 * (rpf_shard_set_timeout, default RPF_WAIT_TIMEOUT_MS = 30 s; 0 = unbounded): on the deadline the call returns
 * RPF_ERR_GPU_COMPUTE and rpf_last_error names the rank, its neighbours, the step that did not get enqueued and what the
 * shard's thread was doing (the halo exchange: a neighbour never posted its half; the launch).
 * rpf_shard_drain: flush, then wait — polling, with the same deadline — until `stream` has drained: the give-up-able
 * replacement of hipStreamSynchronize behind a run of async steps (a step whose neighbour never arrives stalls on the
 * DEVICE; after this error the process should report and exit, not retry).
 * rpf_shard_timing: flush, wait for the timed launches, return their number and summed duration since the last call
 * (at most 4096 launches are kept between two calls; further RPF_SHARD_TIMED steps run untimed). */
int rpf_shard_develop_async(rpf_shard* s, rpf_editor* e, void* d_band, size_t row_stride, uint32_t bits, uint32_t order,
                            const rpf_raw_params* raw, void* d_out, int out_format, void* stream, uint32_t flags);
int rpf_shard_flush(rpf_shard* s);
int rpf_shard_drain(rpf_shard* s, void* stream);
int rpf_shard_set_timeout(rpf_shard* s, int timeout_ms);
int rpf_shard_timing(rpf_shard* s, uint32_t* launches, double* total_ms);

/* Kernel variant selection for A/B measurements (bench.py --variant); 0 = shipped configuration.
 * fused develop: 1 = LDS-tile front end (first design), 2 = 512-thread workgroups, 3 = pixel pair sequentially,
 * 6 = 4 instead of 5 waves per SIMD, 7 = the general multi-mask kernel also when only "main" exists,
 * 8 = no splitting of the last, incomplete round of work items, 9 = 8-row instead of 32-/16-row work items,
 * 10 = brightness curve gathered from L2 instead of staged in LDS, 11 = cos / sin of the hue evaluated (sincos_hue)
 * also when the hue curve is the identity (instead of read with the hue-index table entry);
 * demosaic: 1 = LDS-tile front end, 3 = ordinary (not nontemporal) stores, 4 = 16-row items;
 * RGBA pass with more than RPF_FUSED_MASKS masks: 24 = the masks' parameters read from device memory instead of staged in
 * LDS (what editors with more than 1024 masks run).  Same output bits. */
int rpf_set_kernel_variant(rpf_processor* p, int variant);

#ifdef __cplusplus
}
#endif
#endif /* RPF_H */
