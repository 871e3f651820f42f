// rpf_launch.h — host-callable launchers of the gfx950 kernels (rpf_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include "rpf_chain.h"

namespace rpf {

// Edit state as the kernels see it (all device pointers owned by the caller).
struct DeviceEdit {
    MaskParams mp[RPF_MASKS_PER_WORD];  // the first 16 masks' ("main" + 15 user masks); all of them: mp_wide
    uint32_t num_masks;
    // One u16 per pixel of the tile: bit m - 1 set <=> user mask m covers the pixel (its binarised value is 1.0,
    // lib.rs:483-486); null when there are no user masks.  ONE load per pixel whatever the number of masks — per-mask u8
    // planes cost the kernel a dependent round trip to memory per mask per step.
    // More than RPF_MASKS_PER_WORD masks: ceil(num_masks / 16) such planes, rows x width elements apart; bit j of plane
    // g >= 1 <=> mask 16 g + j (rpf_chain.h WideMasks).
    const uint16_t* mask_bits;
    const uint16_t* bright;      // num_masks x 65536 u16, clamped (wgsl:188)
    const ushort4* hsl;          // num_masks x 65536 {hue, sat, light, 0}
    const MathTables* tabs;
    const HueBound* hue_bounds;  // 65536 boundary directions of the hue index (rpf_math.h hue_index)
    const HueBoundCs* hue_cs;    // the same boundaries with cos / sin of the hue angles attached (HUE_DIRECT launches)
    // One mask with hue / saturation / lightness curves that are not all trivial: the boundaries with what "main"'s
    // curves give at each index attached (rpf_math.h HueFused; per editor, rebuilt when those curves change).  Null
    // otherwise.
    const HueFused* hue_fused;
    float vign_strength;         // (-vignette / 100) * 2, 0 = off       wgsl:167
    // rows x width binary32: the vignette's falloff at every pixel of the tile (vignette_falloff_from_axes: a correctly
    // rounded square root, quotient and pow(t, 1.5) per pixel that depend on the pixel's position alone), computed once
    // per editor when the slider first leaves 0 (launch_vignette_falloff).  k_hp reads it (4 B per pixel instead of ~40
    // instructions); the other kernels evaluate the falloff in place.  Null while vign_strength == 0.
    const float* vign_falloff;
    float lens_strength;         // -0.5 * (lens_distortion / 100), 0 = off   wgsl:110
    // Godot-legacy shader flavour (k_apply only): EF_* bits, and the c = 2.4 table when EF_DECODE_INPUT is set
    uint32_t flags;
    const LegacyTables* legacy;
    // rpf_set_precision: 0 = every builtin correctly rounded (the default), 1 = RPF_PRECISION_FAST — launches that have a
    // binary32-only specialisation (rpf_fast.h) take it and read these tables instead of `tabs`
    uint32_t precision;
    const FastTables* fast_tabs;
    // Which mask's brightness curve a launch that stages one in LDS (128 KiB: there is room for ONE) stages: 0 = "main"; a
    // user mask's index when "main"'s is the identity (chosen per launch by launch_develop: the first mask that has one)
    uint32_t lds_bright_mask;
    // num_masks > RPF_MASKS_PER_WORD (the reference has no limit, lib.rs:481-499): every mask's parameters, in device
    // memory (the kernel-argument segment holds 16).  Such an editor runs the RGBA pass (launch_apply), from a Bayer frame
    // after the demosaic-only kernel; the fused kernels take up to 16 masks.  Null otherwise.
    const MaskParams* mp_wide;
};
enum EditFlags : uint32_t { EF_LEGACY_VIGNETTE = 1u, EF_DECODE_INPUT = 2u };

struct TileGeom {
    uint32_t width, height;  // full frame
    uint32_t y0, rows;       // this tile
};

enum OutFormat { OUT_RGBA_F32 = 0, OUT_RGB_F32 = 1, OUT_RGB_U8 = 2 };

struct LaunchCfg {
    int num_cus;
    int variant;  // 0 default; see rpf_kernels.hip
    // fused develop launches only: an event that becomes the launch's own completion signal (hipExtLaunchKernelGGL's
    // stopEvent) — what a later cross-stream wait can name without a marker packet behind the kernel
    void* stop_event = nullptr;
    void* start_event = nullptr;  // with stop_event: the pair times the kernel itself (rpf_shard_develop_async, RPF_SHARD_TIMED)
};

// d_cfa: (rows + 4) x width u16, row r = frame row y0 - 2 + r
// d_table[0 .. 65535] <- the fused hue table of an editor whose only mask is `main`
hipError_t launch_build_hue_fused(const HueBound* d_hue_bounds, const ushort4* d_hsl_main, const MaskParams& main,
                                  const MathTables* d_tabs, HueFused* d_table, hipStream_t s);
// d_map[r * width + x] = vignette_falloff_from_axes at frame pixel (x, y0 + r): DeviceEdit::vign_falloff of that tile
hipError_t launch_vignette_falloff(const TileGeom& g, float* d_map, hipStream_t s);
hipError_t launch_demosaic(const LaunchCfg& cfg, const uint16_t* d_cfa, const RawParamsDev& raw, const TileGeom& g,
                           float* d_rgba, hipStream_t s);
hipError_t launch_develop(const LaunchCfg& cfg, const uint16_t* d_cfa, const RawParamsDev& raw, const TileGeom& g,
                          const DeviceEdit& ed, void* d_out, int out_format, hipStream_t s);
// the same with the integer unpack folded in: d_packed holds (rows + 4) bit-packed rows (10/12/14/16 bits), `order` =
// RPF_PACK_MSB_FIRST (big-endian bit stream), RPF_PACK_LSB_FIRST (little-endian) or RPF_PACK_MIPI (CSI-2 groups: rows need
// row_stride >= 8 readable bytes, the kernel reads a group through an 8-byte window); all three inside the fused kernel
hipError_t launch_develop_packed(const LaunchCfg& cfg, const uint8_t* d_packed, uint32_t row_stride, uint32_t bits,
                                 uint32_t order, const RawParamsDev& raw, const TileGeom& g, const DeviceEdit& ed, void* d_out,
                                 int out_format, hipStream_t s);
// d_in: rows x width RGBA32F (whole frame when lens_strength != 0)
hipError_t launch_apply(const LaunchCfg& cfg, const float* d_in, const TileGeom& g, const DeviceEdit& ed, void* d_out,
                        int out_format, hipStream_t s);
// packed RAW (10/12/14/16 bits; MSB-first or LSB-first bit stream, or MIPI CSI-2 RAW10/12/14 groups, per `order`) -> u16 plane of pitch `width`
hipError_t launch_unpack(const uint8_t* d_packed, size_t row_stride, uint32_t width, uint32_t rows, uint32_t bits,
                         uint32_t order, uint16_t* d_out, hipStream_t s);
// streaming read of a freshly written buffer (lands it in the Infinity Cache ahead of the fused kernel)
hipError_t launch_touch(const void* d_buf, size_t bytes, int num_cus, hipStream_t s);
// bilinear resize of the preview pyramid (web/main.ts:958-1026), RGBA32F -> RGBA32F
hipError_t launch_resize(const float* d_src, uint32_t sw, uint32_t sh, float* d_dst, uint32_t dw, uint32_t dh, hipStream_t s);
// RGBA32F -> RGB f32 / RGB u8 repack (image.rs:202-277, 375-383)
hipError_t launch_repack(const float* d_rgba, size_t npix, void* d_out, int out_format, hipStream_t s);
// (planes) u8 mask planes of npix pixels, value 1 = covered -> the u16 bit plane DeviceEdit::mask_bits points at
hipError_t launch_combine_masks(const uint8_t* d_planes, uint32_t planes, size_t npix, uint16_t* d_bits, hipStream_t s);
// interleaved RGB f32 -> RGBA32F, alpha = 1 (image.rs:135-139)
hipError_t launch_pad_rgb(const float* d_rgb, size_t npix, float* d_rgba, hipStream_t s);
// decoded 8- / 16-bit interleaved RGB -> RGBA32F: v / 255, v / 65535, or d_lut[v] (256 / 65536 floats on the device)
hipError_t launch_rgb_int(const void* d_in, int bits, size_t npix, const float* d_lut, float* d_rgba, hipStream_t s);
hipError_t launch_rgb16be(const uint8_t* d_in, size_t npix, float* d_rgba, hipStream_t s);

}  // namespace rpf
