// rpf_chain.h — the per-pixel arithmetic of the hot path, shared by the gfx950
// kernels (rpf_kernels.hip) and by a host build used ONLY by tests/ as a twin of
// the device code (tests/twin/).  Every binary32 operation is written in the order
// of rust/photo-editor/src/wgpu_shader.wgsl and must be compiled with
// -ffp-contract=off; the transcendental builtins come from rpf_math.h.
#pragma once
#include "rpf_fast.h"
#include "rpf_math.h"

namespace rpf {

// Masks whose coverage travels in ONE word per pixel ("main" + 15 user masks: DeviceEdit::mask_bits) and whose parameters
// travel in the kernel-argument segment.  An editor may hold more (the reference has no limit, lib.rs:481-499): their
// coverage lies in further 16-bit planes, 16 masks each (WideMasks below).
#define RPF_MASKS_PER_WORD 16
#define RPF_CURVE_RESOLUTION 65536  // lib.rs:17

enum MaskFlags : uint32_t {
    MF_BLACK = 1u << 0,            // black != 0      wgsl:234
    MF_WHITE = 1u << 1,            // white != 0      wgsl:241
    MF_CONTRAST = 1u << 2,         // contrast != 0   wgsl:248
    MF_BRIGHT_IDENTITY = 1u << 3,  // brightness_curve[i] == i for all i (after the fetch clamp)
    MF_HUE_IDENTITY = 1u << 4,     // hue_curve[i] == i
    MF_SAT_CONST = 1u << 5,        // saturation_curve constant -> sat_gain
    MF_LIGHT_CONST = 1u << 6,      // lightness_curve constant -> light_gain
    // exact no-ops the kernel may skip: x * 1.0f == x and 1.0f + 0.0f * m == 1.0f for every finite m
    MF_UNITY_WB = 1u << 7,         // r_gain == g_gain == b_gain == 1
    MF_UNITY_EXPOSURE = 1u << 8,   // pow(2, exposure) == 1
    MF_NO_SHADOW_HIGHLIGHT = 1u << 9,  // shadow == 0 and highlight == 0
};

// One mask's uniform parameters: GpuEditParameters (gpu_image_processing.rs:18-32)
// with the per-pixel-invariant sub-expressions of wgsl tone() evaluated once on the
// host in the same binary32 arithmetic.
struct MaskParams {
    float r_gain, g_gain, b_gain;   // wgsl:285-288
    float exposure_mul;             // pow(2.0, exposure), correctly rounded   wgsl:214
    float shadow, highlight;        // wgsl:227, 231
    float black, white;             // wgsl:234-245
    float contrast_c;               // 1.0 + contrast                          wgsl:250
    uint32_t flags;
    float sat_gain, light_gain;     // f32(v) / 32767.5 when the curve is constant  wgsl:329-330
};

struct RawParamsDev {
    float black[4], range[4], scale[4];  // per RGGB cell position (cx + 2 cy)
    float cam[9];                        // row major camera -> linear sRGB
    int xo, yo;                          // CFA phase
};

RPF_HD float clampf(float v, float lo, float hi)
{
    const float m = v > lo ? v : lo;  // WGSL clamp = min(max(e, low), high)
    return m < hi ? m : hi;
}

// ---------------------------------------------------------------------------
// RAW front half (this build's definition; see oracle/rpf_oracle.c orc_demosaic)
// ---------------------------------------------------------------------------

RPF_HD int reflect101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

RPF_HD float cfa_normalise(uint32_t raw, int cell, const RawParamsDev& rp)
{
    const float n = clampf((float)raw - rp.black[cell], 0.0f, rp.range[cell]);
    return n * rp.scale[cell];
}

// The formulas as DESIGN.md §3 states them, e.g. cross = ((4 c + 2 (s1 + e1)) - (v2 + h2)) / 8, one binary32 rounding
// per operation.  Every factor that is a power of two scales exactly (no operand here is subnormal: samples are 0 or
// >= 1e-5 and the sums cancel to 0 or to a multiple of their ulp), and RN(2^k x) = 2^k RN(x), so the division by 8 can
// be carried into the terms and each "power of two times a term, plus a term" becomes ONE fma whose product is exact:
// the same roundings in the same order with 19 instead of 29 instructions for the four values of a pixel pair.
RPF_HD float mhc_cross(float c, float s1, float e1, float v2, float h2)
{
    // ((4 c + 2 t) - u) / 8  ==  RN(RN(0.25 t + 0.5 c) - 0.125 u)
    const float x = __builtin_fmaf(0.25f, s1 + e1, 0.5f * c);
    return __builtin_fmaf(-0.125f, v2 + h2, x);
}
RPF_HD float mhc_diag(float c, float d, float v2, float h2)
{
    // ((6 c + 2 d) - 1.5 u) / 8  ==  RN(RN(0.25 d + RN(0.75 c)) - RN(0.1875 u))
    return __builtin_fmaf(0.25f, d, 0.75f * c) - 0.1875f * (v2 + h2);
}
RPF_HD float mhc_axis(float c, float along, float d, float same2, float other2)
{
    // horiz: along = e1, same2 = h2, other2 = v2;  vert: along = s1, same2 = v2, other2 = h2
    // ((((5 c + 4 along) - d) - same2) + 0.5 other2) / 8, the division carried through every term
    float x = __builtin_fmaf(0.5f, along, 0.625f * c);
    x = __builtin_fmaf(-0.125f, d, x);
    x = __builtin_fmaf(-0.125f, same2, x);
    return __builtin_fmaf(0.0625f, other2, x);
}

RPF_HD void cam_to_linear(const float* m, float R, float G, float B, float& r, float& g, float& b)
{
    r = m[0] * R + m[1] * G + m[2] * B;
    g = m[3] * R + m[4] * G + m[5] * B;
    b = m[6] * R + m[7] * G + m[8] * B;
}

// ---------------------------------------------------------------------------
// wgsl:166-178 vignette — returns the gain; x, y are GLOBAL pixel coordinates
// ---------------------------------------------------------------------------
// The three steps separately, so that a kernel can hoist what is constant along a column / a row:
//   vignette_axis2(i, n) = (((f32(i) / n) - 0.5) * 1.5)^2 for one axis, vignette_from_axes = the rest.
RPF_HD float vignette_axis2(int i, float n)
{
    const float c = ((float)i / n - 0.5f) * 1.5f;
    return c * c;
}
// The part that depends on the pixel's position alone (a kernel may read it from a map computed once per tile:
// DeviceEdit::vign_falloff) ...
RPF_HD float vignette_falloff_from_axes(float cx2, float cy2)
{
    const float dist = sqrt_cr(cx2 + cy2);
    // (dist - 0.25) / 0.75 by Markstein's sequence: equal to the IEEE quotient (every binary32 in [-0.25, 2] checked,
    // tests/test_math_exhaustive.py)
    const float t = div_by_const(dist - 0.25f, 0.75f, (float)(1.0 / 0.75));
    return pow_1p5(clampf(t, 0.0f, 1.0f));
}
// ... and the part that depends on the slider.
RPF_HD float vignette_gain_from_falloff(float vign_strength, float falloff)
{
    return clampf(1.0f - (vign_strength * falloff), 0.0f, 4.0f);
}
RPF_HD float vignette_from_axes(float vign_strength, float cx2, float cy2)
{
    return vignette_gain_from_falloff(vign_strength, vignette_falloff_from_axes(cx2, cy2));
}
RPF_HD float vignette_gain(float vign_strength, int x, int y, float w, float h)
{
    return vignette_from_axes(vign_strength, vignette_axis2(x, w), vignette_axis2(y, h));
}

// The Godot-legacy flavour of the same shader (rust-godot-legacy/photo-editor/src/wgpu_shader.wgsl:173-186):
// coordinates scaled by 2.0, falloff exponent 4.0, gain clamped to [0, 2].
RPF_HD float vignette_gain_legacy(float vign_strength, int x, int y, float w, float h)
{
    const float cx = ((float)x / w - 0.5f) * 2.0f;
    const float cy = ((float)y / h - 0.5f) * 2.0f;
    const float dist = __builtin_sqrtf(cx * cx + cy * cy);
    const float falloff = pow_4(clampf((dist - 0.25f) / 0.75f, 0.0f, 1.0f));
    return clampf(1.0f - (vign_strength * falloff), 0.0f, 2.0f);
}

// Godot-legacy :89-97 srgb_to_linear, applied to the sampled input before everything else (:281-283)
RPF_HD float srgb_to_linear1(float c, const PowTable& decode)
{
    const float lin = c / 12.92f;
    const float p = powc<POW_DECODE>((c + 0.055f) / 1.055f, decode);
    return c <= 0.04045f ? lin : p;
}

// ---------------------------------------------------------------------------
// wgsl:109-164 lens_distortion_sample, strength != 0 branch.
// fetch(x, y, rgb[3]) reads one RGBA32F texel of the whole frame.
// ---------------------------------------------------------------------------
RPF_HD float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

template <class Fetch>
RPF_HD void lens_sample(const Fetch& fetch, int w_u, int h_u, int x, int y, float strength, float& r, float& g,
                        float& b)
{
    const float w = (float)w_u, h = (float)h_u;
    const float uvx = (float)x / w, uvy = (float)y / h;
    const float cux = uvx - 0.5f, cuy = uvy - 0.5f;
    const float aspect = w / h;
    const float kx = cux * aspect, ky = cuy * 1.0f;
    const float r2 = kx * kx + ky * ky;
    const float den = 1.0f + strength * r2;
    const float dx = kx / den, dy = ky / den;
    const float fx = (dx / aspect) + 0.5f, fy = (dy / 1.0f) + 0.5f;
    if (fx < 0.0f || fy < 0.0f || fx > 1.0f || fy > 1.0f) {
        r = g = b = 0.0f;
        return;
    }
    const float px = fx * (w - 1.0f), py = fy * (h - 1.0f);
    const float x0f = __builtin_floorf(px), y0f = __builtin_floorf(py);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = x0 + 1 < w_u - 1 ? x0 + 1 : w_u - 1;
    const int y1 = y0 + 1 < h_u - 1 ? y0 + 1 : h_u - 1;
    const float tx = px - x0f, ty = py - y0f;
    float c00[3], c10[3], c01[3], c11[3];
    fetch(x0, y0, c00);
    fetch(x1, y0, c10);
    fetch(x0, y1, c01);
    fetch(x1, y1, c11);
    float o[3];
    for (int k = 0; k < 3; ++k) o[k] = mixf(mixf(c00[k], c10[k], tx), mixf(c01[k], c11[k], tx), ty);
    r = o[0]; g = o[1]; b = o[2];
}

// ---------------------------------------------------------------------------
// wgsl main :279-336 — everything after the input sample and the vignette.
//
//   Luts::brightness(m, idx)               -> u32 in [0, 65535]   (lut_fetch(0u, ..))
//   Luts::hsl(m, idx, hue, sat_f, light_f) -> hue u32, gains = f32(v)/32767.5
//   active: bit m set <=> masks_tex(xy, m) == 1.0 (bit 0, "main", is all ones: lib.rs:193-196)
// ---------------------------------------------------------------------------
#define RPF_F(x) ((float)(x))

// clamp for finite inputs: one v_med3_f32 on the device, the oracle's min(max(v, lo), hi) on the host;
// the two agree on every non-NaN value up to the sign of zero, which no later operation of the chain
// observes.  (Measured issue cost on gfx950: med3 1.8 ns, max+min 3.6 ns, cmp+cndmask x2 7 ns.)
RPF_HD float clamp01(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f);
#else
    return clampf(v, 0.0f, 1.0f);
#endif
}

RPF_HD float max0(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaxf(v, 0.0f);
#else
    return v > 0.0f ? v : 0.0f;
#endif
}

// The flag word of a mask whose every stage is the reference default (lib.rs:44-64: a photo that was just opened,
// BASELINE's edit set D): nothing to multiply, identity / constant curves.  Kernels instantiated with
// STATIC_FLAGS = MF_PRISTINE know all of this at compile time; STATIC_FLAGS < 0 reads the flags at run time.
constexpr uint32_t MF_PRISTINE = MF_BRIGHT_IDENTITY | MF_HUE_IDENTITY | MF_SAT_CONST | MF_LIGHT_CONST | MF_UNITY_WB |
                                 MF_UNITY_EXPOSURE | MF_NO_SHADOW_HIGHLIGHT;

// One mask's linear-RGB stage (wgsl:283-307) on N pixels in lockstep.  act == nullptr: the mask is
// "main" (all ones); otherwise pixel n is updated only where act[n] is set (wgsl:281 `continue`).
template <int N, int STATIC_FLAGS = -1, class Luts>
RPF_HD void mask_rgb_stage(const MaskParams& PP, uint32_t m, const Luts& luts, float (&r)[N], float (&g)[N],
                           float (&b)[N], const bool* act)
{
    struct Flags { uint32_t flags; };
    const Flags P{STATIC_FLAGS >= 0 ? (uint32_t)STATIC_FLAGS : PP.flags};
    const float inv65535 = (float)(1.0 / 65535.0);
    // Three phases over all N pixels — tone, curve fetch, result — rather than pixel after pixel: the flag tests are
    // wave-uniform branches, and with the pixels' curve gathers in ONE branch they are issued back to back and waited
    // for once (pixel by pixel, every user mask cost two dependent round trips to the curve table per step).
    float xk[N], yk[N], zk[N];
    for (int n = 0; n < N; ++n) {
        float x = r[n], y = g[n], z = b[n];
        // White balance (wgsl:285-288) and exposure (wgsl:214-215).  x * 1.0f == x exactly, so the unity flags only
        // matter where they are known at compile time: tested at run time the compiler executes the multiplications
        // anyway and adds a select each (measured: 12 + 12 half-rate instructions per pixel pair).
        if (STATIC_FLAGS < 0 || !(P.flags & MF_UNITY_WB)) { x *= PP.r_gain; y *= PP.g_gain; z *= PP.b_gain; }
        // tone(), wgsl:200-259
        if (STATIC_FLAGS < 0 || !(P.flags & MF_UNITY_EXPOSURE)) { x *= PP.exposure_mul; y *= PP.exposure_mul; z *= PP.exposure_mul; }
        float highlight_mask = 0.0f;
        if ((P.flags & (MF_NO_SHADOW_HIGHLIGHT | MF_BLACK | MF_WHITE)) != MF_NO_SHADOW_HIGHLIGHT) {
            const float luma = x * RPF_F(0.2126) + y * RPF_F(0.7152) + z * RPF_F(0.0722);
            highlight_mask = clamp01(luma);
            if (!(P.flags & MF_NO_SHADOW_HIGHLIGHT)) {
                const float shadow_mask = clamp01(1.0f - luma);
                const float shadow_gain = 1.0f + PP.shadow * shadow_mask;
                x *= shadow_gain; y *= shadow_gain; z *= shadow_gain;
                const float highlight_gain = 1.0f + PP.highlight * highlight_mask;
                x *= highlight_gain; y *= highlight_gain; z *= highlight_gain;
            }
        }
        // Black and white points: x + 0.0f == x for every x the chain can hold here (never -0: each row of the camera matrix
        // has a positive coefficient), so with run-time flags the two stages run unconditionally, like white balance
        // and exposure above — tested per launch the compiler if-converts them into three additions AND three selects.
        if (STATIC_FLAGS < 0 || (P.flags & MF_BLACK)) {
            const float t1 = 1.0f - highlight_mask;
            const float add = PP.black * (t1 * t1);  // pow(1 - t, 2.0) correctly rounded
            x += add; y += add; z += add;
        }
        if (STATIC_FLAGS < 0 || (P.flags & MF_WHITE)) {
            const float add = PP.white * (highlight_mask * highlight_mask);
            x += add; y += add; z += add;
        }
        if (P.flags & MF_CONTRAST) {
            x = (x - 0.5f) * PP.contrast_c + 0.5f;
            y = (y - 0.5f) * PP.contrast_c + 0.5f;
            z = (z - 0.5f) * PP.contrast_c + 0.5f;
        }
        x = clamp01(x); y = clamp01(y); z = clamp01(z);
        xk[n] = x * 65535.0f; yk[n] = y * 65535.0f; zk[n] = z * 65535.0f;
    }
    // brightness curve, wgsl:293-305
    // f32(u32(v)) for v in [0, 65535] is trunc(v): with the identity curve no integer is needed at all
    if (!(P.flags & MF_BRIGHT_IDENTITY)) {
        uint32_t xi[N], yi[N], zi[N];
        for (int n = 0; n < N; ++n) {
            xi[n] = luts.brightness(m, (uint32_t)xk[n]);
            yi[n] = luts.brightness(m, (uint32_t)yk[n]);
            zi[n] = luts.brightness(m, (uint32_t)zk[n]);
        }
        for (int n = 0; n < N; ++n) { xk[n] = (float)xi[n]; yk[n] = (float)yi[n]; zk[n] = (float)zi[n]; }
    } else {
        for (int n = 0; n < N; ++n) {
            xk[n] = __builtin_truncf(xk[n]); yk[n] = __builtin_truncf(yk[n]); zk[n] = __builtin_truncf(zk[n]);
        }
    }
    for (int n = 0; n < N; ++n) {
        const float x = div_u16f_by(xk[n], 65535.0f, inv65535);
        const float y = div_u16f_by(yk[n], 65535.0f, inv65535);
        const float z = div_u16f_by(zk[n], 65535.0f, inv65535);
        const bool on = act ? act[n] : true;
        r[n] = on ? x : r[n]; g[n] = on ? y : g[n]; b[n] = on ? z : b[n];
    }
}

// wgsl main :279-336 on N pixels in lockstep (same operations per pixel, interleaved so that
// constants are materialised once and the binary64 chains of different pixels overlap).
// In place: r,g,b hold linear RGB on entry and the clamped display RGB on exit.
// HUE_DIRECT (the caller's promise: num_masks == 1 and "main"'s hue curve is the identity): the index the hue
// becomes IS the new hue, and cos / sin of its angle come with the boundary-table entry (hue_index_cs) instead of
// being evaluated by sincos_hue.  A compile-time choice: as a run-time branch it splits the pair's gathers again.
// Where a mask's uniform parameters come from.  A plain array (the host twin, single-mask kernels) ...
RPF_HD const MaskParams& mask_params_main(const MaskParams* mp) { return mp[0]; }
RPF_HD const MaskParams& mask_params_user(const MaskParams* mp, uint32_t m) { return mp[m]; }
// ... or, in the multi-mask kernels, a source that serves the user masks' parameters from LDS (rpf_kernels.hip
// LdsMaskParams): read through the kernel-argument pointer inside the per-mask loop, every field is a scalar load of
// its own with its own wait, per mask per pixel.

// Constants the kernels want in vector registers: the operand of an operation that carries the clamp modifier (VOP3:
// no literal on gfx9, so the compiler parks it in a scalar register, and an SGPR operand halves the f32 rate), and
// "main"'s constant saturation / lightness gains.  Any Tabs type without overloads of its own (the host twin's
// MathTables) gets the plain values; rpf_kernels.hip overloads them for LdsTabs.
template <class Tabs> RPF_HD float chain_k_gamma_lin(const Tabs&) { return RPF_F(12.92); }
template <class Tabs> RPF_HD float chain_k_gamma_off(const Tabs&) { return RPF_F(0.055); }
template <class Tabs> RPF_HD float chain_k_gamma_mul(const Tabs&) { return RPF_F(1.055); }
template <class Tabs> RPF_HD float chain_main_sat(const Tabs&, const MaskParams& P) { return P.sat_gain; }
template <class Tabs> RPF_HD float chain_main_light(const Tabs&, const MaskParams& P) { return P.light_gain; }

// One entry of the fused hue table (rpf_math.h HueFused) of an editor whose only mask is `P`: boundary j and what the chain
// takes from "main"'s curves at index j — the Luts::hsl quotients and sincos_hue of the hue the curve maps j to.  The
// table is built with this function (rpf_kernels.hip k_build_hue_fused; the host twin), a launch that reads it has the
// bits of one that evaluates the same expressions per pixel.
template <class SinCos>
RPF_HD HueFused hue_fused_entry(const HueBound& hb, uint32_t hue_v, uint32_t sat_v, uint32_t light_v, const MaskParams& P,
                                const SinCos& S)
{
    HueFused e;
    e.c = hb.c; e.s = hb.s;
    e.sat_gain = (P.flags & MF_SAT_CONST) ? P.sat_gain : div_u16_by(sat_v, 32767.5f, (float)(1.0 / 32767.5));
    e.light_gain = (P.flags & MF_LIGHT_CONST) ? P.light_gain : div_u16_by(light_v, 32767.5f, (float)(1.0 / 32767.5));
    sincos_hue(hue_v, S, e.cs, e.sn);
    return e;
}

// Masks beyond the first RPF_MASKS_PER_WORD (edit_chain's `wide` argument).  Mask m >= 16 is bit m % 16 of plane m / 16,
// planes `stride` elements apart (plane 0 is the first word's: bit j = user mask j + 1); their parameters and curves are
// slot m of the same arrays as everybody's (Params, Luts).  NoWideMasks: the editor has none — nothing is instantiated.
struct NoWideMasks {
    static constexpr bool enabled = false;
};
template <int N>
struct WideMasks {
    static constexpr bool enabled = true;
    uint32_t total;           // masks of the editor, "main" included (> RPF_MASKS_PER_WORD)
    const uint16_t* planes;
    size_t stride;
    size_t pix[N];            // the N pixels' index into a plane
    RPF_HD uint32_t word(int n, uint32_t plane) const { return planes[(size_t)plane * stride + pix[n]]; }
};

// The coverage words of plane `plane` at the N pixels, and (returned) which of its masks cover a pixel of this wavefront.
template <int N, class Wide>
RPF_HD uint32_t wide_plane_words(const Wide& wide, uint32_t plane, uint32_t (&w)[N])
{
    uint32_t any = 0;
    for (int n = 0; n < N; ++n) { w[n] = wide.word(n, plane); any |= w[n]; }
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t u = 0;
    for (uint32_t j = 0; j < RPF_MASKS_PER_WORD; ++j)
        u |= (__builtin_amdgcn_ballot_w64((any >> j) & 1u) != 0ull) ? (1u << j) : 0u;
    any = u;
#endif
    const uint32_t left = wide.total - plane * RPF_MASKS_PER_WORD;  // masks the plane holds, if it is the last
    return left < RPF_MASKS_PER_WORD ? any & ((1u << left) - 1u) : any;
}

// STATIC_FLAGS (>= 0): "main"'s flag word, known at compile time (the caller's promise that mp[0].flags equals it).
// num_masks: the masks of the first word (<= RPF_MASKS_PER_WORD); `wide`: the others (WideMasks), in mask order after them.
template <int N, int HUE_MODE = 0, int STATIC_FLAGS = -1, class Params, class Luts, class Tabs, class Wide = NoWideMasks>
RPF_HD void edit_chain(float (&r)[N], float (&g)[N], float (&b)[N], const uint32_t (&active)[N],
                       const Params& mp, uint32_t num_masks, const Luts& luts, const Tabs& T, const Wide& wide = Wide())
{
    const float inv65535 = (float)(1.0 / 65535.0);
    mask_rgb_stage<N, STATIC_FLAGS>(mask_params_main(mp), 0u, luts, r, g, b, nullptr);  // "main" is all ones (lib.rs:193-196)
    // bits of the user masks that cover at least one pixel of this wavefront: a mask that covers none is skipped
    // as a whole (wgsl:281 `continue` for every lane) — masks are regions, most wavefronts lie inside or outside
    uint32_t any_active = 0;
    for (int n = 0; n < N; ++n) any_active |= active[n];
#if defined(__HIP_DEVICE_COMPILE__)
    {
        uint32_t u = 0;
        for (uint32_t m = 1; m < num_masks; ++m)
            u |= (__builtin_amdgcn_ballot_w64((any_active >> m) & 1u) != 0ull) ? (1u << m) : 0u;
        any_active = u | 1u;
    }
#endif
    for (uint32_t m = 1; m < num_masks; ++m) {
        if (!((any_active >> m) & 1u)) continue;
        bool act[N];
        for (int n = 0; n < N; ++n) act[n] = (active[n] >> m) & 1u;
        mask_rgb_stage<N>(mask_params_user(mp, m), m, luts, r, g, b, act);
    }
    // the masks of the further planes, a plane at a time
    if constexpr (Wide::enabled) {
        static_assert(HUE_MODE == 0, "more than one mask: the general hue path");
        for (uint32_t plane = 1; plane * RPF_MASKS_PER_WORD < wide.total; ++plane) {
            uint32_t w[N];
            const uint32_t any = wide_plane_words(wide, plane, w);
            for (uint32_t j = 0; j < RPF_MASKS_PER_WORD; ++j) {
                if (!((any >> j) & 1u)) continue;
                const uint32_t m = plane * RPF_MASKS_PER_WORD + j;
                bool act[N];
                for (int n = 0; n < N; ++n) act[n] = (w[n] >> j) & 1u;
                mask_rgb_stage<N>(mask_params_user(mp, m), m, luts, r, g, b, act);
            }
        }
    }

    // linear_srgb_to_oklch, wgsl:64-75.  The hue is only ever used as a curve index (hue_index, rpf_math.h).
    // HUE_MODE: 0 = any masks and curves; 1 = HUE_DIRECT (see above); 2 = HUE_FUSED: one mask whose curves are folded
    // into the hue table (rpf_math.h HueFused): the index, then cos / sin of the NEW hue and both gains from its entry
    constexpr bool HUE_DIRECT = HUE_MODE == 1, HUE_FUSED = HUE_MODE == 2;
    float L[N], C[N], h[N], cs0[N], sn0[N], sg0[N], lg0[N];
    uint32_t hu0[N];
    for (int n = 0; n < N; ++n) {
        const float l = RPF_F(0.4122214708) * r[n] + RPF_F(0.5363325363) * g[n] + RPF_F(0.0514459929) * b[n];
        const float mm = RPF_F(0.2119034982) * r[n] + RPF_F(0.6806995451) * g[n] + RPF_F(0.1073969566) * b[n];
        const float s = RPF_F(0.0883024619) * r[n] + RPF_F(0.2817188376) * g[n] + RPF_F(0.6299787005) * b[n];
        // pow(max(lms, 0), 1/3), wgsl:67.  The max is the identity here: r, g, b are k/65535 >= 0 after the
        // brightness stage and every M1 coefficient is positive, so l, mm, s >= +0.
        const float lc = powc<POW_CBRT>(l, T.cbrt);
        const float mc = powc<POW_CBRT>(mm, T.cbrt);
        const float sc = powc<POW_CBRT>(s, T.cbrt);
        L[n] = RPF_F(0.2104542553) * lc + RPF_F(0.7936177850) * mc + RPF_F(-0.0040720468) * sc;
        const float A = RPF_F(1.9779984951) * lc + RPF_F(-2.4285922050) * mc + RPF_F(0.4505937099) * sc;
        const float Bq = RPF_F(0.0259040371) * lc + RPF_F(0.7827717662) * mc + RPF_F(-0.8086757660) * sc;
        C[n] = sqrt_cr(A * A + Bq * Bq);
        // u32(h * 65535), h = atan2(b, a) / (2 pi) wrapped (wgsl:72-73, 320).  A is never -0 here: the cube roots are
        // >= +0, so its first product is +0 or positive and a round-to-nearest sum that cancels exactly is +0 —
        // hue_index's atan2(+-0, -0) argument cannot occur (NEG_ZERO_X = false).
        sg0[n] = lg0[n] = 1.0f;
        if constexpr (HUE_FUSED) {
            hu0[n] = hue_index_fused<false>(Bq, A, luts.hue_fused(), cs0[n], sn0[n], sg0[n], lg0[n]);
        } else if constexpr (HUE_DIRECT) {
            hu0[n] = hue_index_cs<false>(Bq, A, luts.hue_bounds_cs(), T.sc, cs0[n], sn0[n]);
        } else {
            hu0[n] = hue_index<false>(Bq, A, luts.hue_bounds());
            cs0[n] = sn0[n] = 0.0f;
        }
    }

    // per-mask OKLCH curves, wgsl:312-331
    uint32_t hue_k[N];
    auto oklch_stage = [&](uint32_t m, bool main_mask) {
        const MaskParams P = main_mask ? mask_params_main(mp) : mask_params_user(mp, m);
        const uint32_t pflags = (main_mask && STATIC_FLAGS >= 0) ? (uint32_t)STATIC_FLAGS : P.flags;
        const bool trivial = (pflags & (MF_HUE_IDENTITY | MF_SAT_CONST | MF_LIGHT_CONST)) ==
                             (MF_HUE_IDENTITY | MF_SAT_CONST | MF_LIGHT_CONST);
        for (int n = 0; n < N; ++n) {
            const uint32_t hu = main_mask ? hu0[n] : (uint32_t)(h[n] * 65535.0f);
            uint32_t nh = hu;
            float sat_gain = main_mask ? chain_main_sat(T, P) : P.sat_gain;
            float light_gain = main_mask ? chain_main_light(T, P) : P.light_gain;
            if (!trivial) luts.hsl(m, hu, P, nh, sat_gain, light_gain);
            const bool on = main_mask ? true : (bool)((active[n] >> m) & 1u);
            hue_k[n] = on ? nh : hue_k[n];
            h[n] = on ? div_u16_by(nh, 65535.0f, inv65535) : h[n];
            C[n] = on ? C[n] * sat_gain : C[n];
            L[n] = on ? L[n] * light_gain : L[n];
        }
    };
    if constexpr (HUE_FUSED) {  // "main"'s curves came with the table entry: only the two products are left
        for (int n = 0; n < N; ++n) { C[n] = C[n] * sg0[n]; L[n] = L[n] * lg0[n]; }
    } else {
        oklch_stage(0u, true);  // "main" is always active and replaces the hue by curve[k] / 65535
    }
    for (uint32_t m = 1; m < num_masks; ++m)
        if ((any_active >> m) & 1u) oklch_stage(m, false);
    if constexpr (Wide::enabled) {  // oklch_stage for the masks of the further planes
        for (uint32_t plane = 1; plane * RPF_MASKS_PER_WORD < wide.total; ++plane) {
            uint32_t w[N];
            const uint32_t any = wide_plane_words(wide, plane, w);
            for (uint32_t j = 0; j < RPF_MASKS_PER_WORD; ++j) {
                if (!((any >> j) & 1u)) continue;
                const uint32_t m = plane * RPF_MASKS_PER_WORD + j;
                const MaskParams P = mask_params_user(mp, m);
                const bool trivial = (P.flags & (MF_HUE_IDENTITY | MF_SAT_CONST | MF_LIGHT_CONST)) ==
                                     (MF_HUE_IDENTITY | MF_SAT_CONST | MF_LIGHT_CONST);
                for (int n = 0; n < N; ++n) {
                    const uint32_t hu = (uint32_t)(h[n] * 65535.0f);
                    uint32_t nh = hu;
                    float sat_gain = P.sat_gain, light_gain = P.light_gain;
                    if (!trivial) luts.hsl(m, hu, P, nh, sat_gain, light_gain);
                    const bool on = (bool)((w[n] >> j) & 1u);
                    hue_k[n] = on ? nh : hue_k[n];
                    h[n] = on ? div_u16_by(nh, 65535.0f, inv65535) : h[n];
                    C[n] = on ? C[n] * sat_gain : C[n];
                    L[n] = on ? L[n] * light_gain : L[n];
                }
            }
        }
    }

    // oklch_to_linear_srgb (wgsl:77-84), linear_to_srgb + clamp (wgsl:95-103, 335-336).  The hue reaching
    // here is always k/65535 for the integer k the last active mask fetched ("main" is always active).
    for (int n = 0; n < N; ++n) {
        float cs = cs0[n], sn = sn0[n];
        if (!HUE_DIRECT && !HUE_FUSED) sincos_hue(hue_k[n], T.sc, cs, sn);
        const float oa = C[n] * cs, ob = C[n] * sn;
        const float l_ = RPF_F(1.0) * L[n] + RPF_F(0.3963377774) * oa + RPF_F(0.2158037573) * ob;
        const float m_ = RPF_F(1.0) * L[n] + RPF_F(-0.1055613458) * oa + RPF_F(-0.0638541728) * ob;
        const float s_ = RPF_F(1.0) * L[n] + RPF_F(-0.0894841775) * oa + RPF_F(-1.2914855480) * ob;
        const float l3 = l_ * l_ * l_, m3 = m_ * m_ * m_, s3 = s_ * s_ * s_;
        const float lin[3] = {RPF_F(4.0767416621) * l3 + RPF_F(-3.3077115913) * m3 + RPF_F(0.2309699292) * s3,
                              RPF_F(-1.2684380046) * l3 + RPF_F(2.6097574011) * m3 + RPF_F(-0.3413193965) * s3,
                              RPF_F(-0.0041960863) * l3 + RPF_F(-0.7034186147) * m3 + RPF_F(1.7076147010) * s3};
        float out[3];
        for (int k = 0; k < 3; ++k) {
            const float c = lin[k];
            // clamp(select(a, b)) == select(clamp(a), clamp(b)); written this way each clamp is the free output
            // modifier of the multiplication / addition in front of it instead of a v_max of its own
            const float hi = clamp01(RPF_F(1.055) * powc<POW_GAMMA>(c, T.gamma) - chain_k_gamma_off(T));  // unused when c <= thr
            const float lo = clamp01(c * chain_k_gamma_lin(T));
            out[k] = c <= RPF_F(0.0031308) ? lo : hi;
        }
        r[n] = out[0]; g[n] = out[1]; b[n] = out[2];
    }
}

// ---------------------------------------------------------------------------
// RPF_PRECISION_FAST (include/rpf.h rpf_set_precision): wgsl main :279-336 with the builtins of rpf_fast.h.
//
// What stays as in edit_chain, bit for bit: everything up to and including the brightness curve (mask_rgb_stage: plain
// binary32 arithmetic in the reference's order, the u32 conversions, the curve look-ups, k / 65535 as the IEEE quotient)
// — so the brightness-curve indices never differ from the oracle's, and with identity OKLCH curves a pixel changes only
// through the roundings below.  What changes after it:
//   * pow, sqrt, cos / sin: within 1 ulp of the correctly rounded value instead of correctly rounded (rpf_fast.h);
//   * the hue index from an eight-term binary32 polynomial alone: equal to the reference's index except within 0.005 index
//     units of a boundary (0.1 % of uniformly distributed directions; the neighbouring index then);
//   * a * b + c is contracted to one fma in the 3x3 matrix products, the chroma and the gamma's affine step — as a WGSL
//     implementation may (and every GPU backend does) — in the reference's left-to-right order.
// No binary64 instruction, no table outside LDS (the per-hue curves of a mask that has them are the one gather left).
// ---------------------------------------------------------------------------
RPF_HD float dot3_fast(float c0, float c1, float c2, float x, float y, float z)
{
    return __builtin_fmaf(c2, z, __builtin_fmaf(c1, y, c0 * x));
}

// HUE_ROT (the caller's promise: every mask's hue curve is the identity — the reference default, every edit set without a
// hue curve): the chain then needs neither the chroma nor an angle.  Replacing the hue h by f32(u32(h 65535)) / 65535 turns
// (a, b) through the residue eps = (h 65535 - u32(h 65535)) 2 pi / 65535 < 9.6e-5, so what wgsl:77-84 computes as
// C cos(2 pi h'), C sin(2 pi h') with C = length(a, b) is (a + b eps, b - a eps) up to eps^2 / 2 < 5e-9 relative — no sqrt,
// no atan2, no cos, no sin, and nothing that can flip: eps is continuous in (a, b) except where the index steps, where
// the reference's own result steps by the same amount.  Against the builtin route: eps is known to 5e-8 rad (rpf_fast.h
// hue_fraction_fast) and the reference rounds its angle to binary32 (2.4e-7 rad at 2 pi), so the two differ by up to
// C 3e-7, the size of the other roundings of this mode.  Saturation gains scale (a, b), lightness gains L, as before; a
// mask with saturation / lightness CURVES still needs the index for its look-up (the full position, wave-uniform branch).
// HUE_TAB (the caller's promise: ONE mask, with hue / saturation / lightness curves, whose fused hue table exists —
// rpf_math.h HueFused, built per editor for the exact kernels): the index comes from hue_position_fast, and cos / sin of the
// hue the curve maps it to and both gains are read from the table entry's 16 data bytes (the exact values; the boundary
// half of the entry is not touched) instead of a curve gather, two quotients and sincos_hue_fast.
template <int N, int STATIC_FLAGS = -1, bool HUE_ROT = false, bool HUE_TAB = false, class Params, class Luts, class Tabs>
RPF_HD void edit_chain_fast(float (&r)[N], float (&g)[N], float (&b)[N], const uint32_t (&active)[N], const Params& mp,
                            uint32_t num_masks, const Luts& luts, const Tabs& T)
{
    mask_rgb_stage<N, STATIC_FLAGS>(mask_params_main(mp), 0u, luts, r, g, b, nullptr);
    uint32_t any_active = 0;
    for (int n = 0; n < N; ++n) any_active |= active[n];
#if defined(__HIP_DEVICE_COMPILE__)
    {
        uint32_t u = 0;
        for (uint32_t m = 1; m < num_masks; ++m)
            u |= (__builtin_amdgcn_ballot_w64((any_active >> m) & 1u) != 0ull) ? (1u << m) : 0u;
        any_active = u | 1u;
    }
#endif
    for (uint32_t m = 1; m < num_masks; ++m) {
        if (!((any_active >> m) & 1u)) continue;
        bool act[N];
        for (int n = 0; n < N; ++n) act[n] = (active[n] >> m) & 1u;
        mask_rgb_stage<N>(mask_params_user(mp, m), m, luts, r, g, b, act);
    }

    // linear_srgb_to_oklch, wgsl:64-75
    float L[N], C[N], hk[N];  // hk: the hue as the integer k of h = k / 65535, held as binary32
    float A[N], Bq[N];        // HUE_ROT: (a, b) themselves; C is then the product of the saturation gains
    for (int n = 0; n < N; ++n) {
        const float l = dot3_fast(RPF_F(0.4122214708), RPF_F(0.5363325363), RPF_F(0.0514459929), r[n], g[n], b[n]);
        const float mm = dot3_fast(RPF_F(0.2119034982), RPF_F(0.6806995451), RPF_F(0.1073969566), r[n], g[n], b[n]);
        const float s = dot3_fast(RPF_F(0.0883024619), RPF_F(0.2817188376), RPF_F(0.6299787005), r[n], g[n], b[n]);
        const float lc = powc_fast<POW_CBRT>(l, T.cbrt);
        const float mc = powc_fast<POW_CBRT>(mm, T.cbrt);
        const float sc = powc_fast<POW_CBRT>(s, T.cbrt);
        L[n] = dot3_fast(RPF_F(0.2104542553), RPF_F(0.7936177850), RPF_F(-0.0040720468), lc, mc, sc);
        A[n] = dot3_fast(RPF_F(1.9779984951), RPF_F(-2.4285922050), RPF_F(0.4505937099), lc, mc, sc);
        Bq[n] = dot3_fast(RPF_F(0.0259040371), RPF_F(0.7827717662), RPF_F(-0.8086757660), lc, mc, sc);
        if constexpr (HUE_ROT) {
            C[n] = 1.0f;
            hk[n] = -1.0f;  // the index: only if a mask's saturation / lightness curve asks for it (below)
        } else {
            C[n] = sqrt_fast(__builtin_fmaf(Bq[n], Bq[n], A[n] * A[n]));
            hk[n] = __builtin_truncf(hue_position_fast(Bq[n], A[n]));  // f32(u32(h * 65535)), wgsl:320
        }
    }

    // per-mask OKLCH curves, wgsl:312-331
    bool have_index = !HUE_ROT;
    auto oklch_stage = [&](uint32_t m, bool main_mask) {
        const MaskParams P = main_mask ? mask_params_main(mp) : mask_params_user(mp, m);
        const uint32_t pflags = (main_mask && STATIC_FLAGS >= 0) ? (uint32_t)STATIC_FLAGS : P.flags;
        const bool trivial = (pflags & (MF_HUE_IDENTITY | MF_SAT_CONST | MF_LIGHT_CONST)) ==
                             (MF_HUE_IDENTITY | MF_SAT_CONST | MF_LIGHT_CONST);
        if (HUE_ROT && !trivial && !have_index) {  // wave-uniform: the first mask with a saturation / lightness curve
            for (int n = 0; n < N; ++n) hk[n] = __builtin_truncf(hue_position_fast(Bq[n], A[n]));
            have_index = true;
        }
        for (int n = 0; n < N; ++n) {
            // a user mask re-derives the index from the hue the previous mask left, u32(f32(k / 65535) * 65535) — which is
            // k again for every k (tests/test_fast_math.py), so the integer is carried instead
            const float hu_f = hk[n];
            float nh_f = hu_f;
            float sat_gain = main_mask ? chain_main_sat(T, P) : P.sat_gain;
            float light_gain = main_mask ? chain_main_light(T, P) : P.light_gain;
            if (!trivial) {
                uint32_t nh;
                luts.hsl(m, (uint32_t)hu_f, P, nh, sat_gain, light_gain);
                nh_f = (float)nh;  // (HUE_ROT: the identity)
            }
            const bool on = main_mask ? true : (bool)((active[n] >> m) & 1u);
            if (!HUE_ROT) hk[n] = on ? nh_f : hk[n];
            C[n] = on ? C[n] * sat_gain : C[n];
            L[n] = on ? L[n] * light_gain : L[n];
        }
    };
    float tcs[N], tsn[N];
    if constexpr (HUE_TAB) {
        static_assert(!HUE_ROT, "a hue curve: the builtin route");
        for (int n = 0; n < N; ++n) {
            const HueFused* e = luts.hue_fused() + (uint32_t)hk[n];
            tcs[n] = e->cs; tsn[n] = e->sn;
            C[n] = C[n] * e->sat_gain;
            L[n] = L[n] * e->light_gain;
        }
    } else {
        oklch_stage(0u, true);
        for (uint32_t m = 1; m < num_masks; ++m)
            if ((any_active >> m) & 1u) oklch_stage(m, false);
    }

    // oklch_to_linear_srgb (wgsl:77-84), linear_to_srgb + clamp (wgsl:95-103, 335-336)
    for (int n = 0; n < N; ++n) {
        float oa, ob;
        if constexpr (HUE_TAB) {
            oa = C[n] * tcs[n]; ob = C[n] * tsn[n];
        } else if constexpr (HUE_ROT) {
            const float eps = hue_fraction_fast(Bq[n], A[n]) * RPF_K_HUE_STEP_F;
            oa = __builtin_fmaf(Bq[n], eps, A[n]) * C[n];
            ob = __builtin_fmaf(-A[n], eps, Bq[n]) * C[n];
        } else {
            float cs, sn;
            sincos_hue_fast(hk[n], cs, sn);
            oa = C[n] * cs; ob = C[n] * sn;
        }
        const float l_ = __builtin_fmaf(RPF_F(0.2158037573), ob, __builtin_fmaf(RPF_F(0.3963377774), oa, L[n]));
        const float m_ = __builtin_fmaf(RPF_F(-0.0638541728), ob, __builtin_fmaf(RPF_F(-0.1055613458), oa, L[n]));
        const float s_ = __builtin_fmaf(RPF_F(-1.2914855480), ob, __builtin_fmaf(RPF_F(-0.0894841775), oa, L[n]));
        const float l3 = l_ * l_ * l_, m3 = m_ * m_ * m_, s3 = s_ * s_ * s_;
        const float lin[3] = {dot3_fast(RPF_F(4.0767416621), RPF_F(-3.3077115913), RPF_F(0.2309699292), l3, m3, s3),
                              dot3_fast(RPF_F(-1.2684380046), RPF_F(2.6097574011), RPF_F(-0.3413193965), l3, m3, s3),
                              dot3_fast(RPF_F(-0.0041960863), RPF_F(-0.7034186147), RPF_F(1.7076147010), l3, m3, s3)};
        float out[3];
        for (int k = 0; k < 3; ++k) {
            const float c = lin[k];
            const float hi = clamp01(srgb_upper_fast(c, T.gamma));  // 1.055 pow(c, 1 / 2.4) - 0.055 in one table evaluation
            const float lo = clamp01(c * chain_k_gamma_lin(T));
            out[k] = c <= RPF_F(0.0031308) ? lo : hi;
        }
        r[n] = out[0]; g[n] = out[1]; b[n] = out[2];
    }
}

}  // namespace rpf
