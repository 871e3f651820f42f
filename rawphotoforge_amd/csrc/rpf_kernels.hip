// rpf_kernels.hip — hand-written gfx950 (CDNA4) kernels of the RAW development hot path.
//
//   k_stream<MODE_DEMOSAIC>   u16 Bayer -> linear RGBA32F          (BASELINE config 2)
//   k_hp<MODE_DEVELOP...>     u16 Bayer -> display RGBA32F, fused  (BASELINE config 3: demosaic + WB + camera matrix + the
//                              whole of wgsl main in ONE pass: 2 B in, 16 B out per pixel) — the horizontal-pair kernel,
//                              shipped since round 3 for launches with only "main"
//   k_stream<MODE_DEVELOP...> the same chain, column-per-lane front end: packed input, launches with user masks
//   k_tile<...>               the same two, LDS-tile front end (variant 1; the first design, kept for A/B)
//   k_apply                   RGBA32F -> RGBA32F, wgsl main only   (what the reference's wgpu pass does); its SINGLE = 7 / 8
//                              specialisations serve editors with more masks than one coverage word holds (any number:
//                              lib.rs:481-499 has no limit) — from a Bayer frame after the demosaic-only kernel
//   k_hp / k_stream / k_apply <..., PREC = 1>: the RPF_PRECISION_FAST specialisations (binary32-only builtins, rpf_fast.h;
//                              the default, PREC = 0, evaluates every builtin correctly rounded and is the parity gate)
//
// Replaces rust/photo-editor/src/wgpu_shader.wgsl:265-337 (kernel) and the dispatch half of
// rust/photo-editor/src/gpu_image_processing.rs:214-400.  Not a translation of the WGSL
// 16x16 one-thread-per-texel pass:
//   * k_stream: a wavefront owns a 64-column strip (STRIP_W) and walks down it two rows at a time.  Each lane
//     keeps a rolling 6-row window of ITS column in registers (two coalesced 128-B row loads per step, every CFA
//     sample normalised once), takes the horizontal taps of the 5x5 stencil from its neighbours with DPP
//     wavefront shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1) and carries the horizontal pair sums of a
//     row down the window, so the stencil costs no LDS traffic and no barrier at all.  The four samples per row
//     beyond the wavefront's 64 columns come from one extra "halo" load per row, held by lanes 0,1 (columns
//     x0-2, x0-1) and 62,63 (x0+64, x0+65): all 64 lanes produce output.
//   * a vertical pixel pair always holds one R/B site and one G site, so both demosaic formula sets run
//     without divergence, and the pair goes through the edit chain in lockstep (rpf_chain.h);
//   * the transcendental tables (28 KiB) and, when it is not the identity, the 65536-entry brightness
//     curve (u16, 128 KiB) are staged in LDS once per workgroup, so curve gathers never leave the CU;
//   * every store is a nontemporal global_store_dwordx4, one aligned KiB per wave-instruction.
// Pointwise/stencil path: HBM-bound by construction, no MFMA.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <string.h>

#include <atomic>

#include "rpf_launch.h"

namespace rpf {

// _SEQ (A/B only): the pair one pixel after the other.  _1: only the "main" mask exists (the common case): the
// per-mask loops and the mask-plane reads are compiled out, which also frees the scalar registers they pin.
// _1H: _1 whose hue curve is the identity as well (edit_chain's HUE_DIRECT).  _1P: _1H whose "main" is pristine
// (MF_PRISTINE: every slider and curve at the reference default — a photo that was just opened; BASELINE's set D):
// the flag word is a compile-time constant, so no stage tests, multiplies by or selects on a default.
enum { MODE_DEMOSAIC = 0, MODE_DEVELOP = 1, MODE_DEVELOP_SEQ = 2, MODE_DEVELOP_1 = 3, MODE_DEVELOP_1H = 4,
       MODE_DEVELOP_1P = 5,
       MODE_DEVELOP_MP = 6,    // _MP: user masks over a pristine "main" (its flag word static, as in _1P)
       MODE_DEVELOP_MPH = 7 }; // _MPH: _MP whose user masks all have identity hue curves as well: HUE_DIRECT (the hue
                               // index "main" settles survives every mask: u32(f32(k / 65535) * 65535) == k for all k)
constexpr bool mode_multi(int mode) { return mode == MODE_DEVELOP || mode == MODE_DEVELOP_MP || mode == MODE_DEVELOP_MPH; }
constexpr bool mode_single(int mode) { return mode == MODE_DEVELOP_1 || mode == MODE_DEVELOP_1H || mode == MODE_DEVELOP_1P; }

// The math tables as the kernels address them: by LDS byte address (rpf_math.h LdsPowTable).  `base` = LDS address of
// the staged MathTables; the sin/cos tables (corner cases and non-identity hue curves only) stay an ordinary reference.
struct LdsTabs {
    LdsPowTable cbrt, gamma;
    const SinCosTable& sc;
    // launch-long constants in vector registers (rpf_chain.h chain_k_*, rpf_math.h pow_c3)
    float k_gamma_lin, k_gamma_off, main_sat, main_light;
    // pin: hold them in vector registers (k_hp, k_apply: four waves per SIMD, 128 VGPRs); the column-per-lane kernels
    // run five waves per SIMD on 96 and keep the values as literals / scalar operands, as before
    __device__ __forceinline__ LdsTabs(uint32_t base, const MathTables* T, const MaskParams& main, bool pin)
        : cbrt{base + (uint32_t)offsetof(MathTables, cbrt), PowCoef<POW_CBRT>::c3},
          gamma{base + (uint32_t)offsetof(MathTables, gamma), PowCoef<POW_GAMMA>::c3}, sc(T->sc),
          k_gamma_lin(12.92f), k_gamma_off(0.055f), main_sat(main.sat_gain), main_light(main.light_gain)
    {
        if (pin)
            asm volatile("" : "+v"(cbrt.c3), "+v"(gamma.c3), "+v"(k_gamma_lin), "+v"(k_gamma_off), "+v"(main_sat), "+v"(main_light));
    }
};
__device__ __forceinline__ float chain_k_gamma_lin(const LdsTabs& T) { return T.k_gamma_lin; }
__device__ __forceinline__ float chain_k_gamma_off(const LdsTabs& T) { return T.k_gamma_off; }
__device__ __forceinline__ float chain_main_sat(const LdsTabs& T, const MaskParams&) { return T.main_sat; }
__device__ __forceinline__ float chain_main_light(const LdsTabs& T, const MaskParams&) { return T.main_light; }
// RPF_PRECISION_FAST: the binary32 tables of rpf_fast.h staged at LDS address `base` (in place of the MathTables), the same
// launch-long constants, and the gamma's 1.055 (its affine step is one fma with the clamp modifier there: no literal)
struct LdsFastTabs {
    LdsFastPow cbrt, gamma;
    float k_gamma_lin, k_gamma_off, k_gamma_mul, main_sat, main_light;
    // biased: `base` is FAST_TABS_BIASED_AT (k_hp's one-workgroup-per-CU launches without a curve in LDS)
    __device__ __forceinline__ LdsFastTabs(uint32_t base, const MaskParams& main, bool biased = false)
        : cbrt(biased ? fast_pow_biased<POW_CBRT>(base + (uint32_t)offsetof(FastTables, cbrt), FastPowCoef<POW_CBRT>::c3)
                      : fast_pow_plain<POW_CBRT>(base + (uint32_t)offsetof(FastTables, cbrt), FastPowCoef<POW_CBRT>::c3)),
          gamma(biased ? fast_pow_biased<POW_GAMMA>(base + (uint32_t)offsetof(FastTables, gamma), FastPowCoef<POW_GAMMA>::c3)
                       : fast_pow_plain<POW_GAMMA>(base + (uint32_t)offsetof(FastTables, gamma), FastPowCoef<POW_GAMMA>::c3)),
          k_gamma_lin(12.92f), k_gamma_off(0.055f), k_gamma_mul(1.055f), main_sat(main.sat_gain), main_light(main.light_gain)
    {
        asm volatile("" : "+v"(cbrt.c3), "+v"(gamma.c3), "+v"(k_gamma_lin), "+v"(k_gamma_off), "+v"(k_gamma_mul), "+v"(main_sat), "+v"(main_light));
    }
};
__device__ __forceinline__ float chain_k_gamma_lin(const LdsFastTabs& T) { return T.k_gamma_lin; }
__device__ __forceinline__ float chain_k_gamma_off(const LdsFastTabs& T) { return T.k_gamma_off; }
__device__ __forceinline__ float chain_k_gamma_mul(const LdsFastTabs& T) { return T.k_gamma_mul; }
__device__ __forceinline__ float chain_main_sat(const LdsFastTabs& T, const MaskParams&) { return T.main_sat; }
__device__ __forceinline__ float chain_main_light(const LdsFastTabs& T, const MaskParams&) { return T.main_light; }
// Where a launch with LDS to spare stages the FastTables so that both tables lie at or above their first row's shifted bit
// pattern (rpf_fast.h fast_pow_biased): the cube-root table's first row is pattern (127 - 22) * 32 * 16 = 53760, the gamma
// table then starts at 53760 + 11776 = 65536 >= (127 - 9) * 32 * 16 = 60416.  70 KiB of LDS, of which the first 52.5 are unused.
constexpr uint32_t FAST_TABS_BIASED_AT = (uint32_t)(127 + FastPowGeom<POW_CBRT>::E0) * FAST_POW_J * 16u;
static_assert(FAST_TABS_BIASED_AT + offsetof(FastTables, gamma) >= (uint32_t)(127 + FastPowGeom<POW_GAMMA>::E0) * FAST_POW_J * 16u &&
              FAST_TABS_BIASED_AT + offsetof(FastTables, gamma) - (uint32_t)(127 + FastPowGeom<POW_GAMMA>::E0) * FAST_POW_J * 16u < 65536u,
              "the gamma table's bias must fit the read's 16-bit immediate offset");
template <int PREC> struct TabsOf { typedef LdsTabs type; static constexpr uint32_t BYTES = (uint32_t)sizeof(MathTables); };
template <> struct TabsOf<1> { typedef LdsFastTabs type; static constexpr uint32_t BYTES = (uint32_t)sizeof(FastTables); };
// the tables staged at the start of the dynamic shared block (LDS address 0)
template <int PREC, bool PIN = true, bool BIASED = false>
__device__ __forceinline__ typename TabsOf<PREC>::type make_lds_tabs(const unsigned char* smem, const MaskParams& main)
{
    if constexpr (PREC == 1) return LdsFastTabs(BIASED ? FAST_TABS_BIASED_AT : 0u, main, BIASED);
    else return LdsTabs(0u, reinterpret_cast<const MathTables*>(smem), main, PIN);
}
// Mask parameters as the multi-mask kernels read them: "main" through the kernel-argument segment (loaded once, scalar
// registers), a user mask's 48 bytes from a copy in LDS at byte address `lds_base` with three 16-byte reads per mask per
// step — wave-uniform values in vector registers, so the tone arithmetic also runs at the full f32 rate (an SGPR operand
// halves it) — and its flag word made scalar again for the stage's uniform branches.
struct LdsMaskParams {
    const MaskParams* karg;
    uint32_t lds_base;
    const MaskParams* main = nullptr;  // k_hp: "main"'s parameters with the floats in vector registers (pin_main_params)
};
__device__ __forceinline__ const MaskParams& mask_params_main(const LdsMaskParams& s) { return s.main ? *s.main : s.karg[0]; }
// "main" alone (the single-mask kernels of k_hp): the same pinned copy
struct PinnedMain {
    const MaskParams* main;
};
__device__ __forceinline__ const MaskParams& mask_params_main(const PinnedMain& s) { return *s.main; }
__device__ __forceinline__ const MaskParams& mask_params_user(const PinnedMain& s, uint32_t) { return *s.main; }  // never called: num_masks == 1
// "main"'s tone / white-balance factors feed up to 33 f32 multiplications and additions per pixel in the run-time-flag
// kernels, and an SGPR operand halves the f32 rate: nine vector registers for the launch (the flag word stays scalar:
// it steers wave-uniform branches).  The pristine kernels never read them and the registers are dead there.
template <bool ALL>
__device__ __forceinline__ MaskParams pin_main_params(const MaskParams& k)
{
    MaskParams P = k;
    // the white-balance gains and the exposure factor: twelve multiplications per pixel
    asm volatile("" : "+v"(P.r_gain), "+v"(P.g_gain), "+v"(P.b_gain), "+v"(P.exposure_mul));
    // the rest (seven operations per pixel) where the registers are there: with the vignette stage's the 128 of four
    // waves per SIMD are used up, and a spill inside the loop costs more than the half-rate operands
    if (ALL) asm volatile("" : "+v"(P.shadow), "+v"(P.highlight), "+v"(P.black), "+v"(P.white), "+v"(P.contrast_c));
    return P;
}
__device__ __forceinline__ MaskParams mask_params_user(const LdsMaskParams& s, uint32_t m)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    static_assert(sizeof(MaskParams) == 48, "three 16-byte reads");
    const uint32_t a = s.lds_base + m * 48u;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(3))) v4u* lds_v4u_ptr;
    const v4u q0 = *(lds_v4u_ptr)(a), q1 = *(lds_v4u_ptr)(a + 16u), q2 = *(lds_v4u_ptr)(a + 32u);
#else  // host pass of the HIP compilation (never called; an LDS address is not a host pointer)
    const v4u q0 = {a, 0, 0, 0}, q1 = q0, q2 = q0;
#endif
    MaskParams P;
    P.r_gain = bits2f(q0.x); P.g_gain = bits2f(q0.y); P.b_gain = bits2f(q0.z); P.exposure_mul = bits2f(q0.w);
    P.shadow = bits2f(q1.x); P.highlight = bits2f(q1.y); P.black = bits2f(q1.z); P.white = bits2f(q1.w);
    P.contrast_c = bits2f(q2.x); P.flags = __builtin_amdgcn_readfirstlane(q2.y);
    P.sat_gain = bits2f(q2.z); P.light_gain = bits2f(q2.w);
    return P;
}
__device__ __forceinline__ void stage_mask_params(const DeviceEdit& ed, unsigned char* lds, int tid, int nt)
{
    const uint32_t* src = reinterpret_cast<const uint32_t*>(ed.mp);
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds);
    for (int i = tid; i < (int)(sizeof(MaskParams) * RPF_MASKS_PER_WORD / 4); i += nt) dst[i] = src[i];
}

// The dynamic shared block must start at LDS address 0 for LdsTabs' constant addresses to be right (it does: these
// kernels declare no static __shared__ variable).  One scalar compare per launch; a mismatch is a build error to be
// seen, not computed through.
__device__ __forceinline__ void assert_lds_base(const unsigned char* smem)
{
    if ((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem != 0u) __builtin_trap();
}
enum { OUT_RGBA_F32_PLAIN = 3 };  // internal (A/B only): RGBA32F with ordinary stores

// ---------------------------------------------------------------------------
// curve access (the Luts policy of edit_chain)
// ---------------------------------------------------------------------------
template <bool LDS_LUT>
struct DevLuts {
    const uint16_t* lds_bright;  // one mask's brightness curve in LDS (LDS_LUT): mask `lds_m`'s
    const uint16_t* g_bright;
    const ushort4* g_hsl;
    const HueBound* g_hue_bounds;
    const HueBoundCs* g_hue_cs;
    const HueFused* g_hue_fused;
    uint32_t lds_m = 0u;  // 0 ("main") in every single-mask kernel — a literal there, so the test below folds away
    __device__ __forceinline__ const HueBound* hue_bounds() const { return g_hue_bounds; }
    __device__ __forceinline__ const HueBoundCs* hue_bounds_cs() const { return g_hue_cs; }
    __device__ __forceinline__ const HueFused* hue_fused() const { return g_hue_fused; }
    __device__ __forceinline__ uint32_t brightness(uint32_t m, uint32_t i) const
    {
        if (LDS_LUT && m == lds_m) return lds_bright[i];
        return g_bright[(size_t)m * RPF_CURVE_RESOLUTION + i];
    }
    __device__ __forceinline__ void hsl(uint32_t m, uint32_t i, const MaskParams& P, uint32_t& nh, float& sg,
                                        float& lg) const
    {
        const ushort4 v = g_hsl[(size_t)m * RPF_CURVE_RESOLUTION + i];
        nh = v.x;
        if (!(P.flags & MF_SAT_CONST)) sg = div_u16_by(v.y, 32767.5f, (float)(1.0 / 32767.5));
        if (!(P.flags & MF_LIGHT_CONST)) lg = div_u16_by(v.z, 32767.5f, (float)(1.0 / 32767.5));
    }
};

// ---------------------------------------------------------------------------
// Issue priority that falls with a wave's progress through its share of the launch.
//
// The SIMD's arbiter serves its resident waves by priority, then AGE: at equal priority the oldest wave runs nearly
// unimpeded and the youngest gets what is left.  In a persistent kernel whose waves start together with equal shares that
// makes them finish one after the other, and the SIMD ends the launch with three, two, one wave — which cannot fill its
// issue slots.  That tail was the "fixed ~10 us" of a fused launch (tools/band_sweep.py) and 5 % of a 6000x4000 frame.
// Here a wave lowers its priority as it advances, in eighths of its share (3, 2, 1, 0, 3, 2, 1, 0: a wave that is ahead by
// less than half of its share yields to the ones behind, except for the eighth after the wrap), so the waves of a SIMD
// stay together and finish together.  Measured: set D 0.1672 -> 0.1594 ms, set S 0.2070 -> 0.1950 (profiles/r03_experiments.md).
// All state is wave-uniform (scalar registers); s_setprio takes an immediate, hence the switch.
// ---------------------------------------------------------------------------
struct ProgressPrio {
    uint32_t total, acc, k;
    __device__ __forceinline__ explicit ProgressPrio(uint32_t total_steps) : total(total_steps ? total_steps : 1u), acc(0), k(0)
    {
        __builtin_amdgcn_s_setprio(3);
    }
    __device__ __forceinline__ void step_done()  // k = floor(8 * steps done / total) without a division
    {
        acc += 8u;
        if (acc < total) return;
        do { acc -= total; ++k; } while (acc >= total);
        switch (k & 3u) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
        }
    }
};

// ---------------------------------------------------------------------------
// DPP wavefront shifts (gfx9 family: wave_shr:1 = 0x138, wave_shl:1 = 0x130)
// ---------------------------------------------------------------------------
// lane L receives v of lane L-1; lane 0 keeps `edge`
__device__ __forceinline__ float from_left(float v, float edge)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
// the same shifts with zero fill (bound_ctrl): lane 0 / lane 63 receive 0.  No `old` value to preserve, so the
// compiler does not have to copy one into the destination first.
__device__ __forceinline__ float from_left0(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right0(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// lane L receives v of lane L+1; lane 63 keeps `edge`
__device__ __forceinline__ float from_right(float v, float edge)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

template <int OUTFMT>
__device__ __forceinline__ void store_px(void* out, size_t pix, float r, float g, float b)
{
    if (OUTFMT == OUT_RGBA_F32) {
        // nontemporal: the frame is written once and never re-read by the kernel; measured on MI355X the
        // demosaic kernel goes from 4.2 to 5.7 TB/s with it (profiles/r01_experiments.md)
        typedef float v4f __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(v4f{r, g, b, 1.0f}, reinterpret_cast<v4f*>(out) + pix);
    } else if (OUTFMT == OUT_RGBA_F32_PLAIN) {
        reinterpret_cast<float4*>(out)[pix] = make_float4(r, g, b, 1.0f);
    } else if (OUTFMT == OUT_RGB_F32) {  // image.rs:202-277
        float* o = reinterpret_cast<float*>(out) + pix * 3;
        o[0] = r; o[1] = g; o[2] = b;
    } else {  // image.rs:375-383: (v.clamp(0,1) * 255.0) as u8
        uint8_t* o = reinterpret_cast<uint8_t*>(out) + pix * 3;
        o[0] = (uint8_t)(clamp01(r) * 255.0f);
        o[1] = (uint8_t)(clamp01(g) * 255.0f);
        o[2] = (uint8_t)(clamp01(b) * 255.0f);
    }
}

// gfx9 raw buffer descriptor, dword 3 (DATA_FORMAT = 32-bit): offsets past num_records are dropped / read as 0
constexpr int RPF_BUFFER_RSRC_FLAGS = 0x00020000;

// The pixel at column `col` of the output row whose first pixel is `row_pix` (wave-uniform), through a buffer
// descriptor of THAT ROW built in scalar registers (base = the row's address, num_records = the row's bytes): the
// lane contributes a 32-bit byte offset and nothing else — `buffer_store_dwordx4 v_data, v_off, s[rsrc], 0 offen nt`,
// no 64-bit vector address arithmetic (through store_px the compiler spends a v_mad_i64_i32 and a v_lshl_add_u64,
// both half-rate, on every pixel) — and a lane whose column lies beyond the row (a ragged last strip) is dropped by
// the hardware's range check instead of by an exec-mask dance around the store.
template <int OUTFMT>
constexpr uint32_t out_bpp() { return (OUTFMT == OUT_RGBA_F32 || OUTFMT == OUT_RGBA_F32_PLAIN) ? 16u : (OUTFMT == OUT_RGB_F32 ? 12u : 3u); }

// one pixel at byte offset `off` of the buffer `rs` (offsets at or past its num_records are dropped by the hardware)
template <int OUTFMT>
__device__ __forceinline__ void store_px_buf(__amdgpu_buffer_rsrc_t rs, uint32_t off, float r, float g, float b,
                                             uint32_t alpha = 0x3f800000u)  // the bits of 1.0f (a register of the caller's, or the literal)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef uint32_t v3u __attribute__((ext_vector_type(3)));
    if (OUTFMT == OUT_RGBA_F32) {
        // nontemporal (aux bit 1): the frame is written once and never re-read by the kernel; measured on MI355X the
        // demosaic kernel goes from 4.2 to 5.7 TB/s with it (profiles/r01_experiments.md)
        __builtin_amdgcn_raw_buffer_store_b128(v4u{f2bits(r), f2bits(g), f2bits(b), alpha}, rs, off, 0, 2);
    } else if (OUTFMT == OUT_RGBA_F32_PLAIN) {
        __builtin_amdgcn_raw_buffer_store_b128(v4u{f2bits(r), f2bits(g), f2bits(b), 0x3f800000u}, rs, off, 0, 0);
    } else if (OUTFMT == OUT_RGB_F32) {  // image.rs:202-277
        __builtin_amdgcn_raw_buffer_store_b96(v3u{f2bits(r), f2bits(g), f2bits(b)}, rs, off, 0, 2);  // nontemporal, like RGBA
    } else {  // image.rs:375-383: (v.clamp(0,1) * 255.0) as u8
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(clamp01(r) * 255.0f), rs, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(clamp01(g) * 255.0f), rs, off + 1u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(clamp01(b) * 255.0f), rs, off + 2u, 0, 0);
    }
}

template <int OUTFMT>
__device__ __forceinline__ void store_px_row(void* out, size_t row_pix, uint32_t width, uint32_t col, float r, float g, float b)
{
    constexpr uint32_t BPP = out_bpp<OUTFMT>();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char*>(out) + row_pix * BPP, 0, (int)(width * BPP), RPF_BUFFER_RSRC_FLAGS);
    store_px_buf<OUTFMT>(rs, col * BPP, r, g, b);
}

__device__ __forceinline__ uint32_t active_masks(const DeviceEdit& ed, size_t pix, size_t plane)
{
    (void)plane;
    // bit 0: "main" is all ones by construction (lib.rs:193-196); bits 1..: the user masks' combined bit plane
    return 1u | ((uint32_t)ed.mask_bits[pix] << 1);
}

// N16 16-byte words from global memory to LDS by the NT threads of a workgroup, up to eight loads per thread in flight
// before the first is waited for: written as `for (i = tid; i < n; i += nt) dst[i] = src[i]` the compiler emits a loop
// with a full wait per trip — one L2 round trip (~1 us) per trip, at the start of every launch.
template <int NT, int N16>
__device__ __forceinline__ void stage16(const uint4* __restrict__ src, uint4* __restrict__ dst, int tid)
{
    constexpr int CH = 8;
#pragma unroll
    for (int base = 0; base < N16; base += NT * CH) {
        uint4 v[CH];
        int idx[CH];
        // a thread past the end of a partial last chunk copies the last word again (same value to the same address):
        // no branch, so nothing keeps the loads from being issued together
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int i = base + k * NT + tid;
            idx[k] = (base + (k + 1) * NT <= N16 || i < N16) ? i : N16 - 1;
            if (base + k * NT < N16) v[k] = src[idx[k]];
        }
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (base + k * NT < N16) dst[idx[k]] = v[k];
    }
}

template <int NT, bool WITH_LUT>
__device__ __forceinline__ void stage_tables(const DeviceEdit& ed, MathTables* T, uint16_t* lut, int tid)
{
    stage16<NT, (int)(sizeof(MathTables) / 16)>(reinterpret_cast<const uint4*>(ed.tabs), reinterpret_cast<uint4*>(T), tid);
    if (WITH_LUT)
        stage16<NT, RPF_CURVE_RESOLUTION * 2 / 16>(reinterpret_cast<const uint4*>(ed.bright + (size_t)ed.lds_bright_mask * RPF_CURVE_RESOLUTION),
                                                   reinterpret_cast<uint4*>(lut), tid);
}

// The edit chain on a pixel pair in the depth of specialisation MODE names (see the enum above).
template <int MODE, class Luts>
__device__ __forceinline__ void run_chain(float (&r)[2], float (&gg)[2], float (&b)[2], const uint32_t (&active)[2],
                                          const DeviceEdit& ed, const Luts& luts, const LdsTabs& T, uint32_t lds_mp,
                                          const MaskParams* mainv = nullptr)  // pin_main_params' copy (k_hp)
{
    if (MODE == MODE_DEVELOP_1P) {
        edit_chain<2, 1, (int)MF_PRISTINE>(r, gg, b, active, ed.mp, 1u, luts, T);
    } else if (MODE == MODE_DEVELOP_1H) {
        if (mainv) edit_chain<2, 1>(r, gg, b, active, PinnedMain{mainv}, 1u, luts, T);
        else edit_chain<2, 1>(r, gg, b, active, ed.mp, 1u, luts, T);
    } else if (MODE == MODE_DEVELOP_1) {  // one mask, non-identity hue curve: its curves are folded into the hue table
        if (mainv) edit_chain<2, 2>(r, gg, b, active, PinnedMain{mainv}, 1u, luts, T);
        else edit_chain<2, 2>(r, gg, b, active, ed.mp, 1u, luts, T);
    } else if (MODE == MODE_DEVELOP_SEQ) {
        for (int k = 0; k < 2; ++k) {
            float r1[1] = {r[k]}, g1[1] = {gg[k]}, b1[1] = {b[k]};
            const uint32_t a1[1] = {active[k]};
            edit_chain<1>(r1, g1, b1, a1, ed.mp, ed.num_masks, luts, T);
            r[k] = r1[0]; gg[k] = g1[0]; b[k] = b1[0];
        }
    } else if (MODE == MODE_DEVELOP_MPH) {
        edit_chain<2, 1, (int)MF_PRISTINE>(r, gg, b, active, LdsMaskParams{ed.mp, lds_mp}, ed.num_masks, luts, T);
    } else if (MODE == MODE_DEVELOP_MP) {
        edit_chain<2, 0, (int)MF_PRISTINE>(r, gg, b, active, LdsMaskParams{ed.mp, lds_mp}, ed.num_masks, luts, T);
    } else if (lds_mp != 0xffffffffu) {
        edit_chain<2>(r, gg, b, active, LdsMaskParams{ed.mp, lds_mp, mainv}, ed.num_masks, luts, T);
    } else {
        edit_chain<2>(r, gg, b, active, ed.mp, ed.num_masks, luts, T);
    }
}

// RPF_PRECISION_FAST: the same depths of specialisation over edit_chain_fast (rpf_chain.h)
template <int MODE, class Luts>
__device__ __forceinline__ void run_chain(float (&r)[2], float (&gg)[2], float (&b)[2], const uint32_t (&active)[2],
                                          const DeviceEdit& ed, const Luts& luts, const LdsFastTabs& T, uint32_t lds_mp,
                                          const MaskParams* mainv = nullptr)
{
    // HUE_ROT (rpf_chain.h): every hue curve is the identity — modes _1P, _1H, _MPH
    if (MODE == MODE_DEVELOP_1P) {
        edit_chain_fast<2, (int)MF_PRISTINE, true>(r, gg, b, active, ed.mp, 1u, luts, T);
    } else if (MODE == MODE_DEVELOP_1H) {
        if (mainv) edit_chain_fast<2, -1, true>(r, gg, b, active, PinnedMain{mainv}, 1u, luts, T);
        else edit_chain_fast<2, -1, true>(r, gg, b, active, ed.mp, 1u, luts, T);
    } else if (mode_single(MODE)) {  // _1: a hue curve — through the editor's fused hue table (HUE_TAB)
        if (mainv) edit_chain_fast<2, -1, false, true>(r, gg, b, active, PinnedMain{mainv}, 1u, luts, T);
        else edit_chain_fast<2, -1, false, true>(r, gg, b, active, ed.mp, 1u, luts, T);
    } else if (MODE == MODE_DEVELOP_MPH) {
        edit_chain_fast<2, (int)MF_PRISTINE, true>(r, gg, b, active, LdsMaskParams{ed.mp, lds_mp}, ed.num_masks, luts, T);
    } else if (MODE == MODE_DEVELOP_MP) {
        edit_chain_fast<2, (int)MF_PRISTINE>(r, gg, b, active, LdsMaskParams{ed.mp, lds_mp}, ed.num_masks, luts, T);
    } else {
        edit_chain_fast<2>(r, gg, b, active, LdsMaskParams{ed.mp, lds_mp, mainv}, ed.num_masks, luts, T);
    }
}

// The vertical pixel pair at one column: demosaic both sites (one is R/B, the other G), camera matrix,
// then — MODE_DEVELOP — vignette and the edit chain on the pair in lockstep, and the two stores.
// c/s1/e1/d/v2/h2 [0] = top pixel, [1] = bottom pixel.  cx = RGGB cell column of this lane, cy_top of the top row.
// VIGN: 0 = the launch has no vignette (code and registers compiled out), 1 = decided at run time from the pixel
// position (generic), 2 = the caller hoisted the squared coordinates of the column and of the two rows.
template <int MODE, bool LDS_LUT, int OUTFMT, int VIGN = 1, class Tabs = LdsTabs>
__device__ __forceinline__ void finish_pair(const float (&c)[2], const float (&s1)[2], const float (&e1)[2],
                                            const float (&d)[2], const float (&v2)[2], const float (&h2)[2], int cx,
                                            int cy_top, const RawParamsDev& raw, const DeviceEdit& ed,
                                            const DevLuts<LDS_LUT>& luts, const Tabs& T, int gx, int ry_top,
                                            const TileGeom& g, bool col_live, void* out, const float* cam,
                                            float vcx2 = 0.0f, float vcy2_top = 0.0f, float vcy2_bot = 0.0f,
                                            uint32_t lds_mp = 0xffffffffu)  // LDS address of the staged mask parameters
{
    const bool topA = cx == cy_top;  // top pixel is the R/B site, bottom the G site
    const float cA = topA ? c[0] : c[1], s1A = topA ? s1[0] : s1[1], e1A = topA ? e1[0] : e1[1];
    const float dA = topA ? d[0] : d[1], v2A = topA ? v2[0] : v2[1], h2A = topA ? h2[0] : h2[1];
    const float cG = topA ? c[1] : c[0], s1G = topA ? s1[1] : s1[0], e1G = topA ? e1[1] : e1[0];
    const float dG = topA ? d[1] : d[0], v2G = topA ? v2[1] : v2[0], h2G = topA ? h2[1] : h2[0];

    const float cross = mhc_cross(cA, s1A, e1A, v2A, h2A);
    const float diag = mhc_diag(cA, dA, v2A, h2A);
    const float horiz = mhc_axis(cG, e1G, dG, h2G, v2G);
    const float vert = mhc_axis(cG, s1G, dG, v2G, h2G);

    // A site: cx == cy -> R site when cx == 0, B site when cx == 1; the G site's row parity is cx ^ 1
    const float RA = cx == 0 ? cA : diag, GA = cross, BA = cx == 0 ? diag : cA;
    const float RG = cx == 1 ? horiz : vert, GG = cG, BG = cx == 1 ? vert : horiz;

    float r[2], gg[2], b[2];
    cam_to_linear(cam, topA ? RA : RG, topA ? GA : GG, topA ? BA : BG, r[0], gg[0], b[0]);
    cam_to_linear(cam, topA ? RG : RA, topA ? GG : GA, topA ? BG : BA, r[1], gg[1], b[1]);

    const int W = (int)g.width, rows = (int)g.rows;
    bool live[2];
    size_t pix[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
        live[k] = col_live && ry_top + k < rows;
        if (!mode_single(MODE)) pix[k] = live[k] ? (size_t)(ry_top + k) * W + gx : 0;  // mask-plane index
    }
    if (MODE != MODE_DEMOSAIC) {
        if (VIGN == 2 || (VIGN == 1 && ed.vign_strength != 0.0f)) {  // wgsl:166-178, global coordinates
            for (int k = 0; k < 2; ++k) {
                const float gain = VIGN == 2
                                       ? vignette_from_axes(ed.vign_strength, vcx2, k ? vcy2_bot : vcy2_top)
                                       : vignette_gain(ed.vign_strength, gx, (int)g.y0 + ry_top + k, (float)W, (float)g.height);
                r[k] *= gain; gg[k] *= gain; b[k] *= gain;
            }
        }
        uint32_t active[2] = {1u, 1u};
        if (!mode_single(MODE) && ed.num_masks > 1)
            for (int k = 0; k < 2; ++k) active[k] = active_masks(ed, pix[k], (size_t)rows * W);
        run_chain<MODE>(r, gg, b, active, ed, luts, T, lds_mp);
    }
    // rows past a ragged band: a wave-uniform test; columns past a ragged strip: the row descriptor's range check
    for (int k = 0; k < 2; ++k)
        if (ry_top + k < rows)
            store_px_row<OUTFMT>(out, (size_t)(ry_top + k) * (size_t)W, (uint32_t)W, (uint32_t)gx, r[k], gg[k], b[k]);
}

// ---------------------------------------------------------------------------
// k_stream — rolling-window front end
// ---------------------------------------------------------------------------
// This file builds as one translation unit (RPF_TU undefined or 0: tests, tools, a plain `hipcc -c`) or as two
// (build.py): RPF_TU == 1, the rolling-window kernels and their launchers, compiled with the scheduler's max-ILP strategy
// (fused develop -1.9 %, the RGBA pass with all four curves +3.8 %: profiles/r02_experiments.md), and RPF_TU == 2,
// everything else with the default strategy.
#ifndef RPF_TU
#define RPF_TU 0
#endif

#if RPF_TU != 2
constexpr int STRIP_W = 64;  // one wavefront = 64 output columns: every store instruction is one aligned KiB

struct StreamArgs {
    const uint16_t* cfa;   // row 0 = frame row y0 - 2
    TileGeom g;
    void* out;
    RawParamsDev raw;
    DeviceEdit ed;
    uint32_t strips, seg_pairs, nitems;
    // Items [0, tail_start) are dealt whole.  The last, incomplete round — fewer items than waves — is cut into
    // single-pair sub-items: virtual item tail_start + t is pair (t % seg_pairs) of item tail_start + t / seg_pairs.
    uint32_t tail_start, nvirt;
    // PACKED input (rpf_dev_develop_packed): cfa points at bit-packed rows (any of the three orders below) instead of a u16 plane
    uint32_t bits, row_stride, row_bytes;  // sample width; bytes between rows; bytes of one row = ceil(W * bits / 8)
    uint32_t lsb;                          // 0 = MSB-first bit stream, 1 = LSB-first (little-endian), 2 = MIPI CSI-2 groups
};

// One CFA row of a strip as the wavefront sees it: v = this lane's sample; p1 / p2 = the W+E sums at
// distance 1 / 2 of the 5x5 stencil.  The four samples outside the wavefront's 64 columns are the halo
// h: lanes 0,1 hold columns x0-2, x0-1 and lanes 62,63 hold x0+64, x0+65 (other lanes: don't care).
struct RowTaps { float v, p1, p2; };
__device__ __forceinline__ RowTaps row_taps(float v, float h, bool need_p2)
{
    const float hl1 = from_right0(h);          // lane 0  <- lane 1's halo  (column x0-1); the other lanes: unused
    const float hr1 = from_left0(h);           // lane 63 <- lane 62's halo (column x0+64); the other lanes: unused
    const float l1 = from_left(v, hl1);        // x-1
    const float r1 = from_right(v, hr1);       // x+1
    RowTaps t;
    t.v = v;
    t.p1 = l1 + r1;
    t.p2 = need_p2 ? from_left(l1, h) + from_right(r1, h) : 0.0f;  // x-2 (lane 0: own halo x0-2), x+2 (lane 63: x0+65)
    return t;
}

typedef uint32_t u32_unaligned __attribute__((aligned(1)));  // one global_load_dword at any byte address

// 256-thread builds are launched five workgroups per CU: tell the register allocator (<= 96 VGPRs)
// PREC = 1: RPF_PRECISION_FAST (rpf_fast.h) — launches with user masks; the tables staged are the FastTables
template <int NT, int MODE, bool LDS_LUT, int OUTFMT, int PACKED = 0, bool VIGN = false, int PREC = 0>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 256 ? 5 : 1)))
void k_stream(const StreamArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t TAB_BYTES = TabsOf<PREC>::BYTES;
    MathTables* T = reinterpret_cast<MathTables*>(smem);
    uint16_t* lut = reinterpret_cast<uint16_t*>(smem + TAB_BYTES);
    const int tid = threadIdx.x;
    // the multi-mask kernels keep the masks' uniform parameters behind the tables (and the curve)
    constexpr uint32_t LDS_MP = TAB_BYTES + (LDS_LUT ? (uint32_t)RPF_CURVE_RESOLUTION * 2u : 0u);
    constexpr bool MULTI = mode_multi(MODE);
    if (MODE != MODE_DEMOSAIC) {
        assert_lds_base(smem);
        if (PREC) {
            stage16<NT, (int)(TAB_BYTES / 16)>(reinterpret_cast<const uint4*>(a.ed.fast_tabs), reinterpret_cast<uint4*>(smem), tid);
            if (LDS_LUT)
                stage16<NT, RPF_CURVE_RESOLUTION * 2 / 16>(reinterpret_cast<const uint4*>(a.ed.bright + (size_t)a.ed.lds_bright_mask * RPF_CURVE_RESOLUTION),
                                                           reinterpret_cast<uint4*>(lut), tid);
        } else {
            stage_tables<NT, LDS_LUT>(a.ed, T, lut, tid);
        }
        if (MULTI) stage_mask_params(a.ed, smem + LDS_MP, tid, NT);
        __syncthreads();  // the only barrier of the kernel
    }
    DevLuts<LDS_LUT> luts{lut, a.ed.bright, a.ed.hsl, a.ed.hue_bounds, a.ed.hue_cs, a.ed.hue_fused, MULTI ? a.ed.lds_bright_mask : 0u};
    const typename TabsOf<PREC>::type tabs = make_lds_tabs<PREC, false>(smem, a.ed.mp[0]);
    // the u16 band as one buffer: (rows + 4) rows of W samples (run_stream checks that it is shorter than 4 GiB)
    const __amdgpu_buffer_rsrc_t band_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.cfa), 0, (int)((a.g.rows + 4u) * a.g.width * 2u), RPF_BUFFER_RSRC_FLAGS);

    const int lane = tid & 63;
    const int W = (int)a.g.width, H = (int)a.g.height;
    const int y0 = (int)a.g.y0, rows = (int)a.g.rows;
    const int total_pairs = (rows + 1) / 2;
    const int bp = (y0 + a.raw.yo) & 1;  // RGGB cell row of window positions 0, 2, 4
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (NT / 64) + (tid >> 6));
    const uint32_t nwaves = gridDim.x * (NT / 64);

    // A strip starts on a multiple of 64 columns, so this lane's RGGB cell column — and with it the
    // normalisation constants of its column for even / odd window positions — is the same for every item.
    const int cx = (lane + a.raw.xo) & 1;
    const int cyE = bp, cyO = bp ^ 1;
    const float blkE = cyE ? (cx ? a.raw.black[3] : a.raw.black[2]) : (cx ? a.raw.black[1] : a.raw.black[0]);
    const float rngE = cyE ? (cx ? a.raw.range[3] : a.raw.range[2]) : (cx ? a.raw.range[1] : a.raw.range[0]);
    const float sclE = cyE ? (cx ? a.raw.scale[3] : a.raw.scale[2]) : (cx ? a.raw.scale[1] : a.raw.scale[0]);
    const float blkO = cyO ? (cx ? a.raw.black[3] : a.raw.black[2]) : (cx ? a.raw.black[1] : a.raw.black[0]);
    const float rngO = cyO ? (cx ? a.raw.range[3] : a.raw.range[2]) : (cx ? a.raw.range[1] : a.raw.range[0]);
    const float sclO = cyO ? (cx ? a.raw.scale[3] : a.raw.scale[2]) : (cx ? a.raw.scale[1] : a.raw.scale[0]);

    // The camera matrix in vector registers: an f32 operation with an SGPR operand issues at half rate on gfx950
    // (tools/microbench.hip: 1.73 vs 1.09 ns), and these nine values feed 18 multiplications per pixel pair.
    float camv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        camv[i] = a.raw.cam[i];
        asm volatile("" : "+v"(camv[i]));
    }

    // Every item (64 columns x 2*seg_pairs rows) costs the same number of instructions whatever the
    // image holds, so the items are dealt round-robin: wave-uniform loop, no queue.
    // (this wave's share in steps, for the progress priority: whole items seg_pairs each, tail sub-items one)
    ProgressPrio prio(MODE == MODE_DEMOSAIC || wave >= a.nvirt ? 0u
                      : (wave < a.tail_start ? (a.tail_start - wave + nwaves - 1) / nwaves * a.seg_pairs : 0u) +
                        (a.nvirt > a.tail_start ? (a.nvirt - max(wave, a.tail_start) + nwaves - 1) / nwaves : 0u));
    for (uint32_t vitem = wave; vitem < a.nvirt; vitem += nwaves) {
        uint32_t item = vitem, sub = 0, sub_pairs = a.seg_pairs;
        if (vitem >= a.tail_start) {
            const uint32_t t = vitem - a.tail_start;
            item = a.tail_start + t / a.seg_pairs;
            sub = t - (item - a.tail_start) * a.seg_pairs;
            sub_pairs = 1;
        }
        const uint32_t seg = item / a.strips, strip = item - seg * a.strips;
        const int x0 = (int)strip * STRIP_W;
        const int pair0 = (int)(seg * a.seg_pairs + sub);
        const int npairs = min((int)sub_pairs, total_pairs - pair0);
        if (npairs <= 0) continue;  // sub-item past the end of a ragged last segment
        const int xl = x0 + lane;  // this lane's column
        int gx = reflect101(xl, W);
        gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);  // columns past a ragged strip: unused
        // halo column of the edge lanes; same RGGB column parity as the lane's own column
        int hx = lane < 2 ? xl - 2 : (lane >= 62 ? xl + 2 : xl);
        hx = reflect101(hx, W);
        hx = hx < 0 ? 0 : (hx > W - 1 ? W - 1 : hx);
        const uint32_t goff = (uint32_t)gx * 2u, hoff = (uint32_t)hx * 2u;  // byte offsets inside a u16 row
        // Row addressing: lane j works out, once per item, the element offset of window row j of this
        // item (reflect-101 at the frame edge, clamp inside the band buffer); the loop then fetches the two
        // offsets it needs per step with v_readlane instead of redoing ~20 scalar operations per row on
        // uniforms that do not fit the SGPR file.  An item spans 2 * seg_pairs + 4 <= 64 rows.
        const int rbase = 2 * pair0 - 2;
        uint32_t rowoff;
        {
            int gy = reflect101(y0 + rbase + lane, H);
            int br = gy - (y0 - 2);
            br = br < 0 ? 0 : (br > rows + 3 ? rows + 3 : br);  // rows past a ragged band: unused
            rowoff = (uint32_t)br * (PACKED ? a.row_stride : (uint32_t)W * 2u);  // bytes
        }
        // PACKED: the integer unpack of rpf_dev_unpack_bayer folded into the row loads.  A lane's column fixes the
        // byte address and the shift of its sample inside a 32-bit big-endian window once per item; the window is
        // pulled back at the end of a row so that it never leaves the row (it still holds the whole sample).
        uint32_t cbyte_g = 0, cshift_g = 0, cbyte_h = 0, cshift_h = 0;
        const uint32_t smask = PACKED ? (1u << a.bits) - 1u : 0xffffu;
        if (PACKED) {
            auto col_addr = [&](int x, uint32_t& byte, uint32_t& shift) {
                if (PACKED == 2) {  // MIPI CSI-2: an 8-byte window that holds the column's whole group (5 / 3 / 7 bytes)
                    const uint32_t n = a.bits == 12u ? 2u : 4u, low = a.bits - 8u, gb = n + n * low / 8u;
                    const uint32_t grp = (uint32_t)x / n, i = (uint32_t)x - grp * n;
                    const uint32_t last = a.row_bytes > 8u ? a.row_bytes - 8u : 0u;  // row_bytes = whole groups
                    const uint32_t b = grp * gb > last ? last : grp * gb;
                    const uint32_t d = grp * gb - b;
                    byte = b;
                    shift = (8u * (d + i)) | ((8u * (d + n) + i * low) << 8);  // of the high byte | of the low bits
                    return;
                }
                const uint32_t bit = (uint32_t)x * a.bits;
                uint32_t b = bit >> 3;
                const uint32_t last = a.row_bytes - 4u;
                b = b > last ? last : b;
                byte = b;
                shift = a.lsb ? bit - 8u * b : 32u - a.bits - (bit - 8u * b);
            };
            col_addr(gx, cbyte_g, cshift_g);
            col_addr(hx, cbyte_h, cshift_h);
        }
        auto sample = [&](int j, bool halo) -> uint32_t {  // j = window row of the item, wave-uniform
            const uint32_t ro = __builtin_amdgcn_readlane(rowoff, j);
            if (PACKED) {
                const uint8_t* p = reinterpret_cast<const uint8_t*>(a.cfa) + ro + (halo ? cbyte_h : cbyte_g);
                if (PACKED == 2) {
                    const uint64_t w = ((uint64_t)*reinterpret_cast<const u32_unaligned*>(p + 4) << 32) |
                                       *reinterpret_cast<const u32_unaligned*>(p);
                    const uint32_t sh = halo ? cshift_h : cshift_g, low = a.bits - 8u;
                    const uint32_t hi = (uint32_t)(w >> (sh & 0xffu)) & 0xffu;
                    const uint32_t lo = (uint32_t)(w >> (sh >> 8)) & ((1u << low) - 1u);
                    return (hi << low) | lo;
                }
                const uint32_t raw32 = *reinterpret_cast<const u32_unaligned*>(p);
                const uint32_t w = a.lsb ? raw32 : __builtin_bswap32(raw32);
                return (w >> (halo ? cshift_h : cshift_g)) & smask;
            }
            // one descriptor of the whole band, the row's byte offset in a scalar register, the lane's column as a
            // 32-bit byte offset: `buffer_load_ushort v, v_off, s[rsrc], s_row offen` — no 64-bit vector address
            return __builtin_amdgcn_raw_buffer_load_b16(band_rsrc, halo ? hoff : goff, ro, 0);
        };
        // Vignette (wgsl:166-178): the column's squared coordinate once per item; lane j holds the squared
        // coordinate of the item's output row j (two IEEE divisions per lane per item instead of two per pixel)
        float vcx2 = 0.0f, vcy2_lane = 0.0f;
        if (VIGN) {
            vcx2 = vignette_axis2(xl, (float)W);
            vcy2_lane = vignette_axis2(y0 + 2 * pair0 + lane, (float)H);
        }
        // clamp(v, 0, range) as one v_med3_f32 (v is finite, range > 0: the same value as min(max(v, 0), range))
        auto normE = [&](uint32_t raw) { return __builtin_amdgcn_fmed3f((float)raw - blkE, 0.0f, rngE) * sclE; };
        auto normO = [&](uint32_t raw) { return __builtin_amdgcn_fmed3f((float)raw - blkO, 0.0f, rngO) * sclO; };

        // window position i <-> band row 2*pair - 2 + i
        const float v0 = normE(sample(0, false));
        RowTaps t1 = row_taps(normO(sample(1, false)), normO(sample(1, true)), false);
        RowTaps t2 = row_taps(normE(sample(2, false)), normE(sample(2, true)), true);
        RowTaps t3 = row_taps(normO(sample(3, false)), normO(sample(3, true)), true);
        uint32_t n4 = sample(4, false), n4h = sample(4, true), n5 = sample(5, false), n5h = sample(5, true);
        float w0 = v0;

        const bool col_live = xl < W;
        for (int p = 0; p < npairs; ++p) {  // wave-uniform
            const RowTaps t4 = row_taps(normE(n4), normE(n4h), true);
            const RowTaps t5 = row_taps(normO(n5), normO(n5h), true);
            if (p + 1 < npairs) {  // prefetch the next step's two rows
                n4 = sample(2 * p + 6, false); n4h = sample(2 * p + 6, true);
                n5 = sample(2 * p + 7, false); n5h = sample(2 * p + 7, true);
            }
            // neighbourhood sums: top pixel = window position 2, bottom = 3
            const float c[2] = {t2.v, t3.v};
            const float s1[2] = {t1.v + t3.v, t2.v + t4.v};
            const float e1[2] = {t2.p1, t3.p1};
            const float d[2] = {t1.p1 + t3.p1, t2.p1 + t4.p1};
            const float vv[2] = {w0 + t4.v, t1.v + t5.v};
            const float h2[2] = {t2.p2, t3.p2};
            float vcy2_top = 0.0f, vcy2_bot = 0.0f;
            if (VIGN) {  // wave-uniform; back into VGPRs (an SGPR operand halves the f32 rate)
                vcy2_top = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vcy2_lane), 2 * p));
                vcy2_bot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vcy2_lane), 2 * p + 1));
            }
            finish_pair<MODE, LDS_LUT, OUTFMT, VIGN ? 2 : 0>(c, s1, e1, d, vv, h2, cx, bp, a.raw, a.ed, luts, tabs, xl,
                                                             2 * (pair0 + p), a.g, col_live, a.out, camv, vcx2, vcy2_top,
                                                             vcy2_bot, MULTI ? LDS_MP : 0xffffffffu);
            // slide the window down two rows
            w0 = t2.v; t1 = t3; t2 = t4; t3 = t5;
            if (MODE != MODE_DEMOSAIC) prio.step_done();
        }
    }
}

// ---------------------------------------------------------------------------
// k_hp — horizontal-pair front end (round 3; the shipped fused kernel for u16 input)
//
// k_stream gives a lane ONE column and a vertical pixel pair, so which of the two pixels is the R/B site alternates from
// lane to lane and every step pays 22 v_cndmask (half-rate on gfx950) to sort the pair into roles and back, plus six DPP
// moves per row.  Here a wavefront owns a 128-column strip and lane l holds the two NEIGHBOURING columns 2l, 2l + 1 of it;
// the strip starts on an R column (x0 = 128 strip - xo) and a step on an R row (band rows 2p - bp, 2p - bp + 1), so the
// roles are the same in every lane and known at compile time: row 0 of a step is (R site, G), row 1 is (G, B site).
// No select anywhere in the front end; the horizontal taps of BOTH columns come from four DPP moves per row (the lane's
// own other column is a neighbour already); and the pair that goes through the edit chain in lockstep is the horizontal
// one.  What a lane then holds — pixels 2l and 2l + 1 of a row — would make every store instruction write every other
// 16 bytes of a line, which nontemporal stores take at a third of their rate (tools/store_probe.hip: 1.9 instead of
// 5.5 TB/s); one v_permlane32_swap per channel moves the second pixels of lanes 0..31 into lanes 32..63 of the first
// register (and vice versa), after which each of the two stores of a row covers one whole KiB (lanes 0..31 its even
// pixels, lanes 32..63 its odd ones: 5.4 TB/s in the same probe).  Same arithmetic, same bits as k_stream.
// The price: two columns' rolling windows in registers (about 110 VGPRs, four waves per SIMD instead of five).
// ---------------------------------------------------------------------------
constexpr int HP_STRIP_W = 128;
constexpr bool hp_fast_biased(int nt, bool lds_lut, int prec) { return prec == 1 && !lds_lut && nt == 1024; }

struct HpRow { float va, vb, p1a, p1b, p2a, p2b; };

// One CFA row of a strip, normalised: a / b = this lane's columns 2l, 2l + 1; ha / hb = the lane pair beyond the strip
// (lane 0: columns x0 - 2, x0 - 1; lane 63: x0 + 128, x0 + 129; other lanes: don't care).  p1 = (x-1) + (x+1),
// p2 = (x-2) + (x+2), left operand first as in row_taps.
__device__ __forceinline__ HpRow hp_row_taps(float a, float b, float ha, float hb)
{
    const float a_l = from_left(a, ha);   // a of the lane to the left: x-2 of a
    const float b_l = from_left(b, hb);   // b of the lane to the left: x-1 of a, x-2 of b
    const float a_r = from_right(a, ha);  // a of the lane to the right: x+2 of a, x+1 of b
    const float b_r = from_right(b, hb);  // x+2 of b
    HpRow t;
    t.va = a; t.vb = b;
    t.p1a = b_l + b;
    t.p1b = a + a_r;
    t.p2a = a_l + a_r;
    t.p2b = b_l + b_r;
    return t;
}

// The two pixels (columns 2l, 2l + 1 of the strip) of output row `row` through vignette, edit chain and the stores.
// colF: the column whose pixel this lane stores from the first register after the half-wave swap (second: + 64).
template <int MODE, bool LDS_LUT, int OUTFMT, bool VIGN, class Tabs>
__device__ __forceinline__ void hp_finish(float (&r)[2], float (&gg)[2], float (&b)[2], int row, const int (&col)[2],
                                          int colF, const TileGeom& g, const DeviceEdit& ed, const DevLuts<LDS_LUT>& luts,
                                          const Tabs& T, void* out, const float (&fall)[2], uint32_t lds_mp,
                                          const uint32_t (&alpha)[2], const MaskParams* mainv)
{
    const int W = (int)g.width, rows = (int)g.rows;
    const bool row_live = row >= 0 && row < rows;  // wave-uniform: false for the virtual row above / below a band
    if (VIGN) {  // wgsl:166-178; the falloff of the two pixels comes from the tile's map (DeviceEdit::vign_falloff)
        for (int k = 0; k < 2; ++k) {
            const float gain = vignette_gain_from_falloff(ed.vign_strength, fall[k]);
            r[k] *= gain; gg[k] *= gain; b[k] *= gain;
        }
    }
    uint32_t active[2] = {1u, 1u};
    if (!mode_single(MODE) && ed.num_masks > 1)
        for (int k = 0; k < 2; ++k)  // col[] is clamped into the frame: dead lanes read a valid entry
            active[k] = active_masks(ed, row_live ? (size_t)row * W + col[k] : 0, (size_t)rows * W);
    run_chain<MODE>(r, gg, b, active, ed, luts, T, lds_mp, mainv);
    // A row that is not stored gets an EMPTY descriptor (num_records 0: the hardware drops the stores) instead of a
    // branch around them: behind a branch the compiler also sinks half of the chain, and the join then needs
    // s_waitcnt vmcnt(0) — a drain of the row's own stores — because the chain's gathers are still in flight on the
    // path that skipped their uses.  (Such rows are the one virtual row above / below a band: their work is noise.)
    constexpr uint32_t BPP = out_bpp<OUTFMT>();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char*>(out) + (size_t)(row_live ? row : 0) * (size_t)W * BPP, 0,
        row_live ? (int)((uint32_t)W * BPP) : 0, RPF_BUFFER_RSRC_FLAGS);
    // v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second
    const auto sr = __builtin_amdgcn_permlane32_swap(f2bits(r[0]), f2bits(r[1]), false, false);
    const auto sg = __builtin_amdgcn_permlane32_swap(f2bits(gg[0]), f2bits(gg[1]), false, false);
    const auto sb = __builtin_amdgcn_permlane32_swap(f2bits(b[0]), f2bits(b[1]), false, false);
    // a column left of the frame (x0 = -1 for CFA layouts that start on a G column) or right of it wraps to / lies at an
    // offset beyond the row descriptor's num_records: dropped by the hardware's range check
    const uint32_t offF = (uint32_t)colF * BPP;
    // alpha: two launch-long registers that hold 1.0f, one per store — as a literal the compiler rebuilds the whole
    // 128-bit operand around it for every store (two v_mov_b64 and three v_mov_b32 each)
    store_px_buf<OUTFMT>(rs, offF, bits2f(sr[0]), bits2f(sg[0]), bits2f(sb[0]), alpha[0]);
    store_px_buf<OUTFMT>(rs, offF + 64u * BPP, bits2f(sr[1]), bits2f(sg[1]), bits2f(sb[1]), alpha[1]);
}

// PREC: 0 = every builtin correctly rounded (rpf_math.h; the default and the parity gate), 1 = RPF_PRECISION_FAST (rpf_fast.h)
template <int NT, int MODE, bool LDS_LUT, int OUTFMT, bool VIGN = false, int PREC = 0>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 256 ? (PREC ? 5 : 4) : 1)))
void k_hp(const StreamArgs a)
{
    static_assert(MODE != MODE_DEMOSAIC, "the demosaic-only kernel is store-bound: k_stream");
    extern __shared__ __align__(16) unsigned char smem[];
    // the binary32 tables at their biased address where the workgroup has the CU's LDS to itself and no curve to stage
    constexpr bool BIASED = hp_fast_biased(NT, LDS_LUT, PREC);
    constexpr uint32_t TAB_AT = BIASED ? FAST_TABS_BIASED_AT : 0u, TAB_BYTES = TAB_AT + TabsOf<PREC>::BYTES;
    uint16_t* lut = reinterpret_cast<uint16_t*>(smem + TAB_BYTES);
    const int tid = threadIdx.x;
    constexpr uint32_t LDS_MP = TAB_BYTES + (LDS_LUT ? (uint32_t)RPF_CURVE_RESOLUTION * 2u : 0u);
    constexpr bool MULTI = mode_multi(MODE);
    assert_lds_base(smem);
    stage16<NT, (int)(TabsOf<PREC>::BYTES / 16)>(PREC ? reinterpret_cast<const uint4*>(a.ed.fast_tabs) : reinterpret_cast<const uint4*>(a.ed.tabs),
                                                 reinterpret_cast<uint4*>(smem + TAB_AT), tid);
    if (LDS_LUT)
        stage16<NT, RPF_CURVE_RESOLUTION * 2 / 16>(reinterpret_cast<const uint4*>(a.ed.bright + (size_t)(MULTI ? a.ed.lds_bright_mask : 0u) * RPF_CURVE_RESOLUTION),
                                                   reinterpret_cast<uint4*>(lut), tid);
    if (MULTI) stage_mask_params(a.ed, smem + LDS_MP, tid, NT);
    __syncthreads();  // the only barrier of the kernel
    DevLuts<LDS_LUT> luts{lut, a.ed.bright, a.ed.hsl, a.ed.hue_bounds, a.ed.hue_cs, a.ed.hue_fused, MULTI ? a.ed.lds_bright_mask : 0u};
    const typename TabsOf<PREC>::type tabs = make_lds_tabs<PREC, true, BIASED>(smem, a.ed.mp[0]);
    const __amdgpu_buffer_rsrc_t band_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.cfa), 0, (int)((a.g.rows + 4u) * a.g.width * 2u), RPF_BUFFER_RSRC_FLAGS);
    // the tile's vignette falloff map: rows x W binary32 (run_hp checks that it exists and is shorter than 4 GiB)
    const __amdgpu_buffer_rsrc_t fall_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.ed.vign_falloff), 0, VIGN ? (int)(a.g.rows * a.g.width * 4u) : 0, RPF_BUFFER_RSRC_FLAGS);

    const int lane = tid & 63;
    const int W = (int)a.g.width, H = (int)a.g.height;
    const int y0 = (int)a.g.y0, rows = (int)a.g.rows;
    const int bp = (y0 + a.raw.yo) & 1;  // step p covers band rows 2p - bp (RGGB cell row 0) and 2p - bp + 1
    const int total_pairs = (rows + bp + 1) / 2;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (NT / 64) + (tid >> 6));
    const uint32_t nwaves = gridDim.x * (NT / 64);

    // Normalisation: n = med3(f32(raw) - black, 0, range) * scale per RGGB cell (cx + 2 cy).  Black and scale feed
    // full-rate f32 operations and therefore live in vector registers (an SGPR operand halves the rate); the range is
    // an operand of v_med3_f32, which is half-rate with any operand.
    float blk[4], scl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        blk[i] = a.raw.black[i]; scl[i] = a.raw.scale[i];
        asm volatile("" : "+v"(blk[i]), "+v"(scl[i]));
    }
    float camv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        camv[i] = a.raw.cam[i];
        asm volatile("" : "+v"(camv[i]));
    }
    auto norm = [&](uint32_t raw, int cell) {
        return __builtin_amdgcn_fmed3f((float)raw - blk[cell], 0.0f, a.raw.range[cell]) * scl[cell];
    };
    // pixel of the first 64 of a strip that this lane stores after the half-wave swap (hp_finish)
    const int pxF = lane < 32 ? 2 * lane : 2 * (lane - 32) + 1;
    uint32_t alpha[2] = {0x3f800000u, 0x3f800000u};
    asm volatile("" : "+v"(alpha[0]), "+v"(alpha[1]));
    const MaskParams mainv = pin_main_params<!VIGN>(a.ed.mp[0]);
    // this wave's share of the launch, in steps (items dealt round-robin; a tail sub-item counts as a whole item)
    ProgressPrio prio(wave < a.nvirt ? (a.nvirt - wave + nwaves - 1) / nwaves * a.seg_pairs : 0u);

    for (uint32_t vitem = wave; vitem < a.nvirt; vitem += nwaves) {
        uint32_t item = vitem, sub = 0, sub_pairs = a.seg_pairs;
        if (vitem >= a.tail_start) {
            const uint32_t t = vitem - a.tail_start;
            item = a.tail_start + t / a.seg_pairs;
            sub = t - (item - a.tail_start) * a.seg_pairs;
            sub_pairs = 1;
        }
        const uint32_t seg = item / a.strips, strip = item - seg * a.strips;
        const int x0 = (int)strip * HP_STRIP_W - a.raw.xo;  // an R column (of the frame, or the virtual column -1)
        const int pair0 = (int)(seg * a.seg_pairs + sub);
        const int npairs = min((int)sub_pairs, total_pairs - pair0);
        if (npairs <= 0) continue;
        const int xa = x0 + 2 * lane;
        auto column = [&](int x) {  // reflect-101 at the frame edge; columns far outside (a ragged strip): any valid one
            int c = reflect101(x, W);
            return c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
        };
        const int col[2] = {column(xa), column(xa + 1)};
        const int hxa = lane == 0 ? xa - 2 : (lane == 63 ? xa + 2 : xa);  // the lane pair beyond the strip
        const uint32_t goffa = (uint32_t)col[0] * 2u, goffb = (uint32_t)col[1] * 2u;  // byte offsets inside a u16 row
        const uint32_t hoffa = (uint32_t)column(hxa) * 2u, hoffb = (uint32_t)column(hxa + 1) * 2u;
        // Row addressing as in k_stream: lane j holds the byte offset of window row j of this item (band rows
        // 2 pair0 - bp - 2 + j: reflect-101 at the frame edge, clamped inside the band buffer — a clamped row only ever
        // feeds the virtual row above / below the band, which is not stored).  An item spans 2 seg_pairs + 4 <= 64 rows.
        const int rbase = 2 * pair0 - bp - 2;
        uint32_t rowoff;
        {
            int gy = reflect101(y0 + rbase + lane, H);
            int br = gy - (y0 - 2);
            br = br < 0 ? 0 : (br > rows + 3 ? rows + 3 : br);
            rowoff = (uint32_t)br * (uint32_t)W * 2u;
        }
        struct Raw4 { uint32_t a, b, ha, hb; };
        auto load_row = [&](int j) {  // j = window row of the item, wave-uniform
            const uint32_t ro = __builtin_amdgcn_readlane(rowoff, j);
            Raw4 q;
            q.a = __builtin_amdgcn_raw_buffer_load_b16(band_rsrc, goffa, ro, 0);
            q.b = __builtin_amdgcn_raw_buffer_load_b16(band_rsrc, goffb, ro, 0);
            q.ha = __builtin_amdgcn_raw_buffer_load_b16(band_rsrc, hoffa, ro, 0);
            q.hb = __builtin_amdgcn_raw_buffer_load_b16(band_rsrc, hoffb, ro, 0);
            return q;
        };
        // window position i <-> band row 2 pair - bp - 2 + i: RGGB cell row i & 1; a is cell column 0, b cell column 1
        auto taps = [&](const Raw4& q, int odd) {
            return hp_row_taps(norm(q.a, 2 * odd), norm(q.b, 2 * odd + 1), norm(q.ha, 2 * odd), norm(q.hb, 2 * odd + 1));
        };
        // Vignette: the falloff of this lane's two pixels of a row, from the tile's map (DeviceEdit::vign_falloff), loaded
        // one step ahead like the CFA rows.  A virtual row above / below the band reads the nearest real one (not stored).
        struct Fall { float a, b; };
        auto load_fall = [&](int row) {
            Fall f{0.0f, 0.0f};
            if (VIGN) {
                const int rc = row < 0 ? 0 : (row > rows - 1 ? rows - 1 : row);
                const uint32_t so = (uint32_t)rc * (uint32_t)W * 4u;
                f.a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fall_rsrc, goffa * 2u, so, 0));
                f.b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fall_rsrc, goffb * 2u, so, 0));
            }
            return f;
        };

        float w0a, w0b;
        {
            const Raw4 q0 = load_row(0);
            w0a = norm(q0.a, 0); w0b = norm(q0.b, 1);
        }
        HpRow t1 = taps(load_row(1), 1), t2 = taps(load_row(2), 0), t3 = taps(load_row(3), 1);
        Raw4 n4 = load_row(4), n5 = load_row(5);

        Fall nf0 = load_fall(2 * pair0 - bp), nf1 = load_fall(2 * pair0 - bp + 1);
        for (int p = 0; p < npairs; ++p) {  // wave-uniform
            const HpRow t4 = taps(n4, 0), t5 = taps(n5, 1);
            const int row = 2 * (pair0 + p) - bp;
            const float fall0[2] = {nf0.a, nf0.b}, fall1[2] = {nf1.a, nf1.b};
            if (p + 1 < npairs) {  // prefetch the next step's two rows (and their falloff)
                n4 = load_row(2 * p + 6);
                n5 = load_row(2 * p + 7);
                nf0 = load_fall(row + 2);
                nf1 = load_fall(row + 3);
            }
            float r[2], gg[2], b[2];
            {   // row 0 of the step (window position 2): a = R site, b = G site on the R row
                const float s1a = t1.va + t3.va, s1b = t1.vb + t3.vb;
                const float da = t1.p1a + t3.p1a, db = t1.p1b + t3.p1b;
                const float vva = w0a + t4.va, vvb = w0b + t4.vb;
                const float cross = mhc_cross(t2.va, s1a, t2.p1a, vva, t2.p2a);
                const float diag = mhc_diag(t2.va, da, vva, t2.p2a);
                const float horiz = mhc_axis(t2.vb, t2.p1b, db, t2.p2b, vvb);
                const float vert = mhc_axis(t2.vb, s1b, db, vvb, t2.p2b);
                cam_to_linear(camv, t2.va, cross, diag, r[0], gg[0], b[0]);
                cam_to_linear(camv, horiz, t2.vb, vert, r[1], gg[1], b[1]);
                hp_finish<MODE, LDS_LUT, OUTFMT, VIGN>(r, gg, b, row, col, x0 + pxF, a.g, a.ed, luts, tabs, a.out, fall0,
                                                       MULTI ? LDS_MP : 0xffffffffu, alpha, &mainv);
            }
            {   // row 1 (window position 3): a = G site on the B row, b = B site
                const float s1a = t2.va + t4.va, s1b = t2.vb + t4.vb;
                const float da = t2.p1a + t4.p1a, db = t2.p1b + t4.p1b;
                const float vva = t1.va + t5.va, vvb = t1.vb + t5.vb;
                const float vert = mhc_axis(t3.va, s1a, da, vva, t3.p2a);
                const float horiz = mhc_axis(t3.va, t3.p1a, da, t3.p2a, vva);
                const float cross = mhc_cross(t3.vb, s1b, t3.p1b, vvb, t3.p2b);
                const float diag = mhc_diag(t3.vb, db, vvb, t3.p2b);
                cam_to_linear(camv, vert, t3.va, horiz, r[0], gg[0], b[0]);
                cam_to_linear(camv, diag, cross, t3.vb, r[1], gg[1], b[1]);
                hp_finish<MODE, LDS_LUT, OUTFMT, VIGN>(r, gg, b, row + 1, col, x0 + pxF, a.g, a.ed, luts, tabs, a.out, fall1,
                                                       MULTI ? LDS_MP : 0xffffffffu, alpha, &mainv);
            }
            // slide the window down two rows
            w0a = t2.va; w0b = t2.vb;
            t1 = t3; t2 = t4; t3 = t5;
            prio.step_done();
        }
    }
}

// ---------------------------------------------------------------------------
// k_tile — LDS-tile front end (variant 1): persistent workgroups walk 128 x TILE_H tiles; the
// (TILE_H+4) x 132 normalised CFA neighbourhood is staged in LDS, lanes read their 6-row column and
// take the horizontal taps by DPP (lanes 0/63 patch the two samples beyond the wavefront from LDS).
// ---------------------------------------------------------------------------
constexpr int TILE_W = 128;
constexpr int PITCH = TILE_W + 4;

struct TileArgs {
    const uint16_t* cfa;
    TileGeom g;
    void* out;
    RawParamsDev raw;
    DeviceEdit ed;
    uint32_t tiles_x, ntiles;
};

template <int NT, int MODE, bool LDS_LUT, int OUTFMT>
__global__ __launch_bounds__(NT) void k_tile(const TileArgs a)
{
    constexpr int TILE_H = 2 * (NT / TILE_W);
    constexpr int TILE_ELEMS = (TILE_H + 4) * PITCH;
    extern __shared__ __align__(16) unsigned char smem[];
    float* tile = reinterpret_cast<float*>(smem);
    MathTables* T = reinterpret_cast<MathTables*>(smem + ((TILE_ELEMS * 4 + 15) & ~15));
    uint16_t* lut = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(T) + sizeof(MathTables));
    const int tid = threadIdx.x;
    const int W = (int)a.g.width, H = (int)a.g.height;
    const int y0 = (int)a.g.y0, rows = (int)a.g.rows;
    if (MODE == MODE_DEVELOP) {
        assert_lds_base(smem);
        stage_tables<NT, LDS_LUT>(a.ed, T, lut, tid);
    }
    const LdsTabs tabs((uint32_t)((TILE_ELEMS * 4 + 15) & ~15), T, a.ed.mp[0], false);
    const int lx = tid & (TILE_W - 1);
    const int ly = 2 * (tid / TILE_W);
    const int lane = tid & 63;
    DevLuts<LDS_LUT> luts{lut, a.ed.bright, a.ed.hsl, a.ed.hue_bounds, a.ed.hue_cs, a.ed.hue_fused};

    for (uint32_t t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const int tx0 = (int)(t % a.tiles_x) * TILE_W;
        const int ty0 = (int)(t / a.tiles_x) * TILE_H;
        __syncthreads();  // previous tile fully consumed (and tables staged)
        for (int i = tid; i < TILE_ELEMS; i += NT) {
            const int r = i / PITCH, c = i - r * PITCH;
            int gx = reflect101(tx0 - 2 + c, W);
            gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
            int gy = reflect101(y0 + ty0 - 2 + r, H);
            int br = gy - (y0 - 2);
            br = br < 0 ? 0 : (br > rows + 3 ? rows + 3 : br);
            const uint32_t raw = a.cfa[(size_t)br * W + gx];
            const int cell = ((gx + a.raw.xo) & 1) + 2 * ((gy + a.raw.yo) & 1);
            tile[i] = cfa_normalise(raw, cell, a.raw);
        }
        __syncthreads();
        const float* col = tile + ly * PITCH + lx + 2;
        float v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = col[i * PITCH];
        float p1[6], p2[6];
        const int hoff = lane == 0 ? -2 : (lane == 63 ? 1 : -(lx & 1));
#pragma unroll
        for (int i = 1; i <= 4; ++i) {
            const float2 hv = *reinterpret_cast<const float2*>(col + i * PITCH + hoff);
            const float l1 = from_left(v[i], hv.y);   // x-1 (lane 0: column -1 of the wavefront)
            const float r1 = from_right(v[i], hv.x);  // x+1 (lane 63: column 64)
            p1[i] = l1 + r1;
            if (i == 2 || i == 3) p2[i] = from_left(l1, hv.x) + from_right(r1, hv.y);
        }
        const int gx = tx0 + lx;
        const int cx = (gx + a.raw.xo) & 1;
        const int cy_top = (y0 + ty0 + ly + a.raw.yo) & 1;
        const float c[2] = {v[2], v[3]};
        const float s1[2] = {v[1] + v[3], v[2] + v[4]};
        const float e1[2] = {p1[2], p1[3]};
        const float d[2] = {p1[1] + p1[3], p1[2] + p1[4]};
        const float vv[2] = {v[0] + v[4], v[1] + v[5]};
        const float h2[2] = {p2[2], p2[3]};
        finish_pair<MODE, LDS_LUT, OUTFMT>(c, s1, e1, d, vv, h2, cx, cy_top, a.raw, a.ed, luts, tabs, gx, ty0 + ly, a.g,
                                           gx < W, a.out, a.raw.cam);
    }
}

#endif  // RPF_TU != 2

#if RPF_TU != 1
// ---------------------------------------------------------------------------
// k_apply — wgsl main on an RGBA32F tile (the reference's own pass), 2 pixels per thread
// ---------------------------------------------------------------------------
struct ApplyArgs {
    const float4* in;
    TileGeom g;
    void* out;
    DeviceEdit ed;
};

struct TexFetch {
    const float4* img; int w;
    __device__ __forceinline__ void operator()(int x, int y, float* rgb) const
    {
        const float4 v = img[(size_t)y * w + x];
        rgb[0] = v.x; rgb[1] = v.y; rgb[2] = v.z;
    }
};

// SINGLE: 0 = any number of masks, 1 = only "main", 2 = only "main" with an identity hue curve (HUE_DIRECT),
// 3 = only a pristine "main" (MF_PRISTINE known at compile time; sincos_hue as in 1)
//         5 (PREC = 1 only) = only "main", without a hue curve: the rotation form of edit_chain_fast
//         6 (PREC = 1 only) = user masks, no mask with a hue curve: the rotation form over all masks (0 with PREC = 1: the builtin route)
//         7 = more than RPF_MASKS_PER_WORD masks: the first 16 as in 0, then the further coverage planes, 16 masks each
//             (rpf_chain.h WideMasks); every mask's parameters staged in LDS from DeviceEdit::mp_wide (48 B per mask: up to
//             WIDE_LDS_MASKS of them)
//         8 = the same with the parameters read from device memory (scalar loads per mask and step): any number of masks
// PREC: 0 = exact (rpf_math.h), 1 = RPF_PRECISION_FAST (rpf_fast.h; SINGLE 3, 5 or 1, no lens resample, active flavour)
template <int NT, bool LDS_LUT, bool LENS, int OUTFMT, int SINGLE, int PREC = 0>
__global__ __launch_bounds__(NT) void k_apply(const ApplyArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t TAB_BYTES = TabsOf<PREC>::BYTES;
    uint16_t* lut = reinterpret_cast<uint16_t*>(smem + TAB_BYTES);
    const int tid = threadIdx.x;
    constexpr uint32_t LDS_MP = TAB_BYTES + (LDS_LUT ? (uint32_t)RPF_CURVE_RESOLUTION * 2u : 0u);
    assert_lds_base(smem);
    stage16<NT, (int)(TAB_BYTES / 16)>(PREC ? reinterpret_cast<const uint4*>(a.ed.fast_tabs) : reinterpret_cast<const uint4*>(a.ed.tabs),
                                       reinterpret_cast<uint4*>(smem), tid);
    constexpr bool MULTI = SINGLE == 0 || SINGLE == 6;
    if (LDS_LUT)  // "main"'s curve, or — launches with user masks whose "main" has none — a user mask's (DeviceEdit::lds_bright_mask)
        stage16<NT, RPF_CURVE_RESOLUTION * 2 / 16>(reinterpret_cast<const uint4*>(a.ed.bright + (size_t)(MULTI ? a.ed.lds_bright_mask : 0u) * RPF_CURVE_RESOLUTION),
                                                   reinterpret_cast<uint4*>(lut), tid);
    if (MULTI) stage_mask_params(a.ed, smem + LDS_MP, tid, NT);
    if (SINGLE == 7) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.ed.mp_wide);
        uint32_t* dst = reinterpret_cast<uint32_t*>(smem + LDS_MP);
        for (uint32_t i = tid; i < a.ed.num_masks * (uint32_t)(sizeof(MaskParams) / 4); i += NT) dst[i] = src[i];
    }
    __syncthreads();
    DevLuts<LDS_LUT> luts{lut, a.ed.bright, a.ed.hsl, a.ed.hue_bounds, a.ed.hue_cs, a.ed.hue_fused, MULTI ? a.ed.lds_bright_mask : 0u};
    const typename TabsOf<PREC>::type tabs = make_lds_tabs<PREC>(smem, a.ed.mp[0]);
    const int W = (int)a.g.width, H = (int)a.g.height;
    const uint32_t npix = a.g.rows * a.g.width;  // <= 2^31 (check_geom)
    const uint32_t stride = gridDim.x * NT * 2;
    // only these two stages look at the pixel position (the vignette not when its falloff comes from the map)
    const bool coords = LENS || (a.ed.vign_strength != 0.0f && ((a.ed.flags & EF_LEGACY_VIGNETTE) || !a.ed.vign_falloff));
    // One iteration of a workgroup = 2 * NT consecutive pixels starting at `base` (wave-uniform).  Input and output
    // of the iteration are addressed through buffer descriptors of exactly that span, built in scalar registers: a
    // thread's byte offsets inside the span never change, and pixels past the end of the image fall outside the
    // descriptor (loads return 0, stores are dropped) — no per-pixel 64-bit addresses, clamps or exec masks.
    // The pair's input is fetched one iteration ahead (the chain that follows is ~600 instructions long).
    // (only r, g, b are loaded: the register of an unused alpha gets reused by the allocator while the load is
    // still in flight, which forces a wait for the prefetch at that point)
    typedef uint32_t v3u __attribute__((ext_vector_type(3)));
    constexpr uint32_t BPP = out_bpp<OUTFMT>();
    v3u nxt[2];
    // the vignette's falloff of the pair's pixels, from the editor's map when there is one (active flavour; rpf_launch.h
    // DeviceEdit::vign_falloff), fetched with the pixels
    const float* fmap = (a.ed.vign_strength != 0.0f && !(a.ed.flags & EF_LEGACY_VIGNETTE)) ? a.ed.vign_falloff : nullptr;
    float nfall[2] = {0.0f, 0.0f};
    auto span = [&](uint32_t base) { return base >= npix ? 0u : (npix - base < 2u * NT ? npix - base : 2u * NT); };
    auto fetch = [&](uint32_t base) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float4*>(a.in) + base, 0, (int)(span(base) * 16u), RPF_BUFFER_RSRC_FLAGS);
#pragma unroll
        for (int k = 0; k < 2; ++k) nxt[k] = __builtin_amdgcn_raw_buffer_load_b96(rs, ((uint32_t)k * NT + tid) * 16u, 0, 2);
        if (fmap) {
            const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(fmap) + base, 0, (int)(span(base) * 4u), RPF_BUFFER_RSRC_FLAGS);
#pragma unroll
            for (int k = 0; k < 2; ++k)
                nfall[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fr, ((uint32_t)k * NT + tid) * 4u, 0, 0));
        }
    };
    uint32_t base = blockIdx.x * NT * 2;
    if (!LENS && base < npix) fetch(base);
    // The pixels' coordinates (vignette): one division per thread at the start, then carried from step to step — a step
    // moves every pixel by the same (stride / W, stride % W).  (Per pixel and step the 32-bit division alone was ~25
    // instructions.)  Pixels past the end of the image get coordinates too; nothing of theirs is stored.
    const uint32_t step_y = stride / (uint32_t)W, step_x = stride - step_y * (uint32_t)W;
    uint32_t cx[2], cy[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t p = base + (uint32_t)k * NT + tid;
        cy[k] = p / (uint32_t)W;
        cx[k] = p - cy[k] * (uint32_t)W;
    }
    // 1.0f for the alpha channel in two launch-long registers, one per store of an iteration (see hp_finish)
    uint32_t alpha[2] = {0x3f800000u, 0x3f800000u};
    asm volatile("" : "+v"(alpha[0]), "+v"(alpha[1]));
    ProgressPrio prio(base < npix ? (npix - base + stride - 1) / stride : 0u);  // this workgroup's iterations
    for (; base < npix; base += stride) {
        float r[2], g[2], b[2];
        uint32_t active[2] = {1u, 1u};
        uint32_t pix[2];
        bool live[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t p = base + (uint32_t)k * NT + tid;
            live[k] = p < npix;
            pix[k] = live[k] ? p : npix - 1;
            if (!LENS) { r[k] = bits2f(nxt[k].x); g[k] = bits2f(nxt[k].y); b[k] = bits2f(nxt[k].z); }
        }
        float fall[2] = {nfall[0], nfall[1]};
        if (LENS && fmap) { fall[0] = fmap[pix[0]]; fall[1] = fmap[pix[1]]; }
        if (!LENS) fetch(base + stride);  // past the end of the image the span is empty: zeros that nobody uses
        const bool decode = (a.ed.flags & EF_DECODE_INPUT) != 0;  // Godot-legacy :281-283: the stored image is sRGB-encoded
        auto decode_px = [&](int k) {
            const PowTable& D = a.ed.legacy->decode;
            r[k] = srgb_to_linear1(r[k], D); g[k] = srgb_to_linear1(g[k], D); b[k] = srgb_to_linear1(b[k], D);
        };
        if (coords) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                // LENS (whole frame, every source coordinate must exist): from the clamped pixel index
                const int ry = LENS ? (int)(pix[k] / (uint32_t)W) : (int)cy[k];
                const int x = LENS ? (int)(pix[k] - (uint32_t)ry * (uint32_t)W) : (int)cx[k];
                const int gy = (int)a.g.y0 + ry;
                if (LENS)  // wgsl:123-163; a.in is the whole frame
                    lens_sample(TexFetch{a.in, W}, W, H, x, gy, a.ed.lens_strength, r[k], g[k], b[k]);
                if (decode) decode_px(k);
                if (a.ed.vign_strength != 0.0f && !fmap) {
                    const float gain = (a.ed.flags & EF_LEGACY_VIGNETTE)
                                           ? vignette_gain_legacy(a.ed.vign_strength, x, gy, (float)W, (float)H)
                                           : vignette_gain(a.ed.vign_strength, x, gy, (float)W, (float)H);
                    r[k] *= gain; g[k] *= gain; b[k] *= gain;
                }
            }
        } else if (decode) {
            decode_px(0);
            decode_px(1);
        }
        if (fmap) {  // (after the lens resample and the legacy decode, where the chain has it)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float gain = vignette_gain_from_falloff(a.ed.vign_strength, fall[k]);
                r[k] *= gain; g[k] *= gain; b[k] *= gain;
            }
        }
        if ((SINGLE == 0 || SINGLE == 6 || SINGLE == 7 || SINGLE == 8) && a.ed.num_masks > 1)
            for (int k = 0; k < 2; ++k) active[k] = active_masks(a.ed, pix[k], npix);
        if constexpr (PREC == 1) {
            if (SINGLE == 3) edit_chain_fast<2, (int)MF_PRISTINE, true>(r, g, b, active, a.ed.mp, 1u, luts, tabs);
            else if (SINGLE == 5) edit_chain_fast<2, -1, true>(r, g, b, active, a.ed.mp, 1u, luts, tabs);
            else if (SINGLE == 6) edit_chain_fast<2, -1, true>(r, g, b, active, LdsMaskParams{a.ed.mp, LDS_MP}, a.ed.num_masks, luts, tabs);
            else if (SINGLE == 0) edit_chain_fast<2>(r, g, b, active, LdsMaskParams{a.ed.mp, LDS_MP}, a.ed.num_masks, luts, tabs);
            else edit_chain_fast<2, -1, false, true>(r, g, b, active, a.ed.mp, 1u, luts, tabs);  // a hue curve: the fused hue table
        } else if (SINGLE == 3)
            edit_chain<2, 0, (int)MF_PRISTINE>(r, g, b, active, a.ed.mp, 1u, luts, tabs);
        else if (SINGLE == 2)
            edit_chain<2, 1>(r, g, b, active, a.ed.mp, 1u, luts, tabs);
        else if (SINGLE == 0)
            edit_chain<2>(r, g, b, active, LdsMaskParams{a.ed.mp, LDS_MP}, a.ed.num_masks, luts, tabs);
        else if (SINGLE == 4)  // one mask with hue / saturation / lightness curves: through the editor's fused hue table
            edit_chain<2, 2>(r, g, b, active, a.ed.mp, 1u, luts, tabs);
        else if (SINGLE == 7)
            edit_chain<2>(r, g, b, active, LdsMaskParams{a.ed.mp, LDS_MP}, (uint32_t)RPF_MASKS_PER_WORD, luts, tabs,
                          WideMasks<2>{a.ed.num_masks, a.ed.mask_bits, (size_t)npix, {pix[0], pix[1]}});
        else if (SINGLE == 8)
            edit_chain<2>(r, g, b, active, a.ed.mp_wide, (uint32_t)RPF_MASKS_PER_WORD, luts, tabs,
                          WideMasks<2>{a.ed.num_masks, a.ed.mask_bits, (size_t)npix, {pix[0], pix[1]}});
        else
            edit_chain<2>(r, g, b, active, a.ed.mp, 1u, luts, tabs);
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<unsigned char*>(a.out) + (size_t)base * BPP, 0, (int)(span(base) * BPP), RPF_BUFFER_RSRC_FLAGS);
#pragma unroll
        for (int k = 0; k < 2; ++k) store_px_buf<OUTFMT>(ors, ((uint32_t)k * NT + tid) * BPP, r[k], g[k], b[k], alpha[k]);
        if (coords && !LENS) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                cx[k] += step_x; cy[k] += step_y;
                const bool wrap = cx[k] >= (uint32_t)W;
                cx[k] -= wrap ? (uint32_t)W : 0u;
                cy[k] += wrap ? 1u : 0u;
            }
        }
        prio.step_done();
    }
}

// ---------------------------------------------------------------------------
// layout helpers (image.rs:135-139, 202-277, 375-383)
// ---------------------------------------------------------------------------
template <int OUTFMT>
__global__ __launch_bounds__(256) void k_repack(const float4* in, size_t npix, void* out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        const float4 v = in[i];
        store_px<OUTFMT>(out, i, v.x, v.y, v.z);
    }
}

// user-mask planes (u8, 1 = covered) -> one u16 of mask bits per pixel (DeviceEdit::mask_bits)
__global__ __launch_bounds__(256) void k_combine_masks(const uint8_t* planes, uint32_t nplanes, size_t npix, uint16_t* bits)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        uint32_t v = 0;
        for (uint32_t m = 0; m < nplanes; ++m) v |= (uint32_t)(planes[(size_t)m * npix + i] == 1) << m;
        bits[i] = (uint16_t)v;
    }
}

// one RGBA32F pixel, nontemporal: these frames are written once and read by a later launch
__device__ __forceinline__ void store_nt(float4* p, float x, float y, float z, float w)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(v4f{x, y, z, w}, reinterpret_cast<v4f*>(p));
}

__global__ __launch_bounds__(256) void k_pad_rgb(const float* in, size_t npix, float4* out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256)
        store_nt(&out[i], in[i * 3], in[i * 3 + 1], in[i * 3 + 2], 1.0f);
}

// 16-bit PPM payload -> RGBA32F (web-ts/core/image.ts:172-189): big-endian samples, v / 65535 (binary64 in the
// JavaScript, rounded once into the Float32Array: equal to the binary32 quotient for all 65536 values, checked in
// tests/test_ppm.py), alpha 1.
__global__ __launch_bounds__(256) void k_rgb16be(const uint16_t* in, size_t npix, float4* out)
{
    const float inv = (float)(1.0 / 65535.0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        const uint32_t r = __builtin_bswap16(in[i * 3]), g = __builtin_bswap16(in[i * 3 + 1]),
                       b = __builtin_bswap16(in[i * 3 + 2]);
        store_nt(&out[i], div_u16_by(r, 65535.0f, inv), div_u16_by(g, 65535.0f, inv), div_u16_by(b, 65535.0f, inv), 1.0f);
    }
}

// ---------------------------------------------------------------------------
// k_rgb_int — decoded 8- / 16-bit interleaved RGB -> RGBA32F (image.rs:443 to_rgb32f + the alpha pad of :135-139):
// v / 255 or v / 65535, one IEEE binary32 division, or lut[v] when the caller supplies the colour-space step as a
// table.  One pixel per lane: three narrow loads that the wavefront covers with 192 / 384 contiguous bytes, one 16-B
// store; 19 / 22 B of HBM traffic per pixel bound it, not the instruction count.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_rgb_int(const T* in, size_t npix, const float* lut, float4* out)
{
    constexpr float MAXV = sizeof(T) == 1 ? 255.0f : 65535.0f;
    __shared__ float lds_lut[256];  // the 8-bit table: three look-ups per pixel, served from LDS (16-bit: 256 KiB, from L2)
    if (sizeof(T) == 1 && lut) {
        lds_lut[threadIdx.x] = lut[threadIdx.x];
        __syncthreads();
    }
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    // the output as one buffer (npix * 16 bytes, below 4 GiB: launch_rgb_int checks); stores nontemporal like the other writers
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(uint32_t)(npix * 16u), RPF_BUFFER_RSRC_FLAGS);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        const uint32_t r = in[i * 3], g = in[i * 3 + 1], b = in[i * 3 + 2];
        float x, y, z;
        if (lut) {  // kernel-uniform
            if (sizeof(T) == 1) { x = lds_lut[r]; y = lds_lut[g]; z = lds_lut[b]; }
            else { x = lut[r]; y = lut[g]; z = lut[b]; }
        } else {
            x = (float)r / MAXV; y = (float)g / MAXV; z = (float)b / MAXV;  // correctly rounded (hipcc default)
        }
        __builtin_amdgcn_raw_buffer_store_b128(v4u{f2bits(x), f2bits(y), f2bits(z), 0x3f800000u}, ors, (uint32_t)i * 16u, 0, 2);
    }
}

// ---------------------------------------------------------------------------
// k_resize — the preview pyramid's bilinear resize (web/main.ts:958-1026 resizeFloat32RGBLongEdge),
// RGBA32F -> RGBA32F on the device, binary64 arithmetic in the JavaScript source order, one thread
// per destination pixel.  Alpha is rewritten as 1.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize(const float4* src, uint32_t sw, uint32_t sh, float4* dst, uint32_t dw,
                                                uint32_t dh)
{
    const double scaleX = (double)sw / (double)dw, scaleY = (double)sh / (double)dh;
    const size_t n = (size_t)dw * dh;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t y = (uint32_t)(i / dw), x = (uint32_t)(i - (size_t)y * dw);
        const double sy = ((double)y + 0.5) * scaleY - 0.5;
        const double sx = ((double)x + 0.5) * scaleX - 0.5;
        const double fy = __builtin_floor(sy), fx = __builtin_floor(sx);
        const int64_t y0 = fy > 0 ? (int64_t)fy : 0, x0 = fx > 0 ? (int64_t)fx : 0;
        const int64_t y1 = y0 + 1 < (int64_t)sh - 1 ? y0 + 1 : (int64_t)sh - 1;
        const int64_t x1 = x0 + 1 < (int64_t)sw - 1 ? x0 + 1 : (int64_t)sw - 1;
        const double ty = sy - (double)y0, tx = sx - (double)x0;
        const float4 p00 = src[(size_t)y0 * sw + x0], p10 = src[(size_t)y0 * sw + x1];
        const float4 p01 = src[(size_t)y1 * sw + x0], p11 = src[(size_t)y1 * sw + x1];
        const float c00[3] = {p00.x, p00.y, p00.z}, c10[3] = {p10.x, p10.y, p10.z};
        const float c01[3] = {p01.x, p01.y, p01.z}, c11[3] = {p11.x, p11.y, p11.z};
        float o[3];
        for (int c = 0; c < 3; ++c) {
            const double cx0 = (double)c00[c] * (1.0 - tx) + (double)c10[c] * tx;
            const double cx1 = (double)c01[c] * (1.0 - tx) + (double)c11[c] * tx;
            o[c] = (float)(cx0 * (1.0 - ty) + cx1 * ty);
        }
        dst[i] = make_float4(o[0], o[1], o[2], 1.0f);
    }
}

// ---------------------------------------------------------------------------
// k_touch — a streaming read of a buffer that was just written by somebody else (an upload, a decoder, the halo
// exchange): it lands the buffer in the 256-MiB Infinity Cache, from where the fused kernel's narrow row loads (128 B
// per wave-instruction, spread over its whole run time and interleaved with 2 TB/s of stores) come back much faster than
// from HBM: measured, a 6000x4000 frame fresh from HBM costs the fused kernel +37 us, this pass ~11 us
// (profiles/r02_experiments.md).  16 B per lane, grid-stride; the values only have to be consumed.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_touch(const uint4* p, size_t n16)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    asm volatile("" ::"v"(acc));  // consumed: the loads cannot be dropped
}

// ---------------------------------------------------------------------------
// k_unpack — packed RAW (10/12/14/16-bit; MSB-first bit stream, or LSB-first = little-endian) -> u16 CFA plane.
// Integer, bit-exact.
// One thread = 16 samples = 2*BITS bytes read as BITS/2 dwords and two 16-B stores; a group that is
// ragged (row end) or unaligned goes sample by sample.
// ---------------------------------------------------------------------------
template <int BITS, bool LSB>
__global__ __launch_bounds__(256) void k_unpack(const uint8_t* packed, size_t row_stride, uint32_t width, uint32_t rows,
                                                uint16_t* out, uint32_t groups_per_row)
{
    constexpr int NW = BITS / 2;  // dwords per 16-sample group
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const size_t total = (size_t)groups_per_row * rows;
    const bool aligned = (row_stride & 3u) == 0 && (width & 7u) == 0 && ((size_t)packed & 3u) == 0 && ((size_t)out & 15u) == 0;
    for (size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (size_t)gridDim.x * 256) {
        const uint32_t y = (uint32_t)(gi / groups_per_row), g = (uint32_t)(gi - (size_t)y * groups_per_row);
        const uint8_t* row = packed + (size_t)y * row_stride;
        uint16_t* orow = out + (size_t)y * width;
        const uint32_t x0 = g * 16u;
        if (aligned && x0 + 16u <= width && (size_t)(g + 1) * (2 * BITS) <= row_stride) {
            uint32_t w[NW + 1];
            const uint32_t* src = reinterpret_cast<const uint32_t*>(row + (size_t)g * (2 * BITS));
#pragma unroll
            for (int k = 0; k < NW; ++k) w[k] = LSB ? src[k] : __builtin_bswap32(src[k]);  // words of the bit stream
            w[NW] = 0;
            uint32_t v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int o = j * BITS, i = o >> 5, sh = o & 31;
                if (LSB) {
                    const uint64_t two = ((uint64_t)w[i + 1] << 32) | w[i];
                    v[j] = (uint32_t)(two >> sh) & MASK;
                } else {
                    const uint64_t two = ((uint64_t)w[i] << 32) | w[i + 1];
                    v[j] = (uint32_t)(two >> (64 - sh - BITS)) & MASK;
                }
            }
            uint4 a, b;
            a.x = v[0] | (v[1] << 16); a.y = v[2] | (v[3] << 16); a.z = v[4] | (v[5] << 16); a.w = v[6] | (v[7] << 16);
            b.x = v[8] | (v[9] << 16); b.y = v[10] | (v[11] << 16); b.z = v[12] | (v[13] << 16); b.w = v[14] | (v[15] << 16);
            uint4* dst = reinterpret_cast<uint4*>(orow + x0);
            dst[0] = a; dst[1] = b;
        } else {
            for (uint32_t x = x0; x < x0 + 16u && x < width; ++x) {
                const uint64_t bit = (uint64_t)x * BITS;
                const size_t by = (size_t)(bit >> 3);
                const uint32_t sh = (uint32_t)(bit & 7);
                // BITS + 7 <= 23 bits span at most 3 bytes (4 for 16-bit samples never: sh is 0 there)
                if (LSB) {
                    uint32_t acc = (uint32_t)row[by] | ((uint32_t)row[by + 1] << 8);
                    if (sh + BITS > 16) acc |= (uint32_t)row[by + 2] << 16;
                    orow[x] = (uint16_t)((acc >> sh) & MASK);
                } else {
                    uint32_t acc = ((uint32_t)row[by] << 16) | ((uint32_t)row[by + 1] << 8);
                    if (sh + BITS > 16) acc |= row[by + 2];
                    orow[x] = (uint16_t)((acc >> (24 - sh - BITS)) & MASK);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_unpack_mipi — MIPI CSI-2 RAW10 / RAW12 / RAW14 (RPF_PACK_MIPI) -> u16 CFA plane.  Groups of N = 4 / 2 / 4 pixels in
// 5 / 3 / 7 bytes: N high bytes, then the pixels' BITS - 8 low bits LSB-first in pixel order.  16 samples are 2*BITS
// bytes here too, so the thread shape of k_unpack is kept: BITS/2 dword loads, byte picks at compile-time positions,
// two 16-B stores; ragged / unaligned groups go sample by sample.
// ---------------------------------------------------------------------------
template <int BITS>
__global__ __launch_bounds__(256) void k_unpack_mipi(const uint8_t* packed, size_t row_stride, uint32_t width,
                                                     uint32_t rows, uint16_t* out, uint32_t groups_per_row)
{
    constexpr int NW = BITS / 2;                   // dwords per 16 samples
    constexpr int N = BITS == 12 ? 2 : 4;          // pixels per CSI-2 group
    constexpr int LOW = BITS - 8;                  // low bits per pixel
    constexpr int GB = N + N * LOW / 8;            // bytes per group
    constexpr uint32_t LMASK = (1u << LOW) - 1u;
    const size_t total = (size_t)groups_per_row * rows;
    const bool aligned = (row_stride & 3u) == 0 && (width & 7u) == 0 && ((size_t)packed & 3u) == 0 && ((size_t)out & 15u) == 0;
    for (size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (size_t)gridDim.x * 256) {
        const uint32_t y = (uint32_t)(gi / groups_per_row), g = (uint32_t)(gi - (size_t)y * groups_per_row);
        const uint8_t* row = packed + (size_t)y * row_stride;
        uint16_t* orow = out + (size_t)y * width;
        const uint32_t x0 = g * 16u;
        if (aligned && x0 + 16u <= width && (size_t)(g + 1) * (2 * BITS) <= row_stride) {
            uint32_t w[NW];
            const uint32_t* src = reinterpret_cast<const uint32_t*>(row + (size_t)g * (2 * BITS));
#pragma unroll
            for (int k = 0; k < NW; ++k) w[k] = src[k];
            uint32_t v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int q = j / N, i = j % N, base = q * GB;
                const int hb = base + i;                       // the pixel's high byte
                const int lbit = (base + N) * 8 + i * LOW;     // first of its low bits, as a bit index into the 2*BITS bytes
                const uint32_t hi = (w[hb >> 2] >> ((hb & 3) * 8)) & 0xffu;
                const int li = lbit >> 5, ls = lbit & 31;
                uint32_t lo = w[li] >> ls;
                if (ls + LOW > 32) lo |= w[li + 1 < NW ? li + 1 : li] << (32 - ls);
                v[j] = (hi << LOW) | (lo & LMASK);
            }
            uint4 a, b;
            a.x = v[0] | (v[1] << 16); a.y = v[2] | (v[3] << 16); a.z = v[4] | (v[5] << 16); a.w = v[6] | (v[7] << 16);
            b.x = v[8] | (v[9] << 16); b.y = v[10] | (v[11] << 16); b.z = v[12] | (v[13] << 16); b.w = v[14] | (v[15] << 16);
            uint4* dst = reinterpret_cast<uint4*>(orow + x0);
            dst[0] = a; dst[1] = b;
        } else {
            for (uint32_t x = x0; x < x0 + 16u && x < width; ++x) {
                const uint8_t* grp = row + (size_t)(x / N) * GB;
                const uint32_t i = x % N, bit = i * LOW;
                uint32_t lo = (uint32_t)grp[N + (bit >> 3)] >> (bit & 7);
                if ((bit & 7) + LOW > 8) lo |= (uint32_t)grp[N + (bit >> 3) + 1] << (8 - (bit & 7));
                orow[x] = (uint16_t)(((uint32_t)grp[i] << LOW) | (lo & LMASK));
            }
        }
    }
}

#endif  // RPF_TU != 1

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
template <class K>
static hipError_t set_lds(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

static bool all_hue_identity(const DeviceEdit& ed)
{
    for (uint32_t m = 0; m < ed.num_masks; ++m)
        if (!(ed.mp[m].flags & MF_HUE_IDENTITY)) return false;
    return true;
}
static bool needs_lds_lut(const DeviceEdit& ed) { return !(ed.mp[0].flags & MF_BRIGHT_IDENTITY); }

// Operand checks shared by the launchers: a tile must be non-empty, inside its frame and large
// enough for the reflect-101 border.
static hipError_t check_geom(const TileGeom& g, bool bayer)
{
    if (g.width == 0 || g.height == 0 || g.rows == 0) return hipErrorInvalidValue;
    if ((uint64_t)g.y0 + g.rows > g.height) return hipErrorInvalidValue;
    if (bayer && (g.width < 3 || g.height < 3)) return hipErrorInvalidValue;
    if ((uint64_t)g.width * g.rows > (1ull << 31)) return hipErrorInvalidValue;
    // the fused kernels address the (rows + 4)-row u16 band and one output row through 32-bit buffer offsets
    if (bayer && (uint64_t)g.width * ((uint64_t)g.rows + 4) * 2 >= (1ull << 32)) return hipErrorInvalidValue;
    if ((uint64_t)g.width * 16 >= (1ull << 32)) return hipErrorInvalidValue;
    return hipSuccess;
}

static hipError_t check_edit(const DeviceEdit& ed)
{
    if (!ed.tabs || !ed.hue_bounds || !ed.hue_cs || !ed.bright || !ed.hsl || ed.num_masks < 1) return hipErrorInvalidValue;
    if (ed.num_masks > RPF_MASKS_PER_WORD && !ed.mp_wide) return hipErrorInvalidValue;
    if (ed.num_masks > 1 && !ed.mask_bits) return hipErrorInvalidValue;
    return hipSuccess;
}

#if RPF_TU != 2
// A fused develop launch; with cfg.stop_event its completion signal is that event (no marker packet behind the kernel).
template <class K, class A>
static inline void launch_fused(const LaunchCfg& cfg, K kernel, uint32_t grid, int nt, size_t lds, hipStream_t s, const A& a)
{
    if (cfg.stop_event)
        hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(nt), lds, s, static_cast<hipEvent_t>(cfg.start_event),
                              static_cast<hipEvent_t>(cfg.stop_event), 0, a);
    else
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(nt), lds, s, a);
}

struct PackedRows {  // bit-packed input of the fused kernel (bits == 0: d_cfa is a u16 plane)
    uint32_t bits, row_stride, lsb;
};

template <int NT, int MODE, bool LDS_LUT, int OUTFMT, int PACKED = 0, int PREC = 0>
static hipError_t run_stream(const LaunchCfg& cfg, const uint16_t* d_cfa, const RawParamsDev& raw, const TileGeom& g,
                             const DeviceEdit* ed, void* d_out, hipStream_t s, uint32_t seg_pairs = 0,
                             uint32_t blocks_per_cu = 0, PackedRows pk = PackedRows{0, 0, 0})
{
    StreamArgs a;
    memset(&a, 0, sizeof(a));
    a.cfa = d_cfa; a.g = g; a.out = d_out; a.raw = raw;
    a.bits = pk.bits; a.row_stride = pk.row_stride; a.lsb = pk.lsb;
    a.row_bytes = (uint32_t)(((uint64_t)g.width * pk.bits + 7) / 8);
    if (pk.lsb == 2u) {  // MIPI CSI-2: whole groups of 4 / 2 / 4 pixels in 5 / 3 / 7 bytes
        const uint32_t n = pk.bits == 12u ? 2u : 4u;
        a.row_bytes = (g.width + n - 1) / n * (n + n * (pk.bits - 8u) / 8u);
    }
    if (ed) a.ed = *ed;
    const uint32_t total_pairs = (g.rows + 1) / 2;
    a.strips = (g.width + STRIP_W - 1) / STRIP_W;
    constexpr uint32_t WPB = NT / 64;
    if (PREC && (!ed || !ed->fast_tabs)) return hipErrorInvalidValue;
    const size_t lds = MODE != MODE_DEMOSAIC ? TabsOf<PREC>::BYTES + (LDS_LUT ? (size_t)RPF_CURVE_RESOLUTION * 2 : 0) +
                                                   (mode_multi(MODE) ? sizeof(MaskParams) * RPF_MASKS_PER_WORD : 0)
                                             : 0;
    // two builds of every fused kernel: with the vignette stage (coordinates hoisted per column / per row) and
    // without it (its code and registers compiled out: the plain kernel stays under 96 VGPRs)
    const bool vign = MODE != MODE_DEMOSAIC && ed && ed->vign_strength != 0.0f;
    auto kernel = vign ? k_stream<NT, MODE, LDS_LUT, OUTFMT, PACKED, MODE != MODE_DEMOSAIC, PREC>
                       : k_stream<NT, MODE, LDS_LUT, OUTFMT, PACKED, false, PREC>;
    hipError_t e = set_lds(kernel, lds);
    if (e != hipSuccess) return e;
    // persistent workgroups for the fused kernel (LDS tables staged once per workgroup), as many as are resident: it
    // needs < 96 VGPRs, i.e. 5 waves per SIMD (variant 6: 4, for A/B); the 128-KiB curve in LDS limits that variant
    // to one 1024-thread workgroup per CU
    uint32_t per_cu = LDS_LUT ? 1u : (uint32_t)((cfg.variant == 6 ? 1024 : 1280) / NT);
    if (MODE != MODE_DEMOSAIC) {  // never more than are resident (a register count over 96 would halve the last round)
        // per build (without / with the vignette stage); atomic: two host threads may launch for the first time
        // together (both then store the same value)
        static std::atomic<int> resident[2];
        int nb = resident[vign].load(std::memory_order_relaxed);
        if (nb == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, NT, lds) != hipSuccess || nb < 1) nb = 1;
            resident[vign].store(nb, std::memory_order_relaxed);
        }
        if (per_cu > (uint32_t)nb) per_cu = (uint32_t)nb;
    }
    const uint32_t cap_waves = (uint32_t)cfg.num_cus * per_cu * WPB;
    auto items_for = [&](uint32_t sp) { return a.strips * ((total_pairs + sp - 1) / sp); };
    // rows per item: 8 (demosaic: measured best; fused: enough waves for small frames); the fused kernel on a frame
    // with more than two rounds of 32-row items (16-row with the curve in LDS: 4096 instead of 5120 waves) uses those
    // (window start-up amortised: -1 % on 6000x4000)
    a.seg_pairs = seg_pairs ? seg_pairs : 4;
    const uint32_t big = LDS_LUT ? 8 : 16;
    if (!seg_pairs && MODE != MODE_DEMOSAIC && cfg.variant != 9 && items_for(big) >= 2 * cap_waves) a.seg_pairs = big;
    // Whole rounds: the fewest rounds R for which a strip cut into floor(R * waves / strips) equal segments gives items
    // of at most 30 row pairs (an item's window rows must fit the 64 lanes that hold their offsets): every wave gets R
    // (a few: R - 1) items of the same size and there is no ragged last round to balance.  Measured on 6000x4000: -3.4 %
    // for the 1024-thread kernels that keep the brightness curve in LDS (4096 waves, 16-row items before), no change
    // for the 256-thread kernels (variant 14 forces it there, variant 15 switches it off everywhere; A/B).
    bool whole_rounds = false;
    // A launch too small for two rounds of 32-row items (a row band of a frame sharded over several GPUs: 6000 x 500 is
    // 5734 eight-row items for 5120 waves, i.e. one round plus a tail, every item paying its 4-row window start-up) takes
    // whole rounds with any workgroup size: 29.6 -> see profiles/r03_experiments.md.
    const bool small_launch = MODE != MODE_DEMOSAIC && a.seg_pairs != big;
    if (!seg_pairs && MODE != MODE_DEMOSAIC && cfg.variant != 15 && cfg.variant != 8 && cfg.variant != 9 &&
        (LDS_LUT || cfg.variant == 14 || small_launch)) {
        for (uint32_t r = 1; r <= 8 && !whole_rounds; ++r) {
            const uint32_t segs = r * cap_waves / a.strips;
            if (segs == 0) continue;
            const uint32_t sp = (total_pairs + segs - 1) / segs;
            if (sp >= 4 && sp <= 30) { a.seg_pairs = sp; whole_rounds = true; }
        }
    }
    a.nitems = items_for(a.seg_pairs);
    uint32_t grid = (a.nitems + WPB - 1) / WPB;  // demosaic only: one item per wave, the hardware balances the blocks
    a.tail_start = a.nvirt = a.nitems;
    if (MODE != MODE_DEMOSAIC) {
        const uint32_t cap = (uint32_t)cfg.num_cus * per_cu;
        if (grid > cap) grid = cap;
        if (cfg.variant != 8 && !whole_rounds) {  // tail balancing (variant 8: off, for A/B): -1 % (D) to -3 % (S) of kernel time
            const uint32_t nwaves = grid * WPB;
            if (a.nitems > nwaves) {  // a launch of at most one round gains nothing from four window start-ups per wave
                a.tail_start = a.nitems / nwaves * nwaves;
                a.nvirt = a.tail_start + (a.nitems - a.tail_start) * a.seg_pairs;
            }
        }
    } else if (blocks_per_cu) {
        const uint32_t cap = (uint32_t)cfg.num_cus * blocks_per_cu;
        if (grid > cap) grid = cap;
    }
    launch_fused(cfg, kernel, grid, NT, lds, s, a);
    return hipGetLastError();
}

// k_hp's launcher: 128-column strips that start on an R column, steps that start on an R row (see the kernel).
template <int NT, int MODE, bool LDS_LUT, int OUTFMT, int PREC = 0>
static hipError_t run_hp(const LaunchCfg& cfg, const uint16_t* d_cfa, const RawParamsDev& raw, const TileGeom& g,
                         const DeviceEdit* ed, void* d_out, hipStream_t s)
{
    StreamArgs a;
    memset(&a, 0, sizeof(a));
    a.cfa = d_cfa; a.g = g; a.out = d_out; a.raw = raw;
    a.ed = *ed;
    const uint32_t bp = (g.y0 + (uint32_t)raw.yo) & 1u;
    const uint32_t total_pairs = (g.rows + bp + 1) / 2;
    a.strips = (g.width + (uint32_t)raw.xo + HP_STRIP_W - 1) / HP_STRIP_W;
    constexpr uint32_t WPB = NT / 64;
    const size_t lds = (hp_fast_biased(NT, LDS_LUT, PREC) ? FAST_TABS_BIASED_AT : 0u) + TabsOf<PREC>::BYTES +
                       (LDS_LUT ? (size_t)RPF_CURVE_RESOLUTION * 2 : 0) + (mode_multi(MODE) ? sizeof(MaskParams) * RPF_MASKS_PER_WORD : 0);
    const bool vign = ed->vign_strength != 0.0f;
    if (PREC && !ed->fast_tabs) return hipErrorInvalidValue;
    // k_hp reads the vignette's falloff from the tile's map (the API layer builds it when the slider leaves 0)
    if (vign && (!ed->vign_falloff || (uint64_t)g.rows * g.width * 4u > 0xffffffffull)) return hipErrorInvalidValue;
    auto kernel = vign ? k_hp<NT, MODE, LDS_LUT, OUTFMT, true, PREC> : k_hp<NT, MODE, LDS_LUT, OUTFMT, false, PREC>;
    hipError_t e = set_lds(kernel, lds);
    if (e != hipSuccess) return e;
    // persistent workgroups, as many as are resident: four 256-thread workgroups per CU (about 110 VGPRs: four waves per
    // SIMD), one 1024-thread workgroup where the 128-KiB curve is staged in LDS
    uint32_t per_cu = LDS_LUT ? 1u : (uint32_t)(1024 / NT);
    if (PREC && NT == 256 && cfg.variant == 30) per_cu = 5;  // A/B: the binary32 kernel needs < 96 VGPRs, five waves per SIMD fit
    {
        static std::atomic<int> resident[2];
        int nb = resident[vign].load(std::memory_order_relaxed);
        if (nb == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, NT, lds) != hipSuccess || nb < 1) nb = 1;
            resident[vign].store(nb, std::memory_order_relaxed);
        }
        if (per_cu > (uint32_t)nb) per_cu = (uint32_t)nb;
    }
    const uint32_t cap_waves = (uint32_t)cfg.num_cus * per_cu * WPB;
    auto items_for = [&](uint32_t sp) { return a.strips * ((total_pairs + sp - 1) / sp); };
    // Whole rounds (see run_stream): the fewest rounds R for which a strip cut into floor(R * waves / strips) equal
    // segments gives items of 4..30 row pairs; 6000 x 4000 on 4096 waves is ONE round of 4089 items of 23 pairs.
    // (a row band of a sharded frame is a small launch — 6000 x 500 is 2.9 steps per wave — so items go down to 2 pairs)
    a.seg_pairs = 4;
    bool whole_rounds = false;
    for (uint32_t r = 1; r <= 8 && !whole_rounds; ++r) {
        const uint32_t segs = r * cap_waves / a.strips;
        if (segs == 0) continue;
        const uint32_t sp = (total_pairs + segs - 1) / segs;
        if (sp >= 2 && sp <= 30) { a.seg_pairs = sp; whole_rounds = true; }
    }
    if (!whole_rounds && items_for(16) >= 2 * cap_waves) a.seg_pairs = 16;
    a.nitems = items_for(a.seg_pairs);
    uint32_t grid = (a.nitems + WPB - 1) / WPB;
    const uint32_t cap = (uint32_t)cfg.num_cus * per_cu;
    if (grid > cap) grid = cap;
    a.tail_start = a.nvirt = a.nitems;
    if (!whole_rounds) {  // the last, incomplete round in single-pair sub-items (see StreamArgs)
        const uint32_t nwaves = grid * WPB;
        if (a.nitems > nwaves) {
            a.tail_start = a.nitems / nwaves * nwaves;
            a.nvirt = a.tail_start + (a.nitems - a.tail_start) * a.seg_pairs;
        }
    }
    launch_fused(cfg, kernel, grid, NT, lds, s, a);
    return hipGetLastError();
}

// The fused kernel of a u16 band: k_hp, or the column-per-lane k_stream where a measurement variant asks for it.
template <int NT, int MODE, bool LDS_LUT, int OUTFMT>
static hipError_t run_fused(bool hp, const LaunchCfg& cfg, const uint16_t* d_cfa, const RawParamsDev& raw, const TileGeom& g,
                            const DeviceEdit* ed, void* d_out, hipStream_t s)
{
    // k_hp always as ONE 1024-thread workgroup per CU (the same four waves per SIMD as four 256-thread ones): the math
    // tables are then staged once per CU instead of four times — 7 instead of 29 MB of L2 -> LDS traffic per launch, which
    // is most of what a launch costs before its first pixel (variant 22, A/B: 256-thread workgroups)
    // (a vignette without its falloff map — a tile of more than 2^30 pixels, whose map would not fit a buffer descriptor —
    // goes to the column-per-lane kernel, which evaluates the falloff in place)
    if (hp && ed->vign_strength != 0.0f && !ed->vign_falloff) hp = false;
    if (hp && cfg.variant != 22) return run_hp<1024, MODE, LDS_LUT, OUTFMT>(cfg, d_cfa, raw, g, ed, d_out, s);
    return hp ? run_hp<NT, MODE, LDS_LUT, OUTFMT>(cfg, d_cfa, raw, g, ed, d_out, s)
              : run_stream<NT, MODE, LDS_LUT, OUTFMT>(cfg, d_cfa, raw, g, ed, d_out, s);
}

template <int NT, int MODE, bool LDS_LUT, int OUTFMT>
static hipError_t run_tile(const LaunchCfg& cfg, const uint16_t* d_cfa, const RawParamsDev& raw, const TileGeom& g,
                           const DeviceEdit* ed, void* d_out, int blocks_per_cu, hipStream_t s)
{
    constexpr int TILE_H = 2 * (NT / TILE_W);
    TileArgs a;
    memset(&a, 0, sizeof(a));
    a.cfa = d_cfa; a.g = g; a.out = d_out; a.raw = raw;
    if (ed) a.ed = *ed;
    a.tiles_x = (g.width + TILE_W - 1) / TILE_W;
    a.ntiles = a.tiles_x * ((g.rows + TILE_H - 1) / TILE_H);
    const size_t lds = (size_t)((((TILE_H + 4) * PITCH * 4) + 15) & ~15) +
                       (MODE == MODE_DEVELOP ? sizeof(MathTables) + (LDS_LUT ? (size_t)RPF_CURVE_RESOLUTION * 2 : 0) : 0);
    auto kernel = k_tile<NT, MODE, LDS_LUT, OUTFMT>;
    hipError_t e = set_lds(kernel, lds);
    if (e != hipSuccess) return e;
    uint32_t grid = (uint32_t)cfg.num_cus * (uint32_t)blocks_per_cu;
    if (grid > a.ntiles) grid = a.ntiles;
    launch_fused(cfg, kernel, grid, NT, lds, s, a);
    return hipGetLastError();
}

hipError_t launch_demosaic(const LaunchCfg& cfg, const uint16_t* d_cfa, const RawParamsDev& raw, const TileGeom& g,
                           float* d_rgba, hipStream_t s)
{
    hipError_t e = check_geom(g, true);
    if (e != hipSuccess) return e;
    if (!d_cfa || !d_rgba) return hipErrorInvalidValue;
    if (cfg.variant == 1) return run_tile<512, MODE_DEMOSAIC, false, OUT_RGBA_F32>(cfg, d_cfa, raw, g, nullptr, d_rgba, 4, s);
    switch (cfg.variant) {  // measurement variants (bench.py --variant); 0 is the shipped configuration
    case 3: return run_stream<256, MODE_DEMOSAIC, false, OUT_RGBA_F32_PLAIN>(cfg, d_cfa, raw, g, nullptr, d_rgba, s);
    case 4: return run_stream<256, MODE_DEMOSAIC, false, OUT_RGBA_F32>(cfg, d_cfa, raw, g, nullptr, d_rgba, s, 8);
    default: return run_stream<256, MODE_DEMOSAIC, false, OUT_RGBA_F32>(cfg, d_cfa, raw, g, nullptr, d_rgba, s);
    }
}

template <int OUTFMT>
static hipError_t develop_fmt(const LaunchCfg& cfg, const uint16_t* d_cfa, const RawParamsDev& raw, const TileGeom& g,
                              const DeviceEdit& ed, void* d_out, hipStream_t s)
{
    const bool lut = needs_lds_lut(ed) && cfg.variant != 10;  // variant 10 (A/B): the brightness curve gathered from L2
    if (cfg.variant == 1)  // LDS-tile front end (its tile + the tables + a 128-KiB curve exceed the LDS: curve from L2)
        return run_tile<512, MODE_DEVELOP, false, OUTFMT>(cfg, d_cfa, raw, g, &ed, d_out, 2, s);
    if (cfg.variant == 2)  // streaming, 512-thread workgroups
        return lut ? run_stream<512, MODE_DEVELOP, true, OUTFMT>(cfg, d_cfa, raw, g, &ed, d_out, s)
                   : run_stream<512, MODE_DEVELOP, false, OUTFMT>(cfg, d_cfa, raw, g, &ed, d_out, s);
    if (OUTFMT == OUT_RGBA_F32 && !lut && cfg.variant == 3)  // A/B: the pixel pair one after the other
        return run_stream<256, MODE_DEVELOP_SEQ, false, OUT_RGBA_F32>(cfg, d_cfa, raw, g, &ed, d_out, s);
    // the shipped kernel is k_hp; the column-per-lane k_stream stays for A/B: variant 20, and the variants that are
    // options of its launcher (6, 8, 9, 14, 15); 7, 10, 11, 12 choose the specialisation and apply to either
    const bool hp_variant = cfg.variant == 0 || cfg.variant == 22 || cfg.variant == 7 || cfg.variant == 10 || cfg.variant == 11 || cfg.variant == 12;
    // ... and for launches with user masks: a mask stage is a chain of dependent LDS reads and L2 gathers per mask, and
    // five waves per SIMD (k_stream: 93 VGPRs) hide that better than four (4 masks 0.324 vs 0.345 ms, 8 masks 0.467 vs
    // 0.538: profiles/r03_experiments.md); variant 21 (A/B) forces k_hp there too
    const bool hp = (hp_variant && ed.num_masks == 1) || cfg.variant == 21;
    // RPF_PRECISION_FAST (rpf_fast.h, edit_chain_fast): a specialisation of k_hp for launches with only "main" — pristine
    // (set D) or with run-time flags and curves.  Every other launch computes the exact chain, which meets the fast
    // mode's tolerances trivially.
    if (ed.precision == 1u && ed.num_masks == 1 && (cfg.variant == 0 || cfg.variant == 30 || cfg.variant == 31) &&
        !(ed.vign_strength != 0.0f && !ed.vign_falloff)) {
        if (ed.mp[0].flags == MF_PRISTINE && cfg.variant != 0)  // A/B (30: five, 31: four 256-thread workgroups per CU)
            return run_hp<256, MODE_DEVELOP_1P, false, OUTFMT, 1>(cfg, d_cfa, raw, g, &ed, d_out, s);
        if (ed.mp[0].flags == MF_PRISTINE) return run_hp<1024, MODE_DEVELOP_1P, false, OUTFMT, 1>(cfg, d_cfa, raw, g, &ed, d_out, s);
        if (ed.mp[0].flags & MF_HUE_IDENTITY)  // no hue curve: the chain needs neither chroma nor angle (edit_chain_fast's HUE_ROT)
            return lut ? run_hp<1024, MODE_DEVELOP_1H, true, OUTFMT, 1>(cfg, d_cfa, raw, g, &ed, d_out, s)
                       : run_hp<1024, MODE_DEVELOP_1H, false, OUTFMT, 1>(cfg, d_cfa, raw, g, &ed, d_out, s);
        if (!ed.hue_fused) return hipErrorInvalidValue;  // a hue curve without its table: the API layer builds it
        return lut ? run_hp<1024, MODE_DEVELOP_1, true, OUTFMT, 1>(cfg, d_cfa, raw, g, &ed, d_out, s)
                   : run_hp<1024, MODE_DEVELOP_1, false, OUTFMT, 1>(cfg, d_cfa, raw, g, &ed, d_out, s);
    }
    if (ed.num_masks == 1 && cfg.variant != 7) {  // variant 7 (A/B): the general kernel on a single mask
        if (ed.mp[0].flags == MF_PRISTINE && cfg.variant != 12)  // variant 12 (A/B): run-time flags also for a pristine "main"
            return run_fused<256, MODE_DEVELOP_1P, false, OUTFMT>(hp, cfg, d_cfa, raw, g, &ed, d_out, s);
        // hue / saturation / lightness curves that are not all trivial come folded into the editor's fused hue table
        // (MODE_DEVELOP_1); without such curves the index is the new hue (MODE_DEVELOP_1H); variant 11 (A/B): the general
        // multi-mask kernel for both
        if (cfg.variant != 11) {
            if (ed.hue_fused)
                return lut ? run_fused<1024, MODE_DEVELOP_1, true, OUTFMT>(hp, cfg, d_cfa, raw, g, &ed, d_out, s)
                           : run_fused<256, MODE_DEVELOP_1, false, OUTFMT>(hp, cfg, d_cfa, raw, g, &ed, d_out, s);
            if (ed.mp[0].flags & MF_HUE_IDENTITY)
                return lut ? run_fused<1024, MODE_DEVELOP_1H, true, OUTFMT>(hp, cfg, d_cfa, raw, g, &ed, d_out, s)
                           : run_fused<256, MODE_DEVELOP_1H, false, OUTFMT>(hp, cfg, d_cfa, raw, g, &ed, d_out, s);
            return hipErrorInvalidValue;  // a non-identity hue curve without its table: the API layer builds it
        }
    }
    // Launches with user masks.  There is room in LDS for ONE mask's brightness curve (128 KiB): "main"'s when it has one,
    // else the first user mask's that has one — three L2 gathers per pixel of that mask become LDS reads (a user-mask stage
    // costs 0.055 ms per full-frame mask, two thirds of it those gathers: profiles/r04_experiments.md).  Variant 23 (A/B):
    // only "main"'s, as before round 4.
    DeviceEdit edm = ed;
    bool lutm = lut;
    // (with the curve in LDS a CU runs one 1024-thread workgroup, four waves per SIMD instead of five: worth it while the
    // staged curve is a fair share of all the masks' gathers — measured up to 8 masks; at 16 it loses 10 %)
    if (!lut && cfg.variant != 10 && cfg.variant != 23 && ed.num_masks <= 8)
        for (uint32_t m = 1; m < ed.num_masks && !lutm; ++m)
            if (!(ed.mp[m].flags & MF_BRIGHT_IDENTITY)) { edm.lds_bright_mask = m; lutm = true; }
    // RPF_PRECISION_FAST with user masks (RGBA32F output; the other layouts compute the exact chain): the column-per-lane
    // kernel over edit_chain_fast — the rotation form when no mask has a hue curve
    if (OUTFMT == OUT_RGBA_F32 && ed.precision == 1u && ed.fast_tabs && cfg.variant == 0 && ed.num_masks > 1) {
        if (ed.mp[0].flags == MF_PRISTINE) {
            if (all_hue_identity(ed))
                return lutm ? run_stream<1024, MODE_DEVELOP_MPH, true, OUT_RGBA_F32, 0, 1>(cfg, d_cfa, raw, g, &edm, d_out, s)
                            : run_stream<256, MODE_DEVELOP_MPH, false, OUT_RGBA_F32, 0, 1>(cfg, d_cfa, raw, g, &ed, d_out, s);
            return lutm ? run_stream<1024, MODE_DEVELOP_MP, true, OUT_RGBA_F32, 0, 1>(cfg, d_cfa, raw, g, &edm, d_out, s)
                        : run_stream<256, MODE_DEVELOP_MP, false, OUT_RGBA_F32, 0, 1>(cfg, d_cfa, raw, g, &ed, d_out, s);
        }
        return lutm ? run_stream<1024, MODE_DEVELOP, true, OUT_RGBA_F32, 0, 1>(cfg, d_cfa, raw, g, &edm, d_out, s)
                    : run_stream<256, MODE_DEVELOP, false, OUT_RGBA_F32, 0, 1>(cfg, d_cfa, raw, g, &ed, d_out, s);
    }
    if (ed.mp[0].flags == MF_PRISTINE && cfg.variant != 12) {  // user masks over an untouched "main" (its own curve: the identity)
        const bool mph = all_hue_identity(ed) && cfg.variant != 11;
        if (lutm && !hp)
            return mph ? run_stream<1024, MODE_DEVELOP_MPH, true, OUTFMT>(cfg, d_cfa, raw, g, &edm, d_out, s)
                       : run_stream<1024, MODE_DEVELOP_MP, true, OUTFMT>(cfg, d_cfa, raw, g, &edm, d_out, s);
        if (mph) return run_fused<256, MODE_DEVELOP_MPH, false, OUTFMT>(hp, cfg, d_cfa, raw, g, &ed, d_out, s);
        return run_fused<256, MODE_DEVELOP_MP, false, OUTFMT>(hp, cfg, d_cfa, raw, g, &ed, d_out, s);
    }
    return lutm ? run_fused<1024, MODE_DEVELOP, true, OUTFMT>(hp, cfg, d_cfa, raw, g, &edm, d_out, s)
                : run_fused<256, MODE_DEVELOP, false, OUTFMT>(hp, cfg, d_cfa, raw, g, &ed, d_out, s);
}

template <int OUTFMT, int PK>
static hipError_t develop_packed_fmt(const LaunchCfg& cfg, const uint8_t* d_packed, PackedRows pk, const RawParamsDev& raw,
                                     const TileGeom& g, const DeviceEdit& ed, void* d_out, hipStream_t s)
{
    const uint16_t* in = reinterpret_cast<const uint16_t*>(d_packed);
    const bool lut = needs_lds_lut(ed);
    // RPF_PRECISION_FAST from packed rows (only "main", RGBA32F output; everything else computes the exact chain)
    if (OUTFMT == OUT_RGBA_F32 && ed.precision == 1u && ed.fast_tabs && ed.num_masks == 1 && cfg.variant == 0) {
        if (ed.mp[0].flags == MF_PRISTINE)
            return run_stream<256, MODE_DEVELOP_1P, false, OUT_RGBA_F32, PK, 1>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk);
        if (ed.mp[0].flags & MF_HUE_IDENTITY)
            return lut ? run_stream<1024, MODE_DEVELOP_1H, true, OUT_RGBA_F32, PK, 1>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk)
                       : run_stream<256, MODE_DEVELOP_1H, false, OUT_RGBA_F32, PK, 1>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk);
        if (ed.hue_fused)
            return lut ? run_stream<1024, MODE_DEVELOP_1, true, OUT_RGBA_F32, PK, 1>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk)
                       : run_stream<256, MODE_DEVELOP_1, false, OUT_RGBA_F32, PK, 1>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk);
    }
    if (ed.num_masks == 1) {
        if (ed.mp[0].flags == MF_PRISTINE)
            return run_stream<256, MODE_DEVELOP_1P, false, OUTFMT, PK>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk);
        if (ed.hue_fused)  // "main"'s hue / saturation / lightness curves folded into the hue table
            return lut ? run_stream<1024, MODE_DEVELOP_1, true, OUTFMT, PK>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk)
                       : run_stream<256, MODE_DEVELOP_1, false, OUTFMT, PK>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk);
        if (ed.mp[0].flags & MF_HUE_IDENTITY)
            return lut ? run_stream<1024, MODE_DEVELOP_1H, true, OUTFMT, PK>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk)
                       : run_stream<256, MODE_DEVELOP_1H, false, OUTFMT, PK>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk);
        return hipErrorInvalidValue;  // a non-identity hue curve without its table: the API layer builds it
    }
    if (!lut && ed.mp[0].flags == MF_PRISTINE) {
        if (all_hue_identity(ed))
            return run_stream<256, MODE_DEVELOP_MPH, false, OUTFMT, PK>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk);
        return run_stream<256, MODE_DEVELOP_MP, false, OUTFMT, PK>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk);
    }
    return lut ? run_stream<1024, MODE_DEVELOP, true, OUTFMT, PK>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk)
               : run_stream<256, MODE_DEVELOP, false, OUTFMT, PK>(cfg, in, raw, g, &ed, d_out, s, 0, 0, pk);
}

// d_packed: (rows + 4) packed rows, row r = frame row y0 - 2 + r, rows row_stride bytes apart (the band-buffer
// layout of launch_develop in packed form)
hipError_t launch_develop_packed(const LaunchCfg& cfg, const uint8_t* d_packed, uint32_t row_stride, uint32_t bits,
                                 uint32_t order, const RawParamsDev& raw, const TileGeom& g, const DeviceEdit& ed, void* d_out,
                                 int out_format, hipStream_t s)
{
    hipError_t e = check_geom(g, true);
    if (e == hipSuccess) e = check_edit(ed);
    if (e != hipSuccess) return e;
    if (!d_packed || !d_out) return hipErrorInvalidValue;
    if (bits != 10 && bits != 12 && bits != 14 && bits != 16) return hipErrorInvalidValue;
    if (order > 2u || (order == 2u && bits == 16)) return hipErrorInvalidValue;
    uint64_t need = ((uint64_t)g.width * bits + 7) / 8;
    if (order == 2u) {  // whole CSI-2 groups, and room for the kernel's 8-byte window in every row of the band
        const uint64_t n = bits == 12 ? 2 : 4;
        need = (g.width + n - 1) / n * (n + n * (bits - 8u) / 8u);
        if (need < 8) need = 8;
    }
    if (row_stride < need) return hipErrorInvalidValue;
    if ((uint64_t)row_stride * (g.rows + 4) > 0xffffffffull) return hipErrorInvalidValue;  // 32-bit row offsets
    if (ed.lens_strength != 0.0f || ed.flags != 0) return hipErrorInvalidValue;
    if (ed.num_masks > RPF_MASKS_PER_WORD) return hipErrorInvalidValue;  // see launch_develop
    const PackedRows pk{bits, row_stride, order};
    switch (out_format) {
    // the CSI-2 group layout is a compile-time variant of the packed row loads (as a run-time branch per sample it cost
    // the plain bit streams 7 %)
    case OUT_RGBA_F32: return order == 2u ? develop_packed_fmt<OUT_RGBA_F32, 2>(cfg, d_packed, pk, raw, g, ed, d_out, s)
                                          : develop_packed_fmt<OUT_RGBA_F32, 1>(cfg, d_packed, pk, raw, g, ed, d_out, s);
    case OUT_RGB_F32: return order == 2u ? develop_packed_fmt<OUT_RGB_F32, 2>(cfg, d_packed, pk, raw, g, ed, d_out, s)
                                         : develop_packed_fmt<OUT_RGB_F32, 1>(cfg, d_packed, pk, raw, g, ed, d_out, s);
    case OUT_RGB_U8: return order == 2u ? develop_packed_fmt<OUT_RGB_U8, 2>(cfg, d_packed, pk, raw, g, ed, d_out, s)
                                        : develop_packed_fmt<OUT_RGB_U8, 1>(cfg, d_packed, pk, raw, g, ed, d_out, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_develop(const LaunchCfg& cfg, const uint16_t* d_cfa, const RawParamsDev& raw, const TileGeom& g,
                          const DeviceEdit& ed, void* d_out, int out_format, hipStream_t s)
{
    hipError_t e = check_geom(g, true);
    if (e == hipSuccess) e = check_edit(ed);
    if (e != hipSuccess) return e;
    if (!d_cfa || !d_out) return hipErrorInvalidValue;
    if (ed.lens_strength != 0.0f) return hipErrorInvalidValue;  // needs the resampling pass (launch_apply)
    if (ed.flags != 0) return hipErrorInvalidValue;             // the Godot-legacy flavour lives in k_apply only
    // more masks than one coverage word holds: demosaic, then the RGBA pass (launch_apply) — the caller's two launches
    if (ed.num_masks > RPF_MASKS_PER_WORD) return hipErrorInvalidValue;
    switch (out_format) {
    case OUT_RGBA_F32: return develop_fmt<OUT_RGBA_F32>(cfg, d_cfa, raw, g, ed, d_out, s);
    case OUT_RGB_F32: return develop_fmt<OUT_RGB_F32>(cfg, d_cfa, raw, g, ed, d_out, s);
    case OUT_RGB_U8: return develop_fmt<OUT_RGB_U8>(cfg, d_cfa, raw, g, ed, d_out, s);
    default: return hipErrorInvalidValue;
    }
}

#endif  // RPF_TU != 2

#if RPF_TU != 1
// k_apply<SINGLE = 7> stages every mask's parameters in LDS: 48 B each behind the 28-KiB tables — 1024 masks are 48 KiB, and
// a CU still runs four 256-thread workgroups.  Editors with more read them from device memory (SINGLE = 8).
constexpr uint32_t WIDE_LDS_MASKS = 1024;
template <bool LDS_LUT, bool LENS, int OUTFMT, int SINGLE, int PREC = 0>
static hipError_t run_apply(const LaunchCfg& cfg, const ApplyArgs& a, hipStream_t s)
{
    constexpr int NT = LDS_LUT ? 1024 : 256;
    // (SINGLE == 7: every mask's parameters, in steps of 64 masks so that few sizes occur)
    const size_t lds = TabsOf<PREC>::BYTES + (LDS_LUT ? (size_t)RPF_CURVE_RESOLUTION * 2 : 0) +
                       (SINGLE == 0 || SINGLE == 6 ? sizeof(MaskParams) * RPF_MASKS_PER_WORD : 0) +
                       (SINGLE == 7 ? sizeof(MaskParams) * ((a.ed.num_masks + 63u) / 64u * 64u) : 0);
    auto kernel = k_apply<NT, LDS_LUT, LENS, OUTFMT, SINGLE, PREC>;
    hipError_t e = set_lds(kernel, lds);
    if (e != hipSuccess) return e;
    const size_t npix = (size_t)a.g.rows * a.g.width;
    // persistent grid-stride workgroups: exactly as many as are resident (registers and the 28-KiB table copy in
    // LDS decide; one more per CU would run as a second, mostly empty round)
    static std::atomic<int> resident_cache;
    int resident = SINGLE == 7 ? 0 : resident_cache.load(std::memory_order_relaxed);  // (7: the LDS size varies with the launch)
    if (resident == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kernel, NT, lds) != hipSuccess || resident < 1) resident = 1;
        if (SINGLE != 7) resident_cache.store(resident, std::memory_order_relaxed);
    }
    size_t grid = (size_t)cfg.num_cus * (size_t)resident;
    const size_t need = (npix + NT * 2 - 1) / (NT * 2);
    if (grid > need) grid = need;
    hipLaunchKernelGGL(kernel, dim3((uint32_t)grid), dim3(NT), lds, s, a);
    return hipGetLastError();
}

template <int OUTFMT>
static hipError_t apply_fmt(const LaunchCfg& cfg, const ApplyArgs& a0, hipStream_t s)
{
    // launches with user masks whose "main" has no brightness curve stage the first user mask's that has one (up to 8 masks;
    // variants 10 and 23, A/B: off) — as the fused develop launches do (develop_fmt)
    // more masks than one coverage word holds: the general chain over further planes (exact arithmetic in either precision)
    // (variant 24, A/B and the test of it: the parameters from device memory whatever the number of masks)
    if (a0.ed.num_masks > RPF_MASKS_PER_WORD) {
        const bool lens7 = a0.ed.lens_strength != 0.0f;
        if (a0.ed.num_masks <= WIDE_LDS_MASKS && cfg.variant != 24)
            return lens7 ? run_apply<false, true, OUTFMT, 7>(cfg, a0, s) : run_apply<false, false, OUTFMT, 7>(cfg, a0, s);
        return lens7 ? run_apply<false, true, OUTFMT, 8>(cfg, a0, s) : run_apply<false, false, OUTFMT, 8>(cfg, a0, s);
    }
    ApplyArgs a = a0;
    bool lut = needs_lds_lut(a.ed);
    if (!lut && a.ed.num_masks > 1 && a.ed.num_masks <= 8 && cfg.variant != 10 && cfg.variant != 23)
        for (uint32_t m = 1; m < a.ed.num_masks && !lut; ++m)
            if (!(a.ed.mp[m].flags & MF_BRIGHT_IDENTITY)) { a.ed.lds_bright_mask = m; lut = true; }
    const bool lens = a.ed.lens_strength != 0.0f;
    // RPF_PRECISION_FAST: the reference's own pass with the binary32-only chain (rpf_fast.h) — "main" alone, no lens
    // resample, the active shader flavour; every other launch computes the exact chain
    if (a.ed.precision == 1u && a.ed.fast_tabs && a.ed.num_masks == 1 && !lens && a.ed.flags == 0 && cfg.variant == 0) {
        if (a.ed.mp[0].flags == MF_PRISTINE) return run_apply<false, false, OUTFMT, 3, 1>(cfg, a, s);
        if (a.ed.mp[0].flags & MF_HUE_IDENTITY)
            return lut ? run_apply<true, false, OUTFMT, 5, 1>(cfg, a, s) : run_apply<false, false, OUTFMT, 5, 1>(cfg, a, s);
        if (!a.ed.hue_fused) return hipErrorInvalidValue;  // a hue curve without its table: the API layer builds it
        return lut ? run_apply<true, false, OUTFMT, 1, 1>(cfg, a, s) : run_apply<false, false, OUTFMT, 1, 1>(cfg, a, s);
    }
    // ... and with user masks (RGBA32F output): the rotation form when no mask has a hue curve, else the builtin route
    if (OUTFMT == OUT_RGBA_F32 && a.ed.precision == 1u && a.ed.fast_tabs && a.ed.num_masks > 1 && !lens && a.ed.flags == 0 && cfg.variant == 0) {
        if (all_hue_identity(a.ed))
            return lut ? run_apply<true, false, OUT_RGBA_F32, 6, 1>(cfg, a, s) : run_apply<false, false, OUT_RGBA_F32, 6, 1>(cfg, a, s);
        return lut ? run_apply<true, false, OUT_RGBA_F32, 0, 1>(cfg, a, s) : run_apply<false, false, OUT_RGBA_F32, 0, 1>(cfg, a, s);
    }
    if (a.ed.num_masks == 1) {  // "main" only (the common case): per-mask loops and mask-plane reads compiled out
        // (HUE_DIRECT, SINGLE == 2, is not used here: with this kernel's 16-B/px input stream the 32-byte table read
        // costs more than sincos_hue saves — D 0.234 -> 0.249 ms)
        if (lens) return lut ? run_apply<true, true, OUTFMT, 1>(cfg, a, s) : run_apply<false, true, OUTFMT, 1>(cfg, a, s);
        if (a.ed.mp[0].flags == MF_PRISTINE && a.ed.flags == 0) return run_apply<false, false, OUTFMT, 3>(cfg, a, s);
        // hue / saturation / lightness curves that are not all trivial: through the editor's fused hue table
        if (a.ed.hue_fused)
            return lut ? run_apply<true, false, OUTFMT, 4>(cfg, a, s) : run_apply<false, false, OUTFMT, 4>(cfg, a, s);
        return lut ? run_apply<true, false, OUTFMT, 1>(cfg, a, s) : run_apply<false, false, OUTFMT, 1>(cfg, a, s);
    }
    if (lens) return lut ? run_apply<true, true, OUTFMT, 0>(cfg, a, s) : run_apply<false, true, OUTFMT, 0>(cfg, a, s);
    return lut ? run_apply<true, false, OUTFMT, 0>(cfg, a, s) : run_apply<false, false, OUTFMT, 0>(cfg, a, s);
}

hipError_t launch_apply(const LaunchCfg& cfg, const float* d_in, const TileGeom& g, const DeviceEdit& ed, void* d_out,
                        int out_format, hipStream_t s)
{
    hipError_t e = check_geom(g, false);
    if (e == hipSuccess) e = check_edit(ed);
    if (e != hipSuccess) return e;
    if (!d_in || !d_out) return hipErrorInvalidValue;
    // the resampling branch reads anywhere in the frame: the tile must be the whole frame
    if (ed.lens_strength != 0.0f && (g.y0 != 0 || g.rows != g.height)) return hipErrorInvalidValue;
    if (ed.lens_strength != 0.0f && d_in == d_out) return hipErrorInvalidValue;
    if ((ed.flags & EF_DECODE_INPUT) && !ed.legacy) return hipErrorInvalidValue;
    ApplyArgs a;
    memset(&a, 0, sizeof(a));
    a.in = reinterpret_cast<const float4*>(d_in); a.g = g; a.out = d_out; a.ed = ed;
    switch (out_format) {
    case OUT_RGBA_F32: return apply_fmt<OUT_RGBA_F32>(cfg, a, s);
    case OUT_RGB_F32: return apply_fmt<OUT_RGB_F32>(cfg, a, s);
    case OUT_RGB_U8: return apply_fmt<OUT_RGB_U8>(cfg, a, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_unpack(const uint8_t* d_packed, size_t row_stride, uint32_t width, uint32_t rows, uint32_t bits,
                         uint32_t order, uint16_t* d_out, hipStream_t s)
{
    if (width == 0 || rows == 0) return hipSuccess;
    if (!d_packed || !d_out) return hipErrorInvalidValue;
    if (order > 2u) return hipErrorInvalidValue;
    const uint32_t gpr = (width + 15) / 16;
    const size_t total = (size_t)gpr * rows;
    const uint32_t grid = (uint32_t)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    if (order == 2u) {  // MIPI CSI-2: whole groups of 4 / 2 / 4 pixels in 5 / 3 / 7 bytes
        if (bits != 10 && bits != 12 && bits != 14) return hipErrorInvalidValue;
        const uint32_t n = bits == 12 ? 2u : 4u;
        if (row_stride < (size_t)((width + n - 1) / n) * (n + n * (bits - 8u) / 8u)) return hipErrorInvalidValue;
        if (bits == 10) hipLaunchKernelGGL((k_unpack_mipi<10>), dim3(grid), dim3(256), 0, s, d_packed, row_stride, width, rows, d_out, gpr);
        else if (bits == 12) hipLaunchKernelGGL((k_unpack_mipi<12>), dim3(grid), dim3(256), 0, s, d_packed, row_stride, width, rows, d_out, gpr);
        else hipLaunchKernelGGL((k_unpack_mipi<14>), dim3(grid), dim3(256), 0, s, d_packed, row_stride, width, rows, d_out, gpr);
        return hipGetLastError();
    }
    if (row_stride < ((size_t)width * bits + 7) / 8) return hipErrorInvalidValue;
    switch (bits) {
    case 10:
        if (order) hipLaunchKernelGGL((k_unpack<10, true>), dim3(grid), dim3(256), 0, s, d_packed, row_stride, width, rows, d_out, gpr);
        else hipLaunchKernelGGL((k_unpack<10, false>), dim3(grid), dim3(256), 0, s, d_packed, row_stride, width, rows, d_out, gpr);
        break;
    case 12:
        if (order) hipLaunchKernelGGL((k_unpack<12, true>), dim3(grid), dim3(256), 0, s, d_packed, row_stride, width, rows, d_out, gpr);
        else hipLaunchKernelGGL((k_unpack<12, false>), dim3(grid), dim3(256), 0, s, d_packed, row_stride, width, rows, d_out, gpr);
        break;
    case 14:
        if (order) hipLaunchKernelGGL((k_unpack<14, true>), dim3(grid), dim3(256), 0, s, d_packed, row_stride, width, rows, d_out, gpr);
        else hipLaunchKernelGGL((k_unpack<14, false>), dim3(grid), dim3(256), 0, s, d_packed, row_stride, width, rows, d_out, gpr);
        break;
    case 16:
        if (order) hipLaunchKernelGGL((k_unpack<16, true>), dim3(grid), dim3(256), 0, s, d_packed, row_stride, width, rows, d_out, gpr);
        else hipLaunchKernelGGL((k_unpack<16, false>), dim3(grid), dim3(256), 0, s, d_packed, row_stride, width, rows, d_out, gpr);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// The fused hue table of an editor with one mask (rpf_math.h HueFused): for every hue index j the boundary direction (from
// the processor's table) and what the chain takes from "main"'s curves at j — cos / sin of the hue curve[j] maps to, the
// saturation and lightness gains — computed with the chain's own functions (luts.hsl's quotients, sincos_hue), so a launch
// that reads them has the bits of one that evaluates them.
__global__ __launch_bounds__(256) void k_build_hue_fused(const HueBound* hb, const ushort4* hsl, MaskParams P, const MathTables* tabs,
                                                         HueFused* table)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;  // 65536 threads
    const ushort4 v = hsl[j];
    table[j] = hue_fused_entry(hb[j], v.x, v.y, v.z, P, tabs->sc);
}

hipError_t launch_build_hue_fused(const HueBound* d_hue_bounds, const ushort4* d_hsl_main, const MaskParams& main,
                                  const MathTables* d_tabs, HueFused* d_table, hipStream_t s)
{
    if (!d_hue_bounds || !d_hsl_main || !d_tabs || !d_table) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_build_hue_fused, dim3(256), dim3(256), 0, s, d_hue_bounds, d_hsl_main, main, d_tabs, d_table);
    return hipGetLastError();
}

// The vignette's falloff of every pixel of a tile (DeviceEdit::vign_falloff), global coordinates (wgsl:169).
__global__ __launch_bounds__(256) void k_vign_falloff(TileGeom g, float* map)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, r = blockIdx.y;
    if (x >= g.width) return;
    map[(size_t)r * g.width + x] = vignette_falloff_from_axes(vignette_axis2((int)x, (float)g.width),
                                                             vignette_axis2((int)(g.y0 + r), (float)g.height));
}

hipError_t launch_vignette_falloff(const TileGeom& g, float* d_map, hipStream_t s)
{
    if (!d_map || g.width == 0 || g.rows == 0 || g.rows > 65535u * 16u) return hipErrorInvalidValue;
    // (grid.y is limited to 65535: taller tiles in slabs)
    for (uint32_t r0 = 0; r0 < g.rows; r0 += 65535u) {
        const uint32_t nr = g.rows - r0 < 65535u ? g.rows - r0 : 65535u;
        const TileGeom slab{g.width, g.height, g.y0 + r0, nr};
        hipLaunchKernelGGL(k_vign_falloff, dim3((g.width + 255u) / 256u, nr), dim3(256), 0, s, slab, d_map + (size_t)r0 * g.width);
    }
    return hipGetLastError();
}

hipError_t launch_touch(const void* d_buf, size_t bytes, int num_cus, hipStream_t s)
{
    if (bytes == 0) return hipSuccess;
    if (!d_buf) return hipErrorInvalidValue;
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_buf);
    const uintptr_t lo = (a + 15) & ~(uintptr_t)15, hi = (a + bytes) & ~(uintptr_t)15;  // whole 16-byte words inside
    if (hi <= lo) return hipSuccess;
    const size_t n16 = (hi - lo) / 16;
    const size_t want = (n16 + 255) / 256;
    const uint32_t grid = (uint32_t)(want < (size_t)num_cus * 8 ? want : (size_t)num_cus * 8);
    hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, reinterpret_cast<const uint4*>(lo), n16);
    return hipGetLastError();
}

hipError_t launch_resize(const float* d_src, uint32_t sw, uint32_t sh, float* d_dst, uint32_t dw, uint32_t dh, hipStream_t s)
{
    if (!d_src || !d_dst || sw == 0 || sh == 0 || dw == 0 || dh == 0) return hipErrorInvalidValue;
    const size_t n = (size_t)dw * dh;
    const uint32_t grid = (uint32_t)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
    hipLaunchKernelGGL(k_resize, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(d_src), sw, sh,
                       reinterpret_cast<float4*>(d_dst), dw, dh);
    return hipGetLastError();
}

hipError_t launch_repack(const float* d_rgba, size_t npix, void* d_out, int out_format, hipStream_t s)
{
    if (npix == 0) return hipSuccess;
    if (!d_rgba || !d_out) return hipErrorInvalidValue;
    const uint32_t grid = (uint32_t)((npix + 255) / 256 > 4096 ? 4096 : (npix + 255) / 256);
    if (out_format == OUT_RGB_F32)
        hipLaunchKernelGGL(k_repack<OUT_RGB_F32>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(d_rgba),
                           npix, d_out);
    else if (out_format == OUT_RGB_U8)
        hipLaunchKernelGGL(k_repack<OUT_RGB_U8>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(d_rgba),
                           npix, d_out);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_combine_masks(const uint8_t* d_planes, uint32_t planes, size_t npix, uint16_t* d_bits, hipStream_t s)
{
    if (npix == 0 || planes == 0) return hipSuccess;
    if (!d_planes || !d_bits || planes > 16) return hipErrorInvalidValue;
    const uint32_t grid = (uint32_t)((npix + 255) / 256 > 8192 ? 8192 : (npix + 255) / 256);
    hipLaunchKernelGGL(k_combine_masks, dim3(grid), dim3(256), 0, s, d_planes, planes, npix, d_bits);
    return hipGetLastError();
}

hipError_t launch_pad_rgb(const float* d_rgb, size_t npix, float* d_rgba, hipStream_t s)
{
    if (npix == 0) return hipSuccess;
    if (!d_rgb || !d_rgba) return hipErrorInvalidValue;
    const uint32_t grid = (uint32_t)((npix + 255) / 256 > 4096 ? 4096 : (npix + 255) / 256);
    hipLaunchKernelGGL(k_pad_rgb, dim3(grid), dim3(256), 0, s, d_rgb, npix, reinterpret_cast<float4*>(d_rgba));
    return hipGetLastError();
}

hipError_t launch_rgb_int(const void* d_in, int bits, size_t npix, const float* d_lut, float* d_rgba, hipStream_t s)
{
    if (npix == 0) return hipSuccess;
    if (!d_in || !d_rgba || (bits != 8 && bits != 16)) return hipErrorInvalidValue;
    if (bits == 16 && (reinterpret_cast<uintptr_t>(d_in) & 1)) return hipErrorInvalidValue;
    if (npix >= (1ull << 28)) return hipErrorInvalidValue;  // the output is addressed through 32-bit buffer offsets
    const uint32_t grid = (uint32_t)((npix + 255) / 256 > 16384 ? 16384 : (npix + 255) / 256);
    if (bits == 8)
        hipLaunchKernelGGL((k_rgb_int<uint8_t>), dim3(grid), dim3(256), 0, s, static_cast<const uint8_t*>(d_in), npix, d_lut,
                           reinterpret_cast<float4*>(d_rgba));
    else
        hipLaunchKernelGGL((k_rgb_int<uint16_t>), dim3(grid), dim3(256), 0, s, static_cast<const uint16_t*>(d_in), npix, d_lut,
                           reinterpret_cast<float4*>(d_rgba));
    return hipGetLastError();
}

hipError_t launch_rgb16be(const uint8_t* d_in, size_t npix, float* d_rgba, hipStream_t s)
{
    if (npix == 0) return hipSuccess;
    if (!d_in || !d_rgba || (reinterpret_cast<uintptr_t>(d_in) & 1)) return hipErrorInvalidValue;
    const uint32_t grid = (uint32_t)((npix + 255) / 256 > 8192 ? 8192 : (npix + 255) / 256);
    hipLaunchKernelGGL(k_rgb16be, dim3(grid), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(d_in), npix,
                       reinterpret_cast<float4*>(d_rgba));
    return hipGetLastError();
}

#endif  // RPF_TU != 1

}  // namespace rpf
