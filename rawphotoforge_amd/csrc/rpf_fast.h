// rpf_fast.h — the transcendental builtins of the edit chain in BINARY32 ONLY, each within 1 ulp of the correctly rounded
// value (RPF_PRECISION_FAST, include/rpf.h).  The default kernels evaluate every builtin in binary64 and round once
// (rpf_math.h): bit-identical to the parity oracle, and about half of their instruction stream.  WGSL itself asks far less of
// pow / atan2 / sin / cos (rust/photo-editor/src/wgpu_shader.wgsl:67, 72, 78-79, 98 run at whatever accuracy the backend's
// builtins have); this header is the other end of that trade: no binary64 instruction, no table outside LDS, no L2 gather.
//
//   pow(x, f32(1/3)), pow(x, f32(1/2.4))   one 16-byte LDS entry per (exponent, top five mantissa bits), a cubic, two
//                                          fmas: < 0.6 ulp from the true value (the stored value carries its own rounding
//                                          residue), checked against the oracle's cr_powf on EVERY binary32 of the chain's
//                                          domains (tests/test_fast_math.py, on the CPU: IEEE operations only)
//   cos / sin(f32(k / 65535) * 2 * pi_f)   Cody-Waite reduction of the reference's own binary32 angle to [-pi/4, pi/4]
//                                          (exact), two short polynomials: <= 1 ulp for all 65 536 k (same test file)
//   sqrt                                   v_sqrt_f32 (1 ulp; the host build is sqrtf)
//   u32(h * 65535), h = atan2(b, a) / 2 pi an eight-term polynomial of min / max straight on the index scale (error 0.0004
//                                          index units + binary32 rounding at 65 535: 0.004): equals the reference's index
//                                          except within that distance of a boundary — the reference's own binary32
//                                          pipeline (atan2, / f32(2 pi), + 1, * 65535) is no better conditioned there;
//                                          the flip rate is measured and asserted (tests/test_fast_math.py, test_gpu_fast.py)
//
// Everything except the hardware reciprocal / square root is IEEE arithmetic, so the host twin (tests/twin) computes the
// device's bits for the pows and the sine / cosine.
#pragma once
#include "rpf_math.h"

namespace rpf {

// ---------------------------------------------------------------------------
// pow(x, c) for the two exponents of the chain
//
// x = 2^e m.  Entry (e, j), j = the top five bits of m's fraction, holds r = 2^-e r0 with r0 ~ 1 / (bin centre), T = RN(V)
// and rho = RN((V - T) / T) for V = 2^(e c) r0^(-c) in long double, so V = T (1 + rho) to 2^-48.  With t = x r - 1 = m r0 - 1
// (one fma on x itself — no exponent or significand is ever extracted: |t| <= 2^-6, rounded at 2^-30) the result is
//     x^c = V (1 + t)^c = T (1 + q),   q = rho + t (c1 + t (c2 + t c3)),   y = fma(T, q, T),
// c_k = binomial(c, k); the dropped c4 t^4 is below 2^-28.6 and the rounding of q (|q| < 2^-7) below 2^-31: the single
// rounding of the last fma is all that is left, < 0.6 ulp in total.  The exponent never meets an arithmetic operation:
// the entry's T carries 2^(e c) exactly as the long-double evaluation produced it.
// The index is the binary32 pattern shifted (exponent and five fraction bits in one field), minus the first row's, clamped
// into the table: inputs below the first row read row 0, inputs above the last row its last entry.
//   cube root   rows for e = -22 .. 0; row e = -22 is all zeros, so 0, subnormals and anything below 2^-21 return 0 (the
//               chain's cube-root arguments are 0 or >= 7e-7 and <= 1.0000001, wgsl:65-67)
//   gamma       rows for e = -9 .. 2 (srgb_upper_fast below: the table carries the whole upper branch of linear_to_srgb);
//               below 2^-9 the result is unspecified (finite): the chain takes the linear branch there (wgsl:99-102); at
//               and above 8 the result is >= 1.9, which the chain's clamp turns into 1 like the exact value (wgsl:335)
// ---------------------------------------------------------------------------
struct FastPowEntry {
    float r, T, rho, pad;
};
constexpr int FAST_POW_J = 32;  // bins per binade
template <int KIND> struct FastPowGeom;
template <> struct FastPowGeom<POW_CBRT> { static constexpr int E0 = -22, ROWS = 23; };
template <> struct FastPowGeom<POW_GAMMA> { static constexpr int E0 = -9, ROWS = 12; };

struct FastTables {
    FastPowEntry cbrt[FastPowGeom<POW_CBRT>::ROWS * FAST_POW_J];
    FastPowEntry gamma[FastPowGeom<POW_GAMMA>::ROWS * FAST_POW_J];
};

// binomial(c, k) as binary32 (tools/gen_math_consts.py prints the binary64 values; these are their roundings)
template <int KIND> struct FastPowCoef;
template <> struct FastPowCoef<POW_CBRT> {
    static constexpr float c1 = (float)PowCoef<POW_CBRT>::c1, c2 = (float)PowCoef<POW_CBRT>::c2, c3 = (float)PowCoef<POW_CBRT>::c3;
};
template <> struct FastPowCoef<POW_GAMMA> {
    static constexpr float c1 = (float)PowCoef<POW_GAMMA>::c1, c2 = (float)PowCoef<POW_GAMMA>::c2, c3 = (float)PowCoef<POW_GAMMA>::c3;
};

// the byte offset of x's entry inside its table
template <int KIND, class TabT>
RPF_HD uint32_t fast_pow_offset(float x, const TabT&)
{
    using G = FastPowGeom<KIND>;
    constexpr int32_t BASE16 = (int32_t)((uint32_t)(127 + G::E0) * FAST_POW_J * 16u);
    constexpr int32_t MAX16 = (int32_t)((G::ROWS * FAST_POW_J - 1) * 16);
    // (exponent, top five fraction bits) * 16; the shift is arithmetic, so a negative x (sign bit set) ends up below row 0
    const int32_t off = (((int32_t)f2bits(x) >> 14) & ~15) - BASE16;
    return (uint32_t)(off < 0 ? 0 : (off > MAX16 ? MAX16 : off));  // one v_med3_i32; a negative x has a negative `off`: row 0
}

// the table in ordinary memory (host twin) ...
RPF_HD FastPowEntry fast_pow_read(const FastPowEntry* tab, uint32_t off)
{
    return *reinterpret_cast<const FastPowEntry*>(reinterpret_cast<const char*>(tab) + off);
}
template <int KIND> RPF_HD float fast_pow_c3(const FastPowEntry*) { return FastPowCoef<KIND>::c3; }
#if defined(__HIPCC__)
// ... or staged in LDS, named by its LDS byte address (see rpf_math.h LdsPowTable), with the cubic's leading coefficient
// in a vector register of the caller's (the cubic's first fma has two constants; an instruction carries one)
struct LdsFastPow {
    uint32_t base;
    float c3;
    // offset = clamp(shifted pattern - sub, lo, hi), address = base + offset.  Plainly: sub = the first row's shifted pattern,
    // lo = 0, hi = the last entry.  BIASED (fast_pow_biased): a table staged at an LDS address not below its first row's
    // shifted pattern keeps the pattern as it is — sub = 0, the clamp's bounds move instead and base = address - first row
    // goes into the read's immediate offset field: one instruction less per evaluation.  All five are literals after inlining.
    int32_t sub, lo, hi;
};
template <int KIND>
__device__ __forceinline__ LdsFastPow fast_pow_plain(uint32_t lds_addr, float c3)
{
    using G = FastPowGeom<KIND>;
    return LdsFastPow{lds_addr, c3, (int32_t)((uint32_t)(127 + G::E0) * FAST_POW_J * 16u), 0, (G::ROWS * FAST_POW_J - 1) * 16};
}
template <int KIND>
__device__ __forceinline__ LdsFastPow fast_pow_biased(uint32_t lds_addr, float c3)  // lds_addr >= the first row's shifted pattern
{
    using G = FastPowGeom<KIND>;
    const int32_t first = (int32_t)((uint32_t)(127 + G::E0) * FAST_POW_J * 16u);
    return LdsFastPow{lds_addr - (uint32_t)first, c3, 0, first, first + (G::ROWS * FAST_POW_J - 1) * 16};
}
__device__ __forceinline__ FastPowEntry fast_pow_read(const LdsFastPow& T, uint32_t off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = *(const __attribute__((address_space(3))) v4f*)(T.base + off);
    return FastPowEntry{v.x, v.y, v.z, v.w};
#else
    (void)T; (void)off;
    return FastPowEntry{0.0f, 0.0f, 0.0f, 0.0f};  // host pass of the HIP compilation: never called
#endif
}
template <int KIND>
__device__ __forceinline__ uint32_t fast_pow_offset(float x, const LdsFastPow& T)
{
    const int32_t off = (((int32_t)f2bits(x) >> 14) & ~15) - T.sub;
    return (uint32_t)(off < T.lo ? T.lo : (off > T.hi ? T.hi : off));
}
template <int KIND> __device__ __forceinline__ float fast_pow_c3(const LdsFastPow& T) { return T.c3; }
#endif

template <int KIND, class TabT>
RPF_HD float powc_fast(float x, const TabT& tab)
{
    using K = FastPowCoef<KIND>;
    const FastPowEntry e = fast_pow_read(tab, fast_pow_offset<KIND>(x, tab));
    const float t = __builtin_fmaf(x, e.r, -1.0f);  // the entry's r carries 2^-e: x r = m / (bin centre)
    float p = __builtin_fmaf(t, fast_pow_c3<KIND>(tab), K::c2);
    p = __builtin_fmaf(t, p, K::c1);
    const float q = __builtin_fmaf(t, p, e.rho);
    return __builtin_fmaf(e.T, q, e.T);
}

// linear_to_srgb's upper branch, 1.055 pow(x, 1 / 2.4) - 0.055 (wgsl:98-100), as ONE evaluation of the gamma table: its
// entries hold T = RN(1.055 V), pad = T2 = RN(W) and rho = RN((W - T2) / T) for W = 1.055 V - 0.055, so that the branch is
// T2 + T q with the same q as a pow — one fma (which also carries the chain's clamp) instead of a pow's last fma and a
// multiply-add.  One rounding for the whole branch: within 1 ulp of its true value on every binary32 of [2^-9, 8)
// (tests/test_fast_math.py; the reference's three roundings — pow, times 1.055, minus 0.055 — stray up to 1.5 ulp from it).
// The gamma table is therefore not a pow table: powc_fast is for the cube root.
template <class TabT>
RPF_HD float srgb_upper_fast(float x, const TabT& tab)
{
    using K = FastPowCoef<POW_GAMMA>;
    const FastPowEntry e = fast_pow_read(tab, fast_pow_offset<POW_GAMMA>(x, tab));
    const float t = __builtin_fmaf(x, e.r, -1.0f);
    float p = __builtin_fmaf(t, fast_pow_c3<POW_GAMMA>(tab), K::c2);
    p = __builtin_fmaf(t, p, K::c1);
    const float q = __builtin_fmaf(t, p, e.rho);
    return __builtin_fmaf(e.T, q, e.pad);
}

// ---------------------------------------------------------------------------
// cos(a), sin(a) for a = f32(f32(k) / 65535) * 2 * pi_f as the reference forms it (wgsl:78-79 after the hue curve):
// `kf` is the integer k held as binary32.  n = nearest integer of a * 2 / pi (the 1.5 * 2^23 trick: n sits in the low
// bits of the sum); r = a - n pi/2 in three fmas (pi/2 = 1.5703125 + an eleven-bit term + the rest: the first two products
// and differences are exact, the third rounds once — a = f32(2 pi) leaves r = 1.7e-7, which two terms get wrong in the
// fourth digit); sin r = r + r^3 S(r^2), cos r = 1 + r^2 (-1/2 + r^2 C(r^2)) with minimax S, C (relative error 4e-9 / 1e-10
// before rounding, tools/gen_math_consts.py); the quadrant swaps and signs.
// ---------------------------------------------------------------------------
RPF_HD void sincos_hue_fast(float kf, float& cs, float& sn)
{
    const float h = div_u16f_by(kf, 65535.0f, (float)(1.0 / 65535.0));
    const float a = (h * 2.0f) * RPF_K_PI_F;
    const float big = __builtin_fmaf(a, 0.6366197466850281f, 12582912.0f);
    const float n = big - 12582912.0f;
    float r = __builtin_fmaf(-n, 1.5703125f, a);               // exact
    r = __builtin_fmaf(-n, 0.0004837512969970703f, r);          // exact (eleven significant bits)
    r = __builtin_fmaf(-n, 7.549790126404332e-08f, r);          // rounds once
    const float z = r * r;
    float sp = __builtin_fmaf(z, -0.00019506941316649318f, 0.00833209976553917f);
    sp = __builtin_fmaf(z, sp, -0.16666653752326965f);
    const float s = __builtin_fmaf(r * z, sp, r);
    float cp = __builtin_fmaf(z, 2.4423623472102918e-05f, -0.001388723379932344f);
    cp = __builtin_fmaf(z, cp, 0.04166664555668831f);
    cp = __builtin_fmaf(z, cp, -0.5f);
    const float c = __builtin_fmaf(z, cp, 1.0f);
    const uint32_t q = f2bits(big);  // low bits: n in 0 .. 4
    const bool swap = (q & 1u) != 0u;
    const uint32_t cb = f2bits(swap ? s : c) ^ (((q + 1u) & 2u) << 30);  // cos: negative in quadrants 1, 2
    const uint32_t sb = f2bits(swap ? c : s) ^ ((q & 2u) << 30);          // sin: negative in quadrants 2, 3
    cs = bits2f(cb);
    sn = bits2f(sb);
}

// ---------------------------------------------------------------------------
// The position of the direction (x, y) on the hue-index scale, in [0, 65535]: atan2(y, x) / f32(2 pi), wrapped into
// [0, 1), times 65535 — what wgsl:72-73, 320 truncate to the curve index.  hue_candidate's arrangement (rpf_math.h) with
// an eight-term polynomial: 0.0004 index units from the true position before the binary32 roundings (0.004 at the top of
// the scale).  trunc() of it is the index.
// ---------------------------------------------------------------------------
RPF_HD float hue_position_fast(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
#if defined(__HIP_DEVICE_COMPILE__)
    float mx, mn;
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(mx) : "v"(x), "v"(y));
    asm("v_min_f32 %0, |%1|, |%2|" : "=v"(mn) : "v"(x), "v"(y));
    const float q = mn * __builtin_amdgcn_rcpf(mx + 0x1p-100f);  // (+ 2^-100: no change to a normal mx; a = b = 0 gives q = 0, not 0 * inf)
#else
    const float mx = ax > ay ? ax : ay;
    const float mn = ax > ay ? ay : ax;
    const float mxc = mx < 0x1p-100f ? 0x1p-100f : (mx > 0x1p+100f ? 0x1p+100f : mx);
    const float q = mn / mxc;
#endif
    const float z = q * q;
    float p = __builtin_fmaf(z, -42.290016174316406f, 228.035400390625f);
    p = __builtin_fmaf(z, p, -583.177734375f);
    p = __builtin_fmaf(z, p, 1005.7022705078125f);
    p = __builtin_fmaf(z, p, -1450.700439453125f);
    p = __builtin_fmaf(z, p, 2080.470458984375f);
    p = __builtin_fmaf(z, p, -3476.37744140625f);
    p = __builtin_fmaf(z, p, 10430.2119140625f);
    float v = p * q;
    v = ay > ax ? 16383.75f - v : v;   // pi/2 - a
    v = x < 0.0f ? 32767.5f - v : v;   // pi - a
    v = y < 0.0f ? 65535.0f - v : v;   // 2 pi - a (the wrap of wgsl:73)
    return v;
}

// The same position's FRACTIONAL part alone, i.e. how far the direction (x, y) lies past the boundary of its hue index, in
// index units in [0, 1): all a launch needs when every hue curve is the identity (edit_chain_fast's HUE_ROT).  The first
// octant's position p q is below 8192 (ulp 2^-11); its fraction is exact, and each reflection c - position acts on
// fractions: frac(c - v) = fract(frac(c) - frac(v)) with frac(c) in {0.75, 0.5, 0}.  Accurate to 0.0005 index units
// (5e-8 rad) all round the circle — the full position above loses that to the binary32 grid near 65535 (0.004).
RPF_HD float hue_fraction_fast(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
#if defined(__HIP_DEVICE_COMPILE__)
    float mx, mn;
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(mx) : "v"(x), "v"(y));
    asm("v_min_f32 %0, |%1|, |%2|" : "=v"(mn) : "v"(x), "v"(y));
    const float q = mn * __builtin_amdgcn_rcpf(mx + 0x1p-100f);  // (+ 2^-100: no change to a normal mx; a = b = 0 gives q = 0, not 0 * inf)
#else
    const float mx = ax > ay ? ax : ay;
    const float mn = ax > ay ? ay : ax;
    const float mxc = mx < 0x1p-100f ? 0x1p-100f : (mx > 0x1p+100f ? 0x1p+100f : mx);
    const float q = mn / mxc;
#endif
    const float z = q * q;
    float p = __builtin_fmaf(z, -42.290016174316406f, 228.035400390625f);
    p = __builtin_fmaf(z, p, -583.177734375f);
    p = __builtin_fmaf(z, p, 1005.7022705078125f);
    p = __builtin_fmaf(z, p, -1450.700439453125f);
    p = __builtin_fmaf(z, p, 2080.470458984375f);
    p = __builtin_fmaf(z, p, -3476.37744140625f);
    p = __builtin_fmaf(z, p, 10430.2119140625f);
    const float v = p * q;  // [0, 8191.875]
    auto fract = [](float t) {  // t - floor(t), in [0, 1)
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_fractf(t);
#else
        return t - __builtin_floorf(t);
#endif
    };
    float f = fract(v);
    f = ay > ax ? fract(0.75f - f) : f;  // 16383.75 - v
    // the two reflections about the axes in one: x < 0 maps f to frac(0.5 - f), y < 0 then to frac(-f); together
    // frac(s f + c) with s = -1 when exactly one of x, y is negative and c = 0.5 when x is — sign bits and one addition
    // (a zero of either sign lies ON the axis of its reflection, which leaves it where it is: -0 may count as negative)
    const uint32_t sx = f2bits(x) & 0x80000000u, sy = f2bits(y) & 0x80000000u;
    const float c = bits2f((uint32_t)((int32_t)f2bits(x) >> 31) & 0x3f000000u);  // 0.5f or 0
    return fract(bits2f(f2bits(f) ^ sx ^ sy) + c);
}

// 2 * f32(pi) / 65535 as binary32: one index step in radians
#define RPF_K_HUE_STEP_F ((float)RPF_K_HUE_STEP)

RPF_HD float sqrt_fast(float s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(s);  // v_sqrt_f32: 1 ulp; the chain's argument a^2 + b^2 is 0 or a normal number
#else
    return __builtin_sqrtf(s);
#endif
}

}  // namespace rpf
