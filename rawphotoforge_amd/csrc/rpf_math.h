// rpf_math.h — correctly rounded binary32 pow / atan2 / sincos for the edit chain,
// written with IEEE-754 operations only (binary64 fma/mul/add/ldexp, binary32 div), so a
// host build of this header computes bit-for-bit what the gfx950 kernels compute.
//
// Why: WGSL leaves pow/atan2/sin/cos accuracy to the backend
// (rust/photo-editor/src/wgpu_shader.wgsl:67,72,79,98); the parity oracle
// (oracle/rpf_oracle.c) fixes them as CORRECTLY ROUNDED.  Hardware v_log/v_exp/
// v_sin are 1-ulp-class approximations, and 1 ulp before a `u32(x * 65535)` LUT
// index (wgsl:293-295, 320-322) or before M1_INV (wgsl:46-50) becomes many ulp
// after it — so the kernels evaluate each transcendental in binary64 to < 2^-49
// relative and round once.  The two pow domains that matter are small enough to
// be checked exhaustively against the oracle (tests/test_math_exhaustive.py).
// Everything here is branch-free (selects only): no divergence, no calls.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RPF_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define RPF_HD inline
#endif

namespace rpf {

// ---------------------------------------------------------------------------
// Tables (built once on the host by build_math_tables, staged in LDS by kernels)
// ---------------------------------------------------------------------------

// x^c = 2^(e c) * r_j^(-c) * (1 + t)^c,  x = 2^e m,  t = m r_j - 1 (exact in binary64)
struct PowTable {
    double E[256];       // E[b] = 2^((b - 127) c) for the biased binary32 exponent b; E[0] = 0 (x = 0)
#ifndef RPF_POW_BITS
#define RPF_POW_BITS 9
#endif
    double TR[1 << RPF_POW_BITS][2];   // TR[j] = { r_j^(-c), r_j },  j = top 9 mantissa bits: |t| <= 2^-10, degree 4 suffices
};

// cos/sin of k * K_HUE_STEP split as k = 256 k1 + k0
struct SinCosTable {
    double hi[256][2];  // { cos, sin }(256 k1 * step)
    double lo[256][2];  // { cos, sin }(k0 * step)
};

struct MathTables {
    PowTable cbrt;   // c = f32(1/3)    wgsl:67
    PowTable gamma;  // c = f32(1/2.4)  wgsl:98
    SinCosTable sc;
};

// binomial(c, k), k = 1..5, for the two exponents (tools/gen_math_consts.py); compile-time
// constants so they live in SGPRs / inline operands instead of LDS
enum PowKind { POW_CBRT = 0, POW_GAMMA = 1, POW_DECODE = 2 };
template <int KIND> struct PowCoef;
template <> struct PowCoef<POW_CBRT> {
    static constexpr double c1 = 0x1.5555560000000p-2, c2 = -0x1.c71c7238e38e0p-4, c3 = 0x1.f9add40ca4583p-5,
                            c4 = -0x1.511e8d48b0fcap-5, c5 = 0x1.ee71136534be7p-6;
};
template <> struct PowCoef<POW_GAMMA> {
    static constexpr double c1 = 0x1.aaaaaa0000000p-2, c2 = -0x1.f1c71c38e38e0p-4, c3 = 0x1.06b74f00ca456p-4,
                            c4 = -0x1.535770cc3f359p-5, c5 = 0x1.e663bb5d82900p-6;
};

// c = f32(2.4): srgb_to_linear of the Godot-legacy shader flavour (rust-godot-legacy/.../wgpu_shader.wgsl:89-97)
template <> struct PowCoef<POW_DECODE> {
    static constexpr double c1 = 0x1.3333340000000p+1, c2 = 0x1.ae147deb85200p+0, c3 = 0x1.cac08d7b90194p-3,
                            c4 = -0x1.13405205bbf99p-5, c5 = 0x1.60525321e5f53p-7;
};

// Tables only the Godot-legacy flavour reads (from global memory; that flavour is not a throughput path)
struct LegacyTables {
    PowTable decode;  // c = f32(2.4)
};

// wgsl:78 `c.z * 2.0 * 3.14159265359`, wgsl:72 `2.0 * 3.14159265359`
#define RPF_K_PI_F ((float)3.14159265359)
#define RPF_K_TWO_PI_F ((float)(2.0 * 3.14159265359))

// 2 * f32(3.14159265359) / 65535 as binary64 (tools/gen_math_consts.py)
#define RPF_K_HUE_STEP 0x1.9221482148215p-14

RPF_HD uint64_t d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
RPF_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
RPF_HD uint32_t f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }
RPF_HD float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }
RPF_HD double hilo2d(uint32_t hi, uint32_t lo) { return u2d(((uint64_t)hi << 32) | lo); }

// f32(k) / d for integer k in [0, 65535] and d in {65535, 32767.5}: the reciprocal as an unevaluated sum of two
// binary32 values (1/65535 = 2^-16 (1 + 2^-16) + 2^-48 (1 + 2^-16) + ..., 1/32767.5 twice that), one product and
// one fma: RN(k rd_hi + RN(k rd_lo)).  Equal to the IEEE quotient for every k (checked exhaustively in
// tests/test_math_exhaustive.py; round 1 used Markstein's three-operation sequence).  wgsl:303-305, 328-330.
// `rd` is RN(1/d) as before (= rd_hi); rd_lo follows from it because both constants share the pattern above.
RPF_HD float div_u16f_by(float kf, float d, float rd)  // kf: an integer in [0, 65535] held as binary32
{
    (void)d;
    const float rd_lo = rd * 0x1p-32f;  // exact scaling, folded at compile time (rd is a literal at every call site)
    return __builtin_fmaf(kf, rd, kf * rd_lo);
}
RPF_HD float div_u16_by(uint32_t k, float d, float rd) { return div_u16f_by((float)k, d, rd); }

// x / d for a constant d: Markstein's sequence (q = RN(x rd), r = fma(-q, d, x) exact,
// q' = RN(q + r rd)) is the correctly rounded quotient when rd = RN(1/d) and d's significand
// is not all ones — true for d = f32(2 pi) (wgsl:72) — as long as nothing underflows; below
// 2^-100 the IEEE division is used instead (a rarely taken, wave-uniformly skipped branch).
// Equal to IEEE division on the whole range of atan2 results (tests/test_math_exhaustive.py).
RPF_HD float div_by_const(float x, float d, float rd)
{
    const float q0 = x * rd;
    const float e = __builtin_fmaf(-q0, d, x);
    float q = __builtin_fmaf(e, rd, q0);
    if (__builtin_expect(__builtin_fabsf(x) < 0x1p-100f, 0)) q = x / d;
    return q;
}

// Correctly rounded x^c for normal x >= 0 (c fixed by KIND).  Exhaustively equal to the oracle's cr_powf on
// [2^-21, 1.25) for cbrt, [2^-9, 2^3) for gamma and [2^-4, 2) for the legacy decode; same algorithm (rel. error
// < 2^-50 before the single rounding) on every other normal binary32.
// The exponent and the table indices are taken straight from the binary32 bit pattern with shifts and masks
// (full-rate on gfx950; the v_bfe / v_lshl_add forms are half-rate), the sign bit is masked off (a negative x
// returns |x|^c: the chain never uses that value — wgsl:67 takes max(x, 0), wgsl:98 the linear branch).
// x = 0 has exponent field 0, whose table entry is 0.0, so pow(0, c) = 0 falls out of the same multiplications
// with no select; a SUBNORMAL x also returns 0.  The chain cannot produce one where it matters: cube-root
// inputs are 0 or >= 7e-7 (M1 times channels that are multiples of 1/65535), and below 0.0031308 the gamma
// stage takes its linear branch.
// The two table reads of powc_d, by byte offset.  A PowTable in ordinary memory (host, global) ...
RPF_HD void pow_table_read(const PowTable& T, uint32_t eoff, uint32_t joff, double& e2, double& tj, double& r)
{
    const double* tr = reinterpret_cast<const double*>(reinterpret_cast<const char*>(&T.TR[0][0]) + joff);
    e2 = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(&T.E[0]) + eoff);
    tj = tr[0]; r = tr[1];
}
#if defined(__HIPCC__)
// ... or one staged in LDS, named by its LDS BYTE ADDRESS.  The kernels' only shared memory is the dynamic block, which
// therefore starts at LDS address 0 (asserted once per launch, k_stream / k_apply); with the address a compile-time
// constant plus the offset the read is `ds_read_b128 v, v_off offset:const` — through a pointer derived from the
// `extern __shared__` symbol the compiler adds the (zero) base to every offset first: 12 extra VALU per pixel.
struct LdsPowTable {
    uint32_t base;
    // The polynomial's first step is fma(t, c4, c3): two constants, and a VOP3 instruction of gfx9 reads only one operand
    // over the constant bus — taken from the instruction stream, c3 costs a v_mov_b64 per evaluation (seven per 128-pixel
    // step).  The kernels therefore keep it in a vector register pair for the whole launch (rpf_kernels.hip LdsTabs).
    double c3;
};
typedef const __attribute__((address_space(3))) double* lds_cdouble_ptr;
typedef double lds_double2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) lds_double2* lds_cdouble2_ptr;
__device__ __forceinline__ void pow_table_read(const LdsPowTable& T, uint32_t eoff, uint32_t joff, double& e2, double& tj,
                                               double& r)
{
#if defined(__HIP_DEVICE_COMPILE__)
    e2 = *(lds_cdouble_ptr)(T.base + (uint32_t)offsetof(PowTable, E) + eoff);
    const lds_double2 tr = *(lds_cdouble2_ptr)(T.base + (uint32_t)offsetof(PowTable, TR) + joff);
    tj = tr.x; r = tr.y;
#else
    (void)T; (void)eoff; (void)joff; e2 = tj = r = 0.0;  // host pass of the HIP compilation: never called
#endif
}
#endif

// where the degree-3 coefficient comes from: the instruction stream, or a register pair the table object holds
template <int KIND>
RPF_HD double pow_c3(const PowTable&) { return PowCoef<KIND>::c3; }
#if defined(__HIPCC__)
template <int KIND>
__device__ __forceinline__ double pow_c3(const LdsPowTable& T) { return T.c3; }
#endif

template <int KIND, class TabT>
RPF_HD double powc_d(float x, const TabT& T)  // the binary64 value before the single rounding
{
    const uint32_t b = f2bits(x);
    const uint32_t eoff = (b >> 20) & 0x7f8u;  // biased exponent * 8: byte offset into E
    const uint32_t joff = (b >> (19 - RPF_POW_BITS)) & (((1u << RPF_POW_BITS) - 1u) << 4);  // top 9 mantissa bits * 16: byte offset into TR
    const double m = (double)bits2f((b & 0x007fffffu) | 0x3f800000u);  // significand in [1, 2), exact
    double e2, tj, r;
    pow_table_read(T, eoff, joff, e2, tj, r);
    const double t = __builtin_fma(m, r, -1.0);
    using K = PowCoef<KIND>;
#if RPF_POW_BITS >= 9
    double p = __builtin_fma(t, K::c4, pow_c3<KIND>(T));  // c5 t^5 < 2^-55: dropped
#else
    double p = __builtin_fma(t, K::c5, K::c4);
    p = __builtin_fma(t, p, K::c3);
#endif
    p = __builtin_fma(t, p, K::c2);
    p = __builtin_fma(t, p, K::c1);
    p = __builtin_fma(t, p, 1.0);
    return (e2 * tj) * p;
}

template <int KIND, class TabT>
RPF_HD float powc(float x, const TabT& T)
{
    return (float)powc_d<KIND>(x, T);
}

// atan(q) = q + q^3 P(q^2), |q| <= tan(pi/8); rel. error 2^-52.7 (tools/gen_math_consts.py)
RPF_HD double atan_small(double q)
{
    const double s = q * q;
    double p = 0x1.74a2d785d5a9fp-6;
    p = __builtin_fma(p, s, -0x1.6f3c3017cb4e7p-5);
    p = __builtin_fma(p, s, 0x1.d5e885b7ee229p-5);
    p = __builtin_fma(p, s, -0x1.105e007ccd67cp-4);
    p = __builtin_fma(p, s, 0x1.3b0688e46275ap-4);
    p = __builtin_fma(p, s, -0x1.745c7f2dc5c2dp-4);
    p = __builtin_fma(p, s, 0x1.c71c6dcf1e479p-4);
    p = __builtin_fma(p, s, -0x1.2492491dcf6e7p-3);
    p = __builtin_fma(p, s, 0x1.9999999990a20p-3);
    p = __builtin_fma(p, s, -0x1.5555555555546p-2);
    return __builtin_fma(q * s, p, q);
}

// atan2(y, x) for finite arguments, rounded to binary32 from a binary64 value with
// rel. error < 2^-49.  Signed zeros as IEEE: atan2(+-0, +0) = +-0, atan2(+-0, -0) = +-pi.
RPF_HD float atan2_cr(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = ax > ay ? ax : ay;
    const float mn = ax > ay ? ay : ax;
    const double dmx = (double)mx, dmn = (double)mn;
    // theta in [0, pi/8]: q = mn/mx;  theta in (pi/8, pi/4]: pi/4 + atan((mn-mx)/(mn+mx))
    const bool upper = mn > 0.41421356f * mx;
    const double num = upper ? dmn - dmx : dmn;
    const double den = upper ? dmn + dmx : dmx;
    // den = 2^E d1, d1 in [1, 2): the binary32 reciprocal below can neither overflow nor lose bits
    const uint64_t ud = d2u(den);
    const uint32_t dhi = (uint32_t)(ud >> 32);
    const int E = (int)((dhi >> 20) & 0x7ffu) - 1023;
    const double d1 = hilo2d((dhi & 0x000fffffu) | 0x3ff00000u, (uint32_t)ud);  // den == 0 -> 1.0, num == 0
    const double n1 = __builtin_ldexp(num, -E);
    // 1/d1 to ~2^-22 by three binary32 Newton steps from the linear seed 24/17 - 8/17 d (error 1/17 ->
    // 3.5e-3 -> 1.2e-5 -> 1.4e-10, then rounding): 7 full-rate fma instead of an IEEE division sequence
    const float df = (float)d1;
    float rc = __builtin_fmaf(df, -0x1.e1e1e2p-2f, 0x1.696969p+0f);
    rc = __builtin_fmaf(rc, __builtin_fmaf(-df, rc, 1.0f), rc);
    rc = __builtin_fmaf(rc, __builtin_fmaf(-df, rc, 1.0f), rc);
    rc = __builtin_fmaf(rc, __builtin_fmaf(-df, rc, 1.0f), rc);
    const double q0 = n1 * (double)rc;
    const double res = __builtin_fma(-q0, d1, n1);
    const double q = __builtin_fma(res, (double)rc, q0);  // rel. error ~2^-46 of a 2^-23 term
    double r = atan_small(q);
    r = upper ? r + 0x1.921fb54442d18p-1 : r;            // + pi/4
    r = ay > ax ? 0x1.921fb54442d18p+0 - r : r;            // pi/2 - r
    r = (f2bits(x) >> 31) ? 0x1.921fb54442d18p+1 - r : r;  // x < 0 or x == -0: pi - r
    const float rf = (float)r;
    return bits2f(f2bits(rf) | (f2bits(y) & 0x80000000u));  // copysign(rf, y)
}

// cos(a), sin(a) for a = f32(f32(k)/65535) * 2 * f32(pi), k in [0, 65535] — the
// only arguments oklch_to_linear_srgb can see after the hue LUT (wgsl:324-328, 78-79).
// Equal to the oracle's cr_cosf/cr_sinf for every k (tests/test_math_exhaustive.py).
template <class TabT>
RPF_HD void sincos_hue(uint32_t k, const TabT& S, float& cs, float& sn)
{
    const float h = div_u16_by(k, 65535.0f, (float)(1.0 / 65535.0));
    const float a = (h * 2.0f) * RPF_K_PI_F;
    const double delta = __builtin_fma(-(double)k, RPF_K_HUE_STEP, (double)a);  // |delta| < 2^-20
    const double c1 = S.hi[k >> 8][0], s1 = S.hi[k >> 8][1];
    const double c0 = S.lo[k & 255u][0], s0 = S.lo[k & 255u][1];
    const double ce = __builtin_fma(c1, c0, -(s1 * s0));
    const double se = __builtin_fma(s1, c0, c1 * s0);
    const double q = __builtin_fma(-0.5 * delta, delta, 1.0);  // cos(delta) to 2^-80
    cs = (float)__builtin_fma(-se, delta, ce * q);
    sn = (float)__builtin_fma(ce, delta, se * q);
}

// ---------------------------------------------------------------------------
// Hue index without an arctangent.
//
// The chain needs atan2 for one thing only: k = u32(h * 65535) with h = atan2(b, a) / (2 pi) wrapped into
// [0, 1) (wgsl:72-73, 320), the index of the per-hue curves; the hue itself is then replaced by curve[k] / 65535.
// k is a circularly monotone step function of the true angle, so it is fixed by 65536 boundary directions:
// HueBound[j] = (cos, sin) of the angle at which k first reaches j, i.e. the midpoint between the two adjacent
// binary32 values t' < t of the correctly rounded atan2 with k(t') < j <= k(t)  (built by build_hue_bounds from
// the same binary32 pipeline: t / f32(2 pi), +1 if negative, * 65535, truncate).  A cheap approximation of the
// angle (error < 0.2 index units) picks the one boundary that can matter, and the sign of the binary64 cross
// product  b cos - a sin  decides on which side of it (a, b) lies — the same decision the correctly rounded
// atan2 followed by the binary32 pipeline makes.  1 MiB table, read from L2.
// One corner is left out: for a true angle in (-3.5 * 2^-149, 0) the reference's t / f32(2 pi) underflows to -0
// and k = 0, here k = 65535.  That needs b < 0 with |b| < 4.9e-45 |a|; in the chain a nonzero |b| is at least
// ~2^-75 (a rounded sum of three products of cube roots >= 2^-50) and |a| < 8, so it cannot occur.
// ---------------------------------------------------------------------------
struct HueBound {
    double c, s;  // direction of the boundary at which the index first reaches j
};
// The table HUE_DIRECT launches read instead (one 32-byte gather; two tables with the same index, or one table for
// everybody, both measured slower): the boundary plus cos, sin of the hue angle of index j (what sincos_hue(j)
// returns) and of index j - 1 — with an identity hue curve the index IS the new hue (wgsl:324-328).
struct HueBoundCs {
    double c, s;
    float cs_k[2];
    float cs_km1[2];
};

// The approximate position of the direction (x, y) on the index scale, in [0, 65535.x]: atan2 / (2 pi), wrapped, times
// 65535, within 0.125 of the true value (the polynomial — tools/gen_math_consts.py — plus the reciprocal and the binary32
// evaluation; measured maximum 0.1227 over 8 M directions, tests/test_math_exhaustive.py).
RPF_HD float hue_candidate(float y, float x, float& mx_out)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    // q = min / max in [0, 1] feeds the approximation only (its error budget is 0.38 index units, a hardware
    // reciprocal's 1 ulp is nothing).  The larger magnitude is clamped into [2^-100, 2^100] first: the floor keeps
    // a = b = 0 at q = 0 instead of 0 * inf, the ceiling keeps the reciprocal a normal number — outside that range
    // (which the chain cannot reach: |a|, |b| < 8, and a nonzero one is > 2^-75) the candidate may be off and the
    // function is not specified.
#if defined(__HIP_DEVICE_COMPILE__)
    // one v_max_f32 / v_min_f32 each with |.| source modifiers (through fmaxf / fminf the compiler first
    // "canonicalises" both magnitudes with a v_max of their own: four half-rate instructions instead of two)
    float mx, mn;
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(mx) : "v"(x), "v"(y));
    asm("v_min_f32 %0, |%1|, |%2|" : "=v"(mn) : "v"(x), "v"(y));
    const float q = mn * __builtin_amdgcn_rcpf(__builtin_amdgcn_fmed3f(mx, 0x1p-100f, 0x1p+100f));
#else
    const float mx = ax > ay ? ax : ay;
    const float mn = ax > ay ? ay : ax;
    const float mxc = mx < 0x1p-100f ? 0x1p-100f : (mx > 0x1p+100f ? 0x1p+100f : mx);
    const float q = mn / mxc;
#endif
    // atan(q) * 65535 / (2 pi) for q in [0, 1], |error| < 0.12 index units (tools/gen_math_consts.py)
    const float z = q * q;
    float p = __builtin_fmaf(z, 217.41815f, -888.19727f);
    p = __builtin_fmaf(z, p, 1879.0994f);
    p = __builtin_fmaf(z, p, -3445.1509f);
    p = __builtin_fmaf(z, p, 10428.825f);
    float v = p * q;
    v = ay > ax ? 16383.75f - v : v;   // pi/2 - a
    v = x < 0.0f ? 32767.5f - v : v;   // pi - a
    v = y < 0.0f ? 65535.0f - v : v;   // 2 pi - a (the wrap of wgsl:73); y = -0 counts as +0, like atan2's -0 result
    mx_out = mx;
    return v;
}

// `bd_out` / `passed_out`: the table entry that was read and the side of its boundary (for hue_index_cs).
// NEG_ZERO_X = false: the caller's promise that x is never -0 while y is +-0 (then atan2 = +-pi, index 32767, is the
// one result no boundary decides); it saves the two compares and the merge that exist for that argument alone.
template <bool NEG_ZERO_X = true, bool SPARSE = false, class Entry>
RPF_HD uint32_t hue_index_core(float y, float x, const Entry* hb, Entry& bd_out, bool& passed_out, uint32_t& zero_negx_out)
{
    float mx;
    const float v = hue_candidate(y, x, mx);
    // nearest integer of v in [0, 65535.x] without a conversion (half-rate on gfx950): the low bits of v + 2^23
    const float vr = v + 8388608.0f;
    const uint32_t j = f2bits(vr) & 0xffffu;
    // SPARSE: v is within 0.125 index units of the true position and boundary j lies within 0.007 of position j (the
    // reference pipeline's own roundings shape it; both measured in tests/test_math_exhaustive.py), so a candidate at
    // least 3/16 away from the integer it rounds to has its side of boundary j settled by v alone.  Only
    // the other lanes (37.5 % for uniformly distributed hues) look their boundary up; the rest read entry 0, one cache
    // line for all of them.  A wavefront's gather touches one 128-byte line per distinct entry, and at two gathers per
    // step that traffic, not its latency, is what the table costs (profiles/r02_experiments.md).
    // Straight-line code on purpose (the ENTRY is selected, not the read): written as a select around the table read, the
    // compiler turns it into a branch per pixel with the gather and its wait inside, which serialises the pair and
    // exposes the whole L2 latency.
    // The 3/16: |v - true position| <= 0.1203 (the polynomial, swept over EVERY binary32 q = min / max in [0, 1] on the
    // device) + 0.001 (v_rcp_f32 and the product) + 0.006 (the three reflections' roundings) and |boundary j - position j|
    // <= 0.007, together 0.134 < 0.1875 (tests/test_gpu_math.py test_hue_candidate_error_over_every_binary32_ratio).
    const float frac = v - (vr - 8388608.0f);  // exact: v - round(v), in [-0.5, 0.5]
    const bool near = !SPARSE || __builtin_fabsf(frac) < 0.1875f;
    const Entry bd = hb[near ? j : 0u];
    const double d = __builtin_fma((double)y, bd.c, -((double)x * bd.s));
    const bool passed = near ? d >= 0.0 : frac >= 0.0f;
    const uint32_t k = (j - (passed ? 0u : 1u)) & 0xffffu;
    bd_out = bd;
    passed_out = passed;
    // a = b = 0 arrives here as j = 0, d = +-0, k = 0, which is atan2(+-0, +0) = +-0 -> 0;
    // atan2(+-0, -0) = +-pi -> 32767 is OR-ed in (k is 0 in that case)
    const uint32_t zero_negx = (NEG_ZERO_X && mx == 0.0f) ? ((f2bits(x) >> 31) * 32767u) : 0u;
    zero_negx_out = zero_negx;
    return k | zero_negx;
}

template <bool NEG_ZERO_X = true>
RPF_HD uint32_t hue_index(float y, float x, const HueBound* hb)
{
    HueBound bd;
    bool passed;
    uint32_t zn;
    return hue_index_core<NEG_ZERO_X, true>(y, x, hb, bd, passed, zn);
}

// The index together with cos / sin of ITS hue angle (= sincos_hue(k)), read from the same table entry: for
// launches whose hue curve is the identity the whole of sincos_hue (four LDS reads, eleven binary64 operations)
// falls away.  The a = b = 0, a = -0 corner (k = 32767, not a neighbour of the entry read) takes the slow path.
template <bool NEG_ZERO_X = true, class TabT>
RPF_HD uint32_t hue_index_cs(float y, float x, const HueBoundCs* hb, const TabT& S, float& cs, float& sn)
{
    HueBoundCs bd;
    bool passed;
    uint32_t zn;
    const uint32_t k = hue_index_core<NEG_ZERO_X>(y, x, hb, bd, passed, zn);
    cs = passed ? bd.cs_k[0] : bd.cs_km1[0];
    sn = passed ? bd.cs_k[1] : bd.cs_km1[1];
    if (NEG_ZERO_X && __builtin_expect(zn != 0u, 0)) sincos_hue(k, S, cs, sn);
    return k;
}

// The table a launch with ONE mask reads when that mask's hue / saturation / lightness curves are not all trivial (built
// per editor from "main"'s curves: rpf_chain.h hue_fused_entry, rpf_kernels.hip k_build_hue_fused): boundary j together
// with everything the chain wants from index j — cos / sin of the hue angle the curve maps j to (what sincos_hue(curve[j])
// returns) and the two gains (wgsl:324-331).  Read the way hue_index reads its boundaries — the boundary only where the
// candidate is near it (hue_index_core's SPARSE rule: 3 lanes of 8, the rest entry 0) — and then the 16 data bytes of the
// index that came out, in place of the curve entry, its two quotients and sincos_hue (four LDS reads, eleven binary64
// operations).  (Read densely — entry j and index j - 1's data with it, one 48-byte gather — the fused develop kernel
// was 2 % slower and the RGBA pass 15 %: profiles/r03_experiments.md section 14.)
struct HueFused {
    double c, s;                         // direction of the boundary at which the index first reaches j
    float cs, sn, sat_gain, light_gain;  // of index j
};

template <bool NEG_ZERO_X = true>
RPF_HD uint32_t hue_index_fused(float y, float x, const HueFused* hf, float& cs, float& sn, float& sat_gain, float& light_gain)
{
    static_assert(!NEG_ZERO_X, "the chain's promise (x is never -0): the k = 32767 corner is not handled here");
    float mx;
    const float v = hue_candidate(y, x, mx);
    const float vr = v + 8388608.0f;
    const uint32_t j = f2bits(vr) & 0xffffu;
    const float frac = v - (vr - 8388608.0f);  // exact: v - round(v); see hue_index_core for the 3/16
    const bool near = __builtin_fabsf(frac) < 0.1875f;
    const HueFused* e = hf + (near ? j : 0u);
    const double bc = e->c, bs = e->s;
    const double d = __builtin_fma((double)y, bc, -((double)x * bs));
    const bool passed = near ? d >= 0.0 : frac >= 0.0f;
    const uint32_t k = (j - (passed ? 0u : 1u)) & 0xffffu;
    const HueFused* ek = hf + k;
    cs = ek->cs; sn = ek->sn; sat_gain = ek->sat_gain; light_gain = ek->light_gain;
    return k;
}

// Correctly rounded binary32 square root (wgsl:70 `sqrt`, wgsl:170 `length`).  The compiler's IEEE expansion of
// sqrtf costs 16 instructions on gfx950 (scaling for subnormal inputs, v_sqrt_f32, two trial residuals, four selects).
// For s in [2^-96, 2^96) none of the scaling is needed and one Markstein step does it: y0 = s g with g = v_rsq_f32(s)
// (1 ulp), the residual r = s - y0^2 is exact in one fma, y = y0 + r (g / 2) rounds once.  Pre-rounding error ~2^-44
// relative; that this always rounds like the IEEE sqrt is checked on EVERY binary32 of the range on the GPU
// (tests/test_gpu_math.py).  Zero, subnormal-range, huge, inf and NaN inputs take the IEEE sequence (a rarely taken,
// wave-uniformly skipped branch).  The host build is sqrtf, which is what the device result equals.
RPF_HD float sqrt_cr(float s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // one unsigned compare on the bit pattern: negative, NaN, inf, zero and everything outside [2^-96, 2^96) fail it
    if (__builtin_expect(f2bits(s) - 0x0f800000u >= 0x60000000u, 0)) return __builtin_sqrtf(s);
    const float g = __builtin_amdgcn_rsqf(s);
    const float y0 = s * g;
    const float r = __builtin_fmaf(-y0, y0, s);
    return __builtin_fmaf(r, 0.5f * g, y0);
#else
    return __builtin_sqrtf(s);
#endif
}

// pow(x, 1.5) correctly rounded for x in [0, 1] (vignette falloff, wgsl:171): x * sqrt(x) in binary64 is within
// 2^-52 of the true value, and exact whenever the result is exactly representable or a binary32 tie (x a perfect
// square).  (The table scheme of powc was tried for it: same kernel time, and it needs a tie fix-up.)
// On the device the binary64 square root (the compiler's expansion: ~17 binary64-rate instructions) is replaced by what
// the binary32 one leaves over: s = sqrt_cr(x) is correctly rounded, so with r = x - s^2 (exact in one fma)
//   x^1.5 = x s + r s / 2 + (3/8) r^2 / s + O(r^3),   x s exact in binary64, 1 / s from v_rsq_f32 (any value within
// 2 ulp of it gives the same bits), the remainder below 2^-70 relative.  Five binary64 operations.  For x in
// {0} u [2^-96, 1] — the vignette's argument is 0 or at least 3.9e-8 — the result has the bits of the formulation above
// for EVERY binary32 of that set: checked on the GPU (tests/test_gpu_math.py).  The host build is the formulation above.
RPF_HD float pow_1p5(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float g = __builtin_amdgcn_rsqf(x);
    const float s0 = x * g;
    const float s = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * g, s0);  // sqrt_cr's Markstein step
    const double sd = (double)s, xd = (double)x;
    const double r = __builtin_fma(-sd, sd, xd);
    const double q = (double)(0.375f * g) * r;
    const double res = __builtin_fma(q, r, __builtin_fma(r, (double)(0.5f * s), xd * sd));
    return x > 0.0f ? (float)res : 0.0f;  // x = 0: g = inf and everything above is NaN
#else
    const double xd = (double)x;
    return (float)(xd * __builtin_sqrt(xd));
#endif
}

// pow(x, 4.0) correctly rounded for finite x >= 0 (Godot-legacy vignette falloff): x^2 is exact in binary64,
// x^4 = hi + lo with hi = RN(x^2 * x^2) and lo the exact fma residual; hi is then rounded to odd with lo as
// the sticky information, after which the single rounding to binary32 is the correct one (53 >= 24 + 2).
RPF_HD float pow_4(float x)
{
    const double s = (double)x * (double)x;
    const double hi = s * s;
    const double lo = __builtin_fma(s, s, -hi);
    uint64_t u = d2u(hi);
    const uint64_t step = lo > 0.0 ? 1ull : ~0ull;  // hi >= 0: towards lo is +1 / -1 on the bit pattern
    u = (lo != 0.0 && (u & 1ull) == 0ull) ? u + step : u;
    return (float)u2d(u);
}

}  // namespace rpf
