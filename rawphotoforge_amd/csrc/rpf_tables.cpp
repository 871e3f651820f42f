// rpf_tables.cpp — host-side construction of the MathTables the kernels stage in LDS.
// Long-double (x87, 64-bit significand) evaluation, rounded once to binary64.
#include "rpf_fast.h"

#include <math.h>

namespace rpf {

static void build_pow(PowTable& P, float cf)
{
    const long double c = (long double)cf;
    for (int b = 1; b < 256; ++b) P.E[b] = (double)powl(2.0L, (long double)(b - 127) * c);
    P.E[0] = 0.0;  // exponent field 0: x = 0 (and subnormals, see powc)
    const int nbins = 1 << RPF_POW_BITS;
    for (int j = 0; j < nbins; ++j) {
        const long double mid = 1.0L + ((long double)j + 0.5L) / (long double)nbins;
        // r_j: reciprocal of the bin centre on a 2^-16 grid, so m * r_j is exact in binary64
        const double r = (double)(roundl((1.0L / mid) * 65536.0L) / 65536.0L);
        P.TR[j][1] = r;
        P.TR[j][0] = (double)powl((long double)r, -c);
    }
}

void build_math_tables(MathTables& T)
{
    build_pow(T.cbrt, (float)(1.0 / 3.0));   // wgsl:67
    build_pow(T.gamma, (float)(1.0 / 2.4));  // wgsl:98
    const long double step = (long double)RPF_K_HUE_STEP;
    for (int k = 0; k < 256; ++k) {
        const long double ah = (long double)(256 * k) * step;
        const long double al = (long double)k * step;
        T.sc.hi[k][0] = (double)cosl(ah);
        T.sc.hi[k][1] = (double)sinl(ah);
        T.sc.lo[k][0] = (double)cosl(al);
        T.sc.lo[k][1] = (double)sinl(al);
    }
}

// wgsl:72-73 and :320 on a correctly rounded atan2 result t
static uint32_t hue_pipeline(float t)
{
    const float hh = t / RPF_K_TWO_PI_F;
    const float h = hh < 0.0f ? hh + 1.0f : hh;
    return (uint32_t)(h * 65535.0f);
}

void build_hue_bounds(HueBound* B, HueBoundCs* Cs, const MathTables& T)
{
    const uint32_t pi_bits = f2bits(RPF_K_PI_F);  // RN(pi): the largest magnitude atan2 returns
    const uint32_t k_pi = hue_pipeline(RPF_K_PI_F);
    B[0].c = 1.0;
    B[0].s = 0.0;
    for (uint32_t j = 0; j < 65536; ++j) {  // what sincos_hue returns for the two indices an entry can decide between
        sincos_hue(j, T.sc, Cs[j].cs_k[0], Cs[j].cs_k[1]);
        sincos_hue((j - 1u) & 0xffffu, T.sc, Cs[j].cs_km1[0], Cs[j].cs_km1[1]);
    }
    for (uint32_t j = 1; j < 65536; ++j) {
        long double theta;
        if (j <= k_pi) {  // first reached on the upper half circle: smallest t in [0, pi] with k(t) >= j
            uint32_t lo = 0, hi = pi_bits;  // k(lo) < j <= k(hi)
            while (hi - lo > 1) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (hue_pipeline(bits2f(mid)) >= j) hi = mid; else lo = mid;
            }
            theta = ((long double)bits2f(lo) + (long double)bits2f(hi)) / 2.0L;
        } else {  // lower half circle, t = -m running from -pi up to -0: largest m with k(-m) >= j
            // k(-m) = 65535 down to m = 4 * 2^-149; below that t / f32(2 pi) underflows to -0 and k = 0 — angles
            // in (-3.5 * 2^-149, 0) are left to boundary 0 = angle 0 (see hue_index: not reachable by the chain)
            uint32_t lo = 4, hi = pi_bits;  // k(-lo) >= j > k(-hi)   (k(-pi) = k_pi)
            while (hi - lo > 1) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (hue_pipeline(-bits2f(mid)) >= j) lo = mid; else hi = mid;
            }
            theta = -((long double)bits2f(lo) + (long double)bits2f(hi)) / 2.0L;
        }
        B[j].c = (double)cosl(theta);
        B[j].s = (double)sinl(theta);
    }
    for (uint32_t j = 0; j < 65536; ++j) {  // the 32-byte entries carry the same boundaries
        Cs[j].c = B[j].c;
        Cs[j].s = B[j].s;
    }
}

// rpf_fast.h: entry (e, j) = { r = 2^-e r0, r0 ~ 1 / bin centre on a 2^-12 grid, T = RN(V), rho = RN((V - T) / T) }, V = 2^(e c) r0^(-c)
template <int KIND>
static void build_fast_pow(FastPowEntry* tab, float cf)
{
    using G = FastPowGeom<KIND>;
    const long double c = (long double)cf;
    for (int row = 0; row < G::ROWS; ++row) {
        const int e = G::E0 + row;
        for (int j = 0; j < FAST_POW_J; ++j) {
            FastPowEntry& o = tab[row * FAST_POW_J + j];
            const long double mid = 1.0L + ((long double)j + 0.5L) / (long double)FAST_POW_J;
            const long double r0 = roundl((1.0L / mid) * 4096.0L) / 4096.0L;
            o.r = (float)ldexpl(r0, -e);  // exact: twelve significant bits
            const long double V = powl(2.0L, (long double)e * c) * powl(r0, -c);
            o.T = (float)V;
            o.rho = (float)((V - (long double)o.T) / (long double)o.T);
            o.pad = 0.0f;
            if (KIND == POW_GAMMA) {  // the whole upper branch of linear_to_srgb (rpf_fast.h srgb_upper_fast): W = 1.055 V - 0.055
                const long double G = (long double)(float)1.055, W = G * V - (long double)(float)0.055;
                o.T = (float)(G * V);
                o.pad = (float)W;
                o.rho = (float)((W - (long double)o.pad) / (long double)o.T);
            }
            if (KIND == POW_CBRT && row == 0) o.r = o.T = o.rho = 0.0f;  // everything below 2^-21 (and 0) -> 0
        }
    }
}

void build_fast_tables(FastTables& T)
{
    build_fast_pow<POW_CBRT>(T.cbrt, (float)(1.0 / 3.0));    // wgsl:67
    build_fast_pow<POW_GAMMA>(T.gamma, (float)(1.0 / 2.4));  // wgsl:98
}

void build_legacy_tables(LegacyTables& T)
{
    build_pow(T.decode, 2.4f);  // rust-godot-legacy/photo-editor/src/wgpu_shader.wgsl:92
}

}  // namespace rpf
