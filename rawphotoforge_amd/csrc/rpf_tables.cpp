// rpf_tables.cpp — host-side construction of the MathTables the kernels stage in LDS.
// Long-double (x87, 64-bit significand) evaluation, rounded once to binary64.
#include "rpf_math.h"

#include <math.h>

namespace rpf {

static void build_pow(PowTable& P, float cf)
{
    const long double c = (long double)cf;
    for (int e = RPF_POW_EMIN; e < RPF_POW_EMIN + RPF_POW_ECNT; ++e)
        P.E[e - RPF_POW_EMIN] = (double)powl(2.0L, (long double)e * c);
    P.E[0] = 0.0;  // 2^-160 is below every binary32: entry 0 is only ever reached by x = 0 (see powc)
    for (int j = 0; j < 128; ++j) {
        const long double mid = 1.0L + ((long double)j + 0.5L) / 128.0L;
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

void build_legacy_tables(LegacyTables& T)
{
    build_pow(T.decode, 2.4f);  // rust-godot-legacy/photo-editor/src/wgpu_shader.wgsl:92
}

}  // namespace rpf
