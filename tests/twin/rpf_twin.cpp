// rpf_twin.cpp — HOST build of the device arithmetic (rpf_math.h / rpf_chain.h).
// TEST INFRASTRUCTURE ONLY: lets the CPU test-suite check, without a GPU, that the
// IEEE-only transcendental code the kernels run agrees with the oracle (exhaustively
// on the pow / sincos domains).  Never loaded by rawphotoforge_amd.
#include "../../rawphotoforge_amd/csrc/rpf_chain.h"

#include <stdint.h>
#include <string.h>

#include <vector>

namespace rpf {
void build_math_tables(MathTables& T);
void build_legacy_tables(LegacyTables& T);
void build_hue_bounds(HueBound* B, HueBoundCs* Cs, const MathTables& T);
void build_fast_tables(FastTables& T);
}

// function-local statics: initialised once, thread-safely (several loops below run under OpenMP)
static const rpf::MathTables& tabs()
{
    static const rpf::MathTables* t = [] {
        rpf::MathTables* m = new rpf::MathTables;
        rpf::build_math_tables(*m);
        return m;
    }();
    return *t;
}

static const rpf::LegacyTables& legacy_tabs()
{
    static const rpf::LegacyTables* t = [] {
        rpf::LegacyTables* m = new rpf::LegacyTables;
        rpf::build_legacy_tables(*m);
        return m;
    }();
    return *t;
}

static const rpf::FastTables& fast_tabs()
{
    static const rpf::FastTables* t = [] {
        rpf::FastTables* m = new rpf::FastTables;
        rpf::build_fast_tables(*m);
        return m;
    }();
    return *t;
}

struct HueTabs {
    rpf::HueBound* b;
    rpf::HueBoundCs* c;
};
static const HueTabs& hue_tabs()
{
    static const HueTabs t = [] {
        HueTabs h{new rpf::HueBound[65536], new rpf::HueBoundCs[65536]};
        rpf::build_hue_bounds(h.b, h.c, tabs());
        return h;
    }();
    return t;
}
static const rpf::HueBound* hue_bounds_tab() { return hue_tabs().b; }

struct HostLuts {
    const int32_t* bright; const int32_t* hue; const int32_t* sat; const int32_t* light;
    const rpf::HueFused* fused = nullptr;  // edit_chain's HUE_MODE 2 (twin_apply_fused)
    const rpf::HueFused* hue_fused() const { return fused; }
    static uint32_t fetch(const int32_t* l, uint32_t m, uint32_t i)
    {
        int32_t v = l[(size_t)m * RPF_CURVE_RESOLUTION + i];
        return (uint32_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
    }
    const rpf::HueBound* hue_bounds() const { return hue_bounds_tab(); }
    const rpf::HueBoundCs* hue_bounds_cs() const { return hue_tabs().c; }
    uint32_t brightness(uint32_t m, uint32_t i) const { return fetch(bright, m, i); }
    void hsl(uint32_t m, uint32_t i, const rpf::MaskParams&, uint32_t& nh, float& sg, float& lg) const
    {
        nh = fetch(hue, m, i);
        sg = rpf::div_u16_by(fetch(sat, m, i), 32767.5f, (float)(1.0 / 32767.5));
        lg = rpf::div_u16_by(fetch(light, m, i), 32767.5f, (float)(1.0 / 32767.5));
    }
};

extern "C" {

float twin_pow_cbrt(float x) { return rpf::powc<rpf::POW_CBRT>(x, tabs().cbrt); }
float twin_pow_gamma(float x) { return rpf::powc<rpf::POW_GAMMA>(x, tabs().gamma); }
float twin_atan2(float y, float x) { return rpf::atan2_cr(y, x); }
void twin_sincos_hue(uint32_t k, float* cs, float* sn) { rpf::sincos_hue(k, tabs().sc, *cs, *sn); }
float twin_pow_1p5(float x) { return rpf::pow_1p5(x); }
float twin_pow_4(float x) { return rpf::pow_4(x); }
uint32_t twin_hue_index(float y, float x) { return rpf::hue_index(y, x, hue_bounds_tab()); }
// the path hue_index replaces: correctly rounded atan2 (here the binary64 twin) through the binary32 pipeline
uint32_t twin_hue_index_via_atan2(float y, float x)
{
    const float hh = rpf::div_by_const(rpf::atan2_cr(y, x), RPF_K_TWO_PI_F, (float)(1.0 / (double)RPF_K_TWO_PI_F));
    const float h = hh < 0.0f ? hh + 1.0f : hh;
    return (uint32_t)(h * 65535.0f);
}
void twin_hue_index_many(const float* y, const float* x, size_t n, uint32_t* out)
{
    const rpf::HueBound* hb = hue_bounds_tab();
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) out[i] = rpf::hue_index(y[i], x[i], hb);
}
void twin_hue_candidate_many(const float* y, const float* x, size_t n, float* out)
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        float mx;
        out[i] = rpf::hue_candidate(y[i], x[i], mx);
    }
}
void twin_hue_bound(uint32_t j, double* c, double* s) { *c = hue_bounds_tab()[j].c; *s = hue_bounds_tab()[j].s; }
float twin_srgb_to_linear(float c) { return rpf::srgb_to_linear1(c, legacy_tabs().decode); }
float twin_vignette_gain(int legacy, float strength, int x, int y, float w, float h)
{
    return legacy ? rpf::vignette_gain_legacy(strength, x, y, w, h) : rpf::vignette_gain(strength, x, y, w, h);
}
float twin_div_u16(uint32_t k, float d) { return rpf::div_u16_by(k, d, 1.0f / d); }
float twin_div_two_pi(float x) { return rpf::div_by_const(x, RPF_K_TWO_PI_F, (float)(1.0 / (double)RPF_K_TWO_PI_F)); }
void twin_div_const_range(float d, uint32_t bits_lo, uint32_t n, float* out)
{
    const float rd = (float)(1.0 / (double)d);
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i)
        out[i] = rpf::div_by_const(rpf::bits2f(bits_lo + (uint32_t)i), d, rd);
}
void twin_div_two_pi_range(uint32_t bits_lo, uint32_t n, float* out)
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i)
        out[i] = twin_div_two_pi(rpf::bits2f(bits_lo + (uint32_t)i));
}
float twin_hue_angle(uint32_t k)
{
    const float h = rpf::div_u16_by(k, 65535.0f, (float)(1.0 / 65535.0));
    return (h * 2.0f) * RPF_K_PI_F;
}

// range sweeps: count mismatches of twin f(x) against ref[] over consecutive bit patterns
void twin_pow_range(int which, uint32_t bits_lo, uint32_t n, float* out)
{
    const rpf::MathTables& T = tabs();
    const rpf::LegacyTables& L = legacy_tabs();
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        float x = rpf::bits2f(bits_lo + (uint32_t)i);
        out[i] = which == 0   ? rpf::powc<rpf::POW_CBRT>(x, T.cbrt)
                 : which == 1 ? rpf::powc<rpf::POW_GAMMA>(x, T.gamma)
                 : which == 2 ? rpf::powc<rpf::POW_DECODE>(x, L.decode)
                 : which == 4 ? rpf::pow_1p5(x)
                              : rpf::pow_4(x);
    }
}

// ---- RPF_PRECISION_FAST builtins (rpf_fast.h) --------------------------------------------------
float twin_fast_pow_cbrt(float x) { return rpf::powc_fast<rpf::POW_CBRT>(x, fast_tabs().cbrt); }
float twin_fast_srgb_upper(float x) { return rpf::srgb_upper_fast(x, fast_tabs().gamma); }
void twin_fast_pow_range(int which, uint32_t bits_lo, uint32_t n, float* out)
{
    const rpf::FastTables& T = fast_tabs();
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        const float x = rpf::bits2f(bits_lo + (uint32_t)i);
        out[i] = which == 0 ? rpf::powc_fast<rpf::POW_CBRT>(x, T.cbrt) : rpf::srgb_upper_fast(x, T.gamma);
    }
}
size_t twin_fast_tables(void* out, size_t cap)  // the FastTables as the product builds them (for the on-device checks)
{
    if (cap >= sizeof(rpf::FastTables)) memcpy(out, &fast_tabs(), sizeof(rpf::FastTables));
    return sizeof(rpf::FastTables);
}
void twin_fast_sincos_all(float* cs, float* sn)  // 65536 each
{
    for (uint32_t k = 0; k < 65536; ++k) rpf::sincos_hue_fast((float)k, cs[k], sn[k]);
}
void twin_fast_hue_fraction_many(const float* y, const float* x, size_t n, float* out)
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) out[i] = rpf::hue_fraction_fast(y[i], x[i]);
}
void twin_fast_hue_position_many(const float* y, const float* x, size_t n, float* out)
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) out[i] = rpf::hue_position_fast(y[i], x[i]);
}

// whole edit chain on npix pixels (rgb in, rgba out), masks all-ones
void twin_edit_chain(const float* rgb, size_t npix, const rpf::MaskParams* mp, uint32_t num_masks,
                     const int32_t* bright, const int32_t* hue, const int32_t* sat, const int32_t* light,
                     float* rgba_out)
{
    HostLuts L{bright, hue, sat, light};
    const rpf::MathTables& T = tabs();
    const uint32_t active = num_masks >= 32 ? 0xffffffffu : ((1u << num_masks) - 1u);
#pragma omp parallel for
    for (long long i = 0; i < (long long)npix; ++i) {
        float r[1] = {rgb[i * 3]}, g[1] = {rgb[i * 3 + 1]}, b[1] = {rgb[i * 3 + 2]};
        const uint32_t act[1] = {active};
        rpf::edit_chain<1>(r, g, b, act, mp, num_masks, L, T);
        rgba_out[i * 4] = r[0]; rgba_out[i * 4 + 1] = g[0]; rgba_out[i * 4 + 2] = b[0]; rgba_out[i * 4 + 3] = 1.0f;
    }
}

// The same chain for ONE mask with its hue / saturation / lightness curves folded into the fused hue table (edit_chain's
// HUE_MODE 2, what the fused develop kernels run for such an editor): the table is built here as k_build_hue_fused builds it.
void twin_edit_chain_fused(const float* rgb, size_t npix, const rpf::MaskParams* mp, const int32_t* bright, const int32_t* hue,
                           const int32_t* sat, const int32_t* light, float* rgba_out)
{
    HostLuts L{bright, hue, sat, light};
    const rpf::MathTables& T = tabs();
    std::vector<rpf::HueFused> store(RPF_CURVE_RESOLUTION);
    rpf::HueFused* table = store.data();
    for (uint32_t j = 0; j < RPF_CURVE_RESOLUTION; ++j)
        table[j] = rpf::hue_fused_entry(hue_bounds_tab()[j], HostLuts::fetch(hue, 0, j), HostLuts::fetch(sat, 0, j),
                                        HostLuts::fetch(light, 0, j), mp[0], T.sc);
    L.fused = table;
#pragma omp parallel for
    for (long long i = 0; i < (long long)npix; ++i) {
        float r[1] = {rgb[i * 3]}, g[1] = {rgb[i * 3 + 1]}, b[1] = {rgb[i * 3 + 2]};
        const uint32_t act[1] = {1u};
        rpf::edit_chain<1, 2>(r, g, b, act, mp, 1u, L, T);
        rgba_out[i * 4] = r[0]; rgba_out[i * 4 + 1] = g[0]; rgba_out[i * 4 + 2] = b[0]; rgba_out[i * 4 + 3] = 1.0f;
    }
}

// RPF_PRECISION_FAST: edit_chain_fast with the host forms of its builtins (the device's pows, sine and cosine bit for bit;
// its square root and the reciprocal inside the hue position are hardware approximations there, IEEE operations here)
struct HostFastTabs {
    const rpf::FastPowEntry* cbrt;
    const rpf::FastPowEntry* gamma;
};
void twin_edit_chain_fast(const float* rgb, size_t npix, const rpf::MaskParams* mp, uint32_t num_masks,
                          const int32_t* bright, const int32_t* hue, const int32_t* sat, const int32_t* light,
                          float* rgba_out, bool allow_rot = true)
{
    HostLuts L{bright, hue, sat, light};
    const HostFastTabs T{fast_tabs().cbrt, fast_tabs().gamma};
    const uint32_t active = num_masks >= 32 ? 0xffffffffu : ((1u << num_masks) - 1u);
    bool all_hue_identity = true;
    for (uint32_t m = 0; m < num_masks; ++m) all_hue_identity = all_hue_identity && (mp[m].flags & rpf::MF_HUE_IDENTITY);
    all_hue_identity = all_hue_identity && allow_rot;
    std::vector<rpf::HueFused> store;
    if (num_masks == 1 && !(mp[0].flags & rpf::MF_HUE_IDENTITY) && allow_rot) {  // as k_build_hue_fused builds it
        store.resize(RPF_CURVE_RESOLUTION);
        for (uint32_t j = 0; j < RPF_CURVE_RESOLUTION; ++j)
            store[j] = rpf::hue_fused_entry(hue_bounds_tab()[j], HostLuts::fetch(hue, 0, j), HostLuts::fetch(sat, 0, j),
                                            HostLuts::fetch(light, 0, j), mp[0], tabs().sc);
        L.fused = store.data();
    }
#pragma omp parallel for
    for (long long i = 0; i < (long long)npix; ++i) {
        float r[1] = {rgb[i * 3]}, g[1] = {rgb[i * 3 + 1]}, b[1] = {rgb[i * 3 + 2]};
        const uint32_t act[1] = {active};
        // as the kernels choose: without a hue curve on any mask, the rotation form (edit_chain_fast's HUE_ROT)
        if (all_hue_identity) rpf::edit_chain_fast<1, -1, true>(r, g, b, act, mp, num_masks, L, T);
        else if (L.fused) rpf::edit_chain_fast<1, -1, false, true>(r, g, b, act, mp, num_masks, L, T);   // one mask: its fused hue table
        else rpf::edit_chain_fast<1>(r, g, b, act, mp, num_masks, L, T);
        rgba_out[i * 4] = r[0]; rgba_out[i * 4 + 1] = g[0]; rgba_out[i * 4 + 2] = b[0]; rgba_out[i * 4 + 3] = 1.0f;
    }
}

size_t twin_sizeof_mask_params(void) { return sizeof(rpf::MaskParams); }
}

// ---- whole-chain twin driven from reference-style EditParameters ------------------------------
#include "../../rawphotoforge_amd/csrc/rpf_host.h"

extern "C" {

struct twin_edit_params {  // same layout as oracle's orc_edit_params
    float exposure;
    int32_t contrast, shadow, highlight, black, white, wb_temperature, wb_tint, vignette, lens_distortion;
    float mask_range;
};

// rgb: npix x 3 linear input; luts: num_masks x 65536 each.  Runs make_mask_params (the product's
// host code) + edit_chain (the kernels' arithmetic) on the CPU.
void twin_apply_mode(const float* rgb, size_t npix, const twin_edit_params* ep, uint32_t num_masks,
                     const int32_t* bright, const int32_t* hue, const int32_t* sat, const int32_t* light,
                     float* rgba_out, int fused)   // 1 (num_masks == 1): through the fused hue table; 2: RPF_PRECISION_FAST
{
    std::vector<rpf::MaskParams> mp(num_masks);
    for (uint32_t m = 0; m < num_masks; ++m) {
        rpf::EditParameters p;
        p.exposure = ep[m].exposure; p.contrast = ep[m].contrast; p.shadow = ep[m].shadow;
        p.highlight = ep[m].highlight; p.black = ep[m].black; p.white = ep[m].white;
        p.wb_temperature = ep[m].wb_temperature; p.wb_tint = ep[m].wb_tint;
        const int32_t* src[4] = {bright, hue, sat, light};
        for (int c = 0; c < 4; ++c)
            p.curve[c].assign(src[c] + (size_t)m * RPF_CURVE_RESOLUTION, src[c] + (size_t)(m + 1) * RPF_CURVE_RESOLUTION);
        mp[m] = rpf::make_mask_params(p);
    }
    if (fused == 2 || fused == 3)  // 3: the builtin route also where no mask has a hue curve (A/B of edit_chain_fast's HUE_ROT)
        twin_edit_chain_fast(rgb, npix, mp.data(), num_masks, bright, hue, sat, light, rgba_out, fused == 2);
    else if (fused) twin_edit_chain_fused(rgb, npix, mp.data(), bright, hue, sat, light, rgba_out);
    else twin_edit_chain(rgb, npix, mp.data(), num_masks, bright, hue, sat, light, rgba_out);
}

void twin_apply(const float* rgb, size_t npix, const twin_edit_params* ep, uint32_t num_masks,
                const int32_t* bright, const int32_t* hue, const int32_t* sat, const int32_t* light,
                float* rgba_out)
{
    twin_apply_mode(rgb, npix, ep, num_masks, bright, hue, sat, light, rgba_out, 0);
}

// The exact chain under real coverage: `planes` = ceil(num_masks / 16) planes of npix u16 as the product lays them out
// (rpf_launch.h DeviceEdit::mask_bits: plane 0 bit j = user mask j + 1, plane g >= 1 bit j = mask 16 g + j); more than 16
// masks go through edit_chain's WideMasks as in k_apply<SINGLE = 7>.
void twin_apply_masked(const float* rgb, size_t npix, const twin_edit_params* ep, uint32_t num_masks, const int32_t* bright,
                       const int32_t* hue, const int32_t* sat, const int32_t* light, const uint16_t* planes, float* rgba_out)
{
    std::vector<rpf::MaskParams> mp(num_masks);
    for (uint32_t m = 0; m < num_masks; ++m) {
        rpf::EditParameters p;
        p.exposure = ep[m].exposure; p.contrast = ep[m].contrast; p.shadow = ep[m].shadow;
        p.highlight = ep[m].highlight; p.black = ep[m].black; p.white = ep[m].white;
        p.wb_temperature = ep[m].wb_temperature; p.wb_tint = ep[m].wb_tint;
        const int32_t* src[4] = {bright, hue, sat, light};
        for (int c = 0; c < 4; ++c)
            p.curve[c].assign(src[c] + (size_t)m * RPF_CURVE_RESOLUTION, src[c] + (size_t)(m + 1) * RPF_CURVE_RESOLUTION);
        mp[m] = rpf::make_mask_params(p);
    }
    HostLuts L{bright, hue, sat, light};
    const rpf::MathTables& T = tabs();
#pragma omp parallel for
    for (long long i = 0; i < (long long)npix; ++i) {
        float r[1] = {rgb[i * 3]}, g[1] = {rgb[i * 3 + 1]}, b[1] = {rgb[i * 3 + 2]};
        const uint32_t act[1] = {1u | ((uint32_t)planes[i] << 1)};
        if (num_masks > RPF_MASKS_PER_WORD)
            rpf::edit_chain<1>(r, g, b, act, mp.data(), (uint32_t)RPF_MASKS_PER_WORD, L, T,
                               rpf::WideMasks<1>{num_masks, planes, npix, {(size_t)i}});
        else
            rpf::edit_chain<1>(r, g, b, act, mp.data(), num_masks, L, T);
        rgba_out[i * 4] = r[0]; rgba_out[i * 4 + 1] = g[0]; rgba_out[i * 4 + 2] = b[0]; rgba_out[i * 4 + 3] = 1.0f;
    }
}
}

// ---- the demosaic formulas as the kernels evaluate them (rpf_chain.h mhc_*: powers of two carried into fma) ---------
extern "C" void twin_mhc(size_t n, const float* c, const float* s1, const float* e1, const float* d, const float* v2,
                         const float* h2, float* cross, float* diag, float* horiz, float* vert)
{
    for (size_t i = 0; i < n; ++i) {
        cross[i] = rpf::mhc_cross(c[i], s1[i], e1[i], v2[i], h2[i]);
        diag[i] = rpf::mhc_diag(c[i], d[i], v2[i], h2[i]);
        horiz[i] = rpf::mhc_axis(c[i], e1[i], d[i], h2[i], v2[i]);
        vert[i] = rpf::mhc_axis(c[i], s1[i], d[i], v2[i], h2[i]);
    }
}

// ---- host state machine under sanitizers (tools/sanitize_cpu.sh) ---------------------------------
extern "C" {
// the PPM header grammar of the C ABI (rpf_host.h), for the sanitizer run and the no-GPU tests
int twin_ppm_header(const uint8_t* file, size_t len, uint32_t* w, uint32_t* h, size_t* off)
{
    return rpf::ppm16_parse_header(file, len, w, h, off).code;
}
}

extern "C" int twin_state_exercise(void)
{
    rpf::EditState st(7, 5);
    int fails = 0;
    fails += !st.set_whitebalance(500, -500, nullptr) || st.masks[0].params.wb_temperature != 100;
    fails += !!st.set_whitebalance(1, 1, "nope");
    fails += !st.set_tone(99.0f, 500, -500, 3, 4, 5, nullptr) || st.masks[0].params.exposure != 10.0f;
    const int32_t xs[4] = {0, 100, 50000, 65535}, ys[4] = {0, 70000, -5, 65535}, bad[3] = {0, 5, 5};
    fails += !st.set_curve(RPF_CURVE_HUE, nullptr, 0, xs, 4, ys, 4, nullptr);
    fails += st.set_curve(RPF_CURVE_HUE, nullptr, 0, bad, 3, ys, 3, nullptr).code != RPF_ERR_INTERP_NOT_STRICTLY_INCREASING;
    fails += st.set_curve(RPF_CURVE_HUE, nullptr, 0, xs, 4, ys, 3, nullptr).code != RPF_ERR_INTERP_MISMATCHED_LENGTHS;
    fails += st.set_curve(RPF_CURVE_HUE, xs, 4, nullptr, 0, nullptr, 0, nullptr).code != RPF_ERR_INTERP_INVALID_CURVE_LENGTH;
    fails += st.set_curve(RPF_CURVE_HUE, nullptr, 0, nullptr, 0, nullptr, 0, nullptr).code != RPF_ERR_INTERP_MISSING_CURVE_OR_CONTROL_POINTS;
    std::vector<float> m(35, 0.5f);
    for (int i = 0; i < 40; ++i) fails += !st.add_mask(("m" + std::to_string(i)).c_str(), m.data());  // lib.rs:481-499: no limit
    fails += !st.remove_mask("m3") || !st.remove_mask("main") || st.masks.size() != (size_t)40;
    for (auto& mk : st.masks) (void)rpf::make_mask_params(mk.params);
    st.reset();
    fails += st.masks.size() != 1 || st.masks[0].params.wb_temperature != 0;
    return fails;
}
