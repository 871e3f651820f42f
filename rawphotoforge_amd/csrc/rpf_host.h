// rpf_host.h — host-side operator state of the hot path (no HIP in here):
// EditParameters / Mask bookkeeping, the setters' clamps and validation, PCHIP,
// the EditParameters -> GpuEditParameters conversion and the packing of uniform
// parameters and curves for the kernels.  Mirrors rust/photo-editor/src/lib.rs,
// interpolation.rs and gpu_image_processing.rs:232-254.  Compile with
// -ffp-contract=off: the f32 arithmetic here is part of the parity contract.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rpf.h"
#include "rpf_chain.h"

namespace rpf {

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
struct Status {
    int code = RPF_OK;
    std::string msg;
    static Status ok() { return {}; }
    static Status err(int c, std::string m) { Status s; s.code = c; s.msg = std::move(m); return s; }
    explicit operator bool() const { return code == RPF_OK; }
};

// ---------------------------------------------------------------------------
// pchip_interpolate — interpolation.rs:11-123 (f32 arithmetic, op for op)
// ---------------------------------------------------------------------------
inline int32_t f32_as_i32(float v)  // Rust `as`: truncate, saturate, NaN -> 0
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

inline Status pchip_interpolate_i32(const int32_t* x_pts, size_t nx, const int32_t* y_pts, size_t ny,
                                    const int32_t* x_eval, size_t n_eval, int32_t* y_eval, int32_t* bad_index)
{
    if (nx != ny)
        return Status::err(RPF_ERR_INTERP_MISMATCHED_LENGTHS, "Input arrays must have the same length: x has " +
                                                                  std::to_string(nx) + ", y has " + std::to_string(ny));
    if (nx < 2)
        return Status::err(RPF_ERR_INTERP_NOT_ENOUGH_POINTS,
                           "At least two points are required for interpolation, but got " + std::to_string(nx));
    const size_t n = nx;
    std::vector<float> xp(n), yp(n), del(n - 1), h(n - 1), slopes(n, 0.0f);
    for (size_t i = 0; i < n; ++i) { xp[i] = (float)x_pts[i]; yp[i] = (float)y_pts[i]; }
    for (size_t i = 0; i + 1 < n; ++i) {
        const float h_i = xp[i + 1] - xp[i];
        if (h_i <= 0.0f) {
            if (bad_index) *bad_index = (int32_t)i;
            return Status::err(RPF_ERR_INTERP_NOT_STRICTLY_INCREASING,
                               "x_pts must be strictly increasing, but found a non-increasing value at index " +
                                   std::to_string(i));
        }
        h[i] = h_i;
        del[i] = (yp[i + 1] - yp[i]) / h_i;
    }
    slopes[0] = del[0];
    slopes[n - 1] = del[n - 2];
    for (size_t i = 1; i + 1 < n; ++i) {
        if (del[i - 1] * del[i] <= 0.0f) {
            slopes[i] = 0.0f;
        } else {
            const float w1 = 2.0f * h[i] + h[i - 1];
            const float w2 = h[i] + 2.0f * h[i - 1];
            slopes[i] = (w1 + w2) / (w1 / del[i - 1] + w2 / del[i]);
        }
    }
    for (size_t k = 0; k < n_eval; ++k) {
        const float x = (float)x_eval[k];
        float y;
        if (x <= xp[0]) {
            y = yp[0];
        } else if (x >= xp[n - 1]) {
            y = yp[n - 1];
        } else {
            size_t lo = 0, hi = n;  // last i with xp[i] <= x
            while (lo < hi) {
                const size_t mid = lo + (hi - lo) / 2;
                if (xp[mid] <= x) lo = mid + 1; else hi = mid;
            }
            size_t i = lo - 1;
            if (i > n - 2) i = n - 2;
            const float h_val = h[i];
            const float t = (x - xp[i]) / h_val;
            const float t2 = t * t;
            const float t3 = t2 * t;
            const float h00 = 2.0f * t3 - 3.0f * t2 + 1.0f;
            const float h10 = t3 - 2.0f * t2 + t;
            const float h01 = -2.0f * t3 + 3.0f * t2;
            const float h11 = t3 - t2;
            y = h00 * yp[i] + h10 * h_val * slopes[i] + h01 * yp[i + 1] + h11 * h_val * slopes[i + 1];
        }
        y_eval[k] = f32_as_i32(y);
    }
    return Status::ok();
}

// ---------------------------------------------------------------------------
// EditParameters / Mask — lib.rs:19-75
// ---------------------------------------------------------------------------
struct EditParameters {
    float exposure = 0.0f;
    int32_t contrast = 0, shadow = 0, highlight = 0, black = 0, white = 0;
    int32_t wb_temperature = 0, wb_tint = 0;
    int32_t vignette = 0;
    int32_t lens_distortion = 0;
    float mask_range = 0.0f;
    std::vector<int32_t> curve[4];  // brightness, hue, saturation, lightness
    // Bookkeeping for callers that re-send every curve on every frame (web/main.ts:809-826): a stamp per curve
    // (0 = the default; set_curve draws a new one from EditState::stamp only when the curve really changes), the
    // control points that produced it, and what make_mask_params derives from it (identity / constant).
    uint64_t curve_ver[4] = {0, 0, 0, 0};
    std::vector<int32_t> cp_x[4], cp_y[4];
    mutable uint64_t trait_ver[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    mutable bool trait_trivial[4] = {false, false, false, false};
    mutable uint32_t trait_first[4] = {0, 0, 0, 0};

    // identity (brightness, hue) / constant (saturation, lightness) after the fetch-time clamp, cached per stamp
    bool curve_trivial(int which, uint32_t* first = nullptr) const
    {
        if (trait_ver[which] != curve_ver[which]) {
            const std::vector<int32_t>& c = curve[which];
            auto cl = [](int32_t v) { return (uint32_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v)); };
            const uint32_t v0 = cl(c[0]);
            bool t = true;
            if (which == RPF_CURVE_BRIGHTNESS || which == RPF_CURVE_HUE) {
                for (int i = 0; i < RPF_CURVE_RESOLUTION && t; ++i) t = cl(c[i]) == (uint32_t)i;
            } else {
                for (int i = 0; i < RPF_CURVE_RESOLUTION && t; ++i) t = cl(c[i]) == v0;
            }
            trait_trivial[which] = t;
            trait_first[which] = v0;
            trait_ver[which] = curve_ver[which];
        }
        if (first) *first = trait_first[which];
        return trait_trivial[which];
    }

    EditParameters()
    {
        for (auto& c : curve) c.resize(RPF_CURVE_RESOLUTION);
        for (int i = 0; i < RPF_CURVE_RESOLUTION; ++i) {  // lib.rs:58-61
            curve[RPF_CURVE_BRIGHTNESS][i] = i;
            curve[RPF_CURVE_HUE][i] = i;
            curve[RPF_CURVE_SATURATION][i] = 32767;
            curve[RPF_CURVE_LIGHTNESS][i] = 32767;
        }
    }
};

inline int32_t clamp_i32(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// gpu_image_processing.rs:235-247
inline rpf_gpu_edit_parameters convert_params(const EditParameters& p)
{
    rpf_gpu_edit_parameters g;
    g.r_gain = 1.0f + 0.5f * ((float)p.wb_temperature / 100.0f);
    g.g_gain = 1.0f - 0.25f * ((float)p.wb_tint / 100.0f);
    g.b_gain = 1.0f - 0.5f * ((float)p.wb_temperature / 100.0f);
    g.vignette = (float)p.vignette;
    g.lens_distortion = (float)p.lens_distortion;
    g.exposure = p.exposure;
    g.contrast = (float)p.contrast / 100.0f;
    g.shadow = (float)p.shadow / 100.0f;
    g.highlight = (float)p.highlight / 100.0f;
    g.black = (float)p.black / 100.0f;
    g.white = (float)p.white / 100.0f;
    return g;
}

// pow(2.0, exposure) correctly rounded to binary32 (wgsl:214): long-double exp2 has 11
// guard bits beyond binary64, far inside the margin needed for |exposure| <= 10.
inline float exp2_cr(float e) { return (float)exp2l((long double)e); }

inline uint32_t lut_clamp(int32_t v) { return (uint32_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v)); }  // wgsl:188

// Uniform block for one mask + what the kernels may skip.
inline MaskParams make_mask_params(const EditParameters& p)
{
    const rpf_gpu_edit_parameters g = convert_params(p);
    MaskParams m;
    memset(&m, 0, sizeof(m));
    m.r_gain = g.r_gain; m.g_gain = g.g_gain; m.b_gain = g.b_gain;
    m.exposure_mul = exp2_cr(g.exposure);
    m.shadow = g.shadow; m.highlight = g.highlight;
    m.black = g.black; m.white = g.white;
    m.contrast_c = 1.0f + g.contrast;
    uint32_t f = 0;
    if (g.black != 0.0f) f |= MF_BLACK;
    if (g.white != 0.0f) f |= MF_WHITE;
    if (g.contrast != 0.0f) f |= MF_CONTRAST;
    if (m.r_gain == 1.0f && m.g_gain == 1.0f && m.b_gain == 1.0f) f |= MF_UNITY_WB;
    if (m.exposure_mul == 1.0f) f |= MF_UNITY_EXPOSURE;
    if (g.shadow == 0.0f && g.highlight == 0.0f) f |= MF_NO_SHADOW_HIGHLIGHT;
    uint32_t s0 = 0, l0 = 0;
    if (p.curve_trivial(RPF_CURVE_BRIGHTNESS)) f |= MF_BRIGHT_IDENTITY;
    if (p.curve_trivial(RPF_CURVE_HUE)) f |= MF_HUE_IDENTITY;
    if (p.curve_trivial(RPF_CURVE_SATURATION, &s0)) { f |= MF_SAT_CONST; m.sat_gain = (float)s0 / 32767.5f; }    // wgsl:329
    if (p.curve_trivial(RPF_CURVE_LIGHTNESS, &l0)) { f |= MF_LIGHT_CONST; m.light_gain = (float)l0 / 32767.5f; }  // wgsl:330
    m.flags = f;
    return m;
}

inline void cfa_phase(uint32_t pattern, int& xo, int& yo)
{
    switch (pattern & 3u) {
    default:
    case RPF_CFA_RGGB: xo = 0; yo = 0; break;
    case RPF_CFA_BGGR: xo = 1; yo = 1; break;
    case RPF_CFA_GRBG: xo = 1; yo = 0; break;
    case RPF_CFA_GBRG: xo = 0; yo = 1; break;
    }
}

inline RawParamsDev make_raw_params(const rpf_raw_params& r)
{
    RawParamsDev d;
    for (int c = 0; c < 4; ++c) {
        d.black[c] = r.black[c];
        d.range[c] = r.white - r.black[c];
        d.scale[c] = r.wb[c] / d.range[c];
    }
    for (int i = 0; i < 9; ++i) d.cam[i] = r.cam2rgb[i];
    cfa_phase(r.cfa_pattern, d.xo, d.yo);
    return d;
}

// ---------------------------------------------------------------------------
// EditState — the device-free half of PhotoEditor (lib.rs:77-84, 227-505)
// ---------------------------------------------------------------------------
struct MaskState {
    std::string name;
    EditParameters params;
    std::vector<uint8_t> binary;  // empty for "main" (all ones, lib.rs:193-196)
};

struct EditState {
    uint32_t width = 0, height = 0;
    std::vector<MaskState> masks;
    uint64_t stamp = 0;           // source of EditParameters::curve_ver
    uint64_t params_version = 1;  // bumped by every setter; device copies are refreshed lazily
    uint64_t masks_version = 1;

    EditState(uint32_t w, uint32_t h) : width(w), height(h)
    {
        masks.emplace_back();
        masks[0].name = "main";
    }

    EditParameters* find(const char* mask_name, Status& st)
    {
        const std::string name = mask_name ? mask_name : "main";
        for (auto& m : masks)
            if (m.name == name) return &m.params;
        st = Status::err(RPF_ERR_MASK_NOT_FOUND, "Mask not found: The specified mask '" + name + "' does not exist.");
        return nullptr;
    }
    const EditParameters* find(const char* mask_name, Status& st) const
    {
        return const_cast<EditState*>(this)->find(mask_name, st);
    }

    Status set_whitebalance(int32_t temperature, int32_t tint, const char* mask_name)  // lib.rs:255-265
    {
        Status st;
        EditParameters* a = find(mask_name, st);
        if (!a) return st;
        a->wb_temperature = clamp_i32(temperature, -100, 100);
        a->wb_tint = clamp_i32(tint, -100, 100);
        ++params_version;
        return st;
    }
    Status set_vignette(int32_t v)  // lib.rs:267-270
    {
        masks[0].params.vignette = clamp_i32(v, -100, 100);
        ++params_version;
        return Status::ok();
    }
    Status set_lens_distortion(int32_t v)  // lib.rs:272-275
    {
        masks[0].params.lens_distortion = clamp_i32(v, -100, 100);
        ++params_version;
        return Status::ok();
    }
    Status set_tone(float exposure, int32_t contrast, int32_t shadow, int32_t highlight, int32_t black,
                    int32_t white, const char* mask_name)  // lib.rs:278-298
    {
        Status st;
        EditParameters* a = find(mask_name, st);
        if (!a) return st;
        // f32::clamp(-10, 10): NaN stays NaN in Rust; keep that
        a->exposure = exposure < -10.0f ? -10.0f : (exposure > 10.0f ? 10.0f : exposure);
        a->contrast = clamp_i32(contrast, -100, 100);
        a->shadow = clamp_i32(shadow, -100, 100);
        a->highlight = clamp_i32(highlight, -100, 100);
        a->black = clamp_i32(black, -100, 100);
        a->white = clamp_i32(white, -100, 100);
        ++params_version;
        return st;
    }

    // lib.rs:300-479 (the four setters are copies of one another)
    Status set_curve(int which, const int32_t* curve, size_t curve_len, const int32_t* xs, size_t n_x,
                     const int32_t* ys, size_t n_y, const char* mask_name)
    {
        if (!curve && !xs)
            return Status::err(RPF_ERR_INTERP_MISSING_CURVE_OR_CONTROL_POINTS,
                               "Interpolation failed: You must provide either a full curve or control points.");
        // Same control points as the ones this curve was built from: nothing to do (they passed every check then).
        {
            Status ignore;
            const EditParameters* cur = find(mask_name, ignore);
            if (cur && !curve && ys && n_x == n_y && n_x > 0 && cur->cp_x[which].size() == n_x &&
                memcmp(cur->cp_x[which].data(), xs, n_x * sizeof(int32_t)) == 0 &&
                memcmp(cur->cp_y[which].data(), ys, n_y * sizeof(int32_t)) == 0)
                return Status::ok();
        }
        std::vector<int32_t> final_curve;
        if (curve) {
            if (curve_len != RPF_CURVE_RESOLUTION)
                return Status::err(RPF_ERR_INTERP_INVALID_CURVE_LENGTH,
                                   "Interpolation failed: The provided curve must have a length of 65536, but it was " +
                                       std::to_string(curve_len));
            final_curve.assign(curve, curve + curve_len);  // stored as given; clamped at fetch (wgsl:188)
        } else {
            if (!ys)
                return Status::err(RPF_ERR_INTERP_MISSING_CONTROL_POINTS,
                                   "Interpolation failed: Control points for x and y must be provided together.");
            if (n_x != n_y)
                return Status::err(RPF_ERR_INTERP_MISMATCHED_LENGTHS,
                                   "Interpolation failed: Input arrays must have the same length: x has " +
                                       std::to_string(n_x) + ", y has " + std::to_string(n_y));
            if (n_x == 0)
                return Status::err(RPF_ERR_INTERP_EMPTY_CONTROL_POINTS,
                                   "Interpolation failed: Control point arrays cannot be empty.");
            std::vector<int32_t> x_eval(RPF_CURVE_RESOLUTION);
            for (int i = 0; i < RPF_CURVE_RESOLUTION; ++i) x_eval[i] = i;
            final_curve.resize(RPF_CURVE_RESOLUTION);
            Status st = pchip_interpolate_i32(xs, n_x, ys, n_y, x_eval.data(), x_eval.size(), final_curve.data(),
                                              nullptr);
            if (!st) { st.msg = "Interpolation failed: " + st.msg; return st; }
            for (auto& v : final_curve) v = clamp_i32(v, 0, RPF_CURVE_RESOLUTION - 1);
        }
        Status st;
        EditParameters* a = find(mask_name, st);  // looked up last, as in lib.rs:341
        if (!a) return st;
        if (curve) {
            a->cp_x[which].clear();
            a->cp_y[which].clear();
            if (a->curve[which] == final_curve) return st;  // the same 65536 values again
        } else {
            a->cp_x[which].assign(xs, xs + n_x);
            a->cp_y[which].assign(ys, ys + n_y);
        }
        a->curve[which] = std::move(final_curve);
        a->curve_ver[which] = ++stamp;
        ++params_version;
        return st;
    }

    // lib.rs:481-499
    Status add_mask(const char* name, const float* mask)
    {
        if (!name || !mask) return Status::err(RPF_ERR_INVALID_ARGUMENT, "add_mask: null argument");
        const float mask_range = masks[0].params.mask_range;
        MaskState m;
        m.name = name;
        const size_t n = (size_t)width * height;
        m.binary.resize(n);
        for (size_t i = 0; i < n; ++i) m.binary[i] = mask[i] >= mask_range ? 1 : 0;
        masks.push_back(std::move(m));
        ++masks_version;
        ++params_version;
        return Status::ok();
    }
    Status remove_mask(const char* name)  // lib.rs:501-505
    {
        if (!name) return Status::err(RPF_ERR_INVALID_ARGUMENT, "remove_mask: null name");
        const std::string n = name;
        if (n != "main") {
            for (size_t i = masks.size(); i-- > 0;)
                if (masks[i].name == n) masks.erase(masks.begin() + (long)i);
            ++masks_version;
            ++params_version;
        }
        return Status::ok();
    }
    void reset()  // lib.rs:227-235 (the image half is the Editor's)
    {
        for (size_t i = masks.size(); i-- > 0;)
            if (masks[i].name != "main") masks.erase(masks.begin() + (long)i);
        for (auto& m : masks)
            if (m.name == "main") { m.params = EditParameters(); break; }
        ++masks_version;
        ++params_version;
    }
};

// 16-bit PPM header, web-ts/core/image.ts:146-170 (loadPpm): three lines, each ended by 0x0A and trimmed:
// "P6", "<width> <height>" (split on a single space), "65535".  Number() semantics are narrowed to plain
// unsigned decimal integers; anything else is a read failure instead of a NaN-sized image.
inline Status ppm16_parse_header(const uint8_t* file, size_t len, uint32_t* width, uint32_t* height, size_t* data_offset)
{
    size_t off = 0;
    auto read_line = [&](std::string& line) -> bool {
        const size_t start = off;
        while (off < len && file[off] != 0x0A) ++off;
        if (off >= len) return false;  // the reference would scan past the buffer
        line.assign(reinterpret_cast<const char*>(file) + start, off - start);
        ++off;
        const char* ws = " \t\r\n\v\f";
        const size_t a = line.find_first_not_of(ws);
        line = a == std::string::npos ? std::string() : line.substr(a, line.find_last_not_of(ws) - a + 1);
        return true;
    };
    auto parse_uint = [](const std::string& t, uint32_t* v) -> bool {
        if (t.empty() || t.size() > 9) return false;
        uint32_t acc = 0;
        for (char c : t) {
            if (c < '0' || c > '9') return false;
            acc = acc * 10 + (uint32_t)(c - '0');
        }
        *v = acc;
        return true;
    };
    const Status bad_header{RPF_ERR_FAILED_TO_READ_IMAGE, "Failed to read image: truncated PPM header"};
    std::string line;
    if (!file || !read_line(line)) return bad_header;
    if (line != "P6") return Status{RPF_ERR_FAILED_TO_READ_IMAGE, "Failed to read image: PPM format must be P6"};
    if (!read_line(line)) return bad_header;
    const size_t sp = line.find(' ');
    uint32_t w = 0, h = 0;
    if (sp == std::string::npos || !parse_uint(line.substr(0, sp), &w) || !parse_uint(line.substr(sp + 1), &h) || w == 0 ||
        h == 0)
        return Status{RPF_ERR_FAILED_TO_READ_IMAGE, "Failed to read image: PPM size line must be \"<width> <height>\""};
    if (!read_line(line)) return bad_header;
    uint32_t maxval = 0;
    if (!parse_uint(line, &maxval) || maxval != 65535)
        return Status{RPF_ERR_FAILED_TO_READ_IMAGE, "Failed to read image: Only 16bit PPM (max=65535) supported"};
    if ((len - off) / 6 < (size_t)w * h)
        return Status{RPF_ERR_FAILED_TO_READ_IMAGE, "Failed to read image: PPM payload shorter than width*height*6 bytes"};
    *width = w;
    *height = h;
    *data_offset = off;
    return Status{};
}

}  // namespace rpf
