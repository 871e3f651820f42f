// rpf_kernels.hip — hand-written gfx950 (CDNA4) kernels of the RAW development hot path.
//
//   k_develop<MODE_DEMOSAIC>  u16 Bayer -> linear RGBA32F          (BASELINE config 2)
//   k_develop<MODE_DEVELOP>   u16 Bayer -> display RGBA32F, fused  (BASELINE config 3: demosaic + WB +
//                              camera matrix + the whole of wgsl main in ONE pass: 2 B in, 16 B out per pixel)
//   k_apply                   RGBA32F -> RGBA32F, wgsl main only   (what the reference's wgpu pass does)
//
// Replaces rust/photo-editor/src/wgpu_shader.wgsl:265-337 (kernel) and the dispatch half of
// rust/photo-editor/src/gpu_image_processing.rs:214-400.  Not a translation of the WGSL
// 16x16 one-thread-per-texel pass:
//   * persistent workgroups (one per CU) walk 128-pixel-wide row tiles; a wavefront owns 64
//     consecutive columns of TWO rows, so every store instruction writes 1 KiB contiguous;
//   * the (TH+4) x 132 normalised CFA neighbourhood of the tile is staged once in LDS; each
//     lane reads its 6-row column and gets the horizontal taps of the 5x5 stencil from its
//     neighbours with DPP wavefront shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1);
//   * the 65536-entry brightness curve (as u16, 128 KiB) and the transcendental tables
//     (13 KiB) live in LDS for the lifetime of the workgroup, so curve gathers never
//     leave the CU;
//   * a vertical pixel pair always holds one R/B site and one G site, so both demosaic
//     formula sets run without divergence.
// Pointwise/stencil path: HBM-bound by construction, no MFMA.
#include <hip/hip_runtime.h>
#include <string.h>

#include "rpf_launch.h"

namespace rpf {

enum { MODE_DEMOSAIC = 0, MODE_DEVELOP = 1 };

constexpr int TILE_W = 128;          // two wavefronts side by side
constexpr int PITCH = TILE_W + 4;    // 2-sample apron left and right

struct DevelopArgs {
    const uint16_t* cfa;   // row 0 = frame row y0 - 2
    TileGeom g;
    void* out;
    RawParamsDev raw;
    DeviceEdit ed;
    uint32_t tiles_x, ntiles;
};

// ---------------------------------------------------------------------------
// curve access (the Luts policy of edit_chain)
// ---------------------------------------------------------------------------
template <bool LDS_LUT>
struct DevLuts {
    const uint16_t* lds_bright;  // mask 0's brightness curve in LDS (LDS_LUT)
    const uint16_t* g_bright;
    const ushort4* g_hsl;
    __device__ __forceinline__ uint32_t brightness(uint32_t m, uint32_t i) const
    {
        if (LDS_LUT && m == 0) return lds_bright[i];
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
// DPP wavefront shifts (gfx9 family: wave_shr:1 = 0x138, wave_shl:1 = 0x130)
// ---------------------------------------------------------------------------
// lane L receives v of lane L-1; lane 0 keeps `edge`
__device__ __forceinline__ float from_left(float v, float edge)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
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
        reinterpret_cast<float4*>(out)[pix] = make_float4(r, g, b, 1.0f);
    } else if (OUTFMT == OUT_RGB_F32) {  // image.rs:202-277
        float* o = reinterpret_cast<float*>(out) + pix * 3;
        o[0] = r; o[1] = g; o[2] = b;
    } else {  // image.rs:375-383: (v.clamp(0,1) * 255.0) as u8
        uint8_t* o = reinterpret_cast<uint8_t*>(out) + pix * 3;
        o[0] = (uint8_t)(clampf(r, 0.0f, 1.0f) * 255.0f);
        o[1] = (uint8_t)(clampf(g, 0.0f, 1.0f) * 255.0f);
        o[2] = (uint8_t)(clampf(b, 0.0f, 1.0f) * 255.0f);
    }
}

__device__ __forceinline__ uint32_t active_masks(const DeviceEdit& ed, size_t pix, size_t plane)
{
    uint32_t active = 1u;  // "main" is all ones by construction (lib.rs:193-196)
    for (uint32_t m = 1; m < ed.num_masks; ++m)
        active |= (uint32_t)(ed.mask_planes[(size_t)(m - 1) * plane + pix] == 1) << m;
    return active;
}

// ---------------------------------------------------------------------------
// k_develop
// ---------------------------------------------------------------------------
template <int NT, int MODE, bool LDS_LUT, bool DPP, int OUTFMT>
__global__ __launch_bounds__(NT) void k_develop(const DevelopArgs a)
{
    constexpr int PAIRS = NT / TILE_W;   // row pairs per tile
    constexpr int TILE_H = 2 * PAIRS;
    constexpr int TROWS = TILE_H + 4;
    constexpr int TILE_ELEMS = TROWS * PITCH;

    extern __shared__ __align__(16) unsigned char smem[];
    float* tile = reinterpret_cast<float*>(smem);
    MathTables* T = reinterpret_cast<MathTables*>(smem + ((TILE_ELEMS * 4 + 15) & ~15));
    uint16_t* lut = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(T) + sizeof(MathTables));

    const int tid = threadIdx.x;
    const int W = (int)a.g.width, H = (int)a.g.height;
    const int y0 = (int)a.g.y0, rows = (int)a.g.rows;

    if (MODE == MODE_DEVELOP) {
        // stage the transcendental tables (and mask 0's brightness curve) once per workgroup
        const uint4* src = reinterpret_cast<const uint4*>(a.ed.tabs);
        uint4* dst = reinterpret_cast<uint4*>(T);
        for (int i = tid; i < (int)(sizeof(MathTables) / 16); i += NT) dst[i] = src[i];
        if (LDS_LUT) {
            const uint4* ls = reinterpret_cast<const uint4*>(a.ed.bright);
            uint4* ld = reinterpret_cast<uint4*>(lut);
            for (int i = tid; i < RPF_CURVE_RESOLUTION * 2 / 16; i += NT) ld[i] = ls[i];
        }
    }

    const int lx = tid & (TILE_W - 1);
    const int ly = 2 * (tid / TILE_W);
    const int lane = tid & 63;
    DevLuts<LDS_LUT> luts{lut, a.ed.bright, a.ed.hsl};

    for (uint32_t t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const int tx0 = (int)(t % a.tiles_x) * TILE_W;
        const int ty0 = (int)(t / a.tiles_x) * TILE_H;  // tile-relative row of the band

        __syncthreads();  // previous tile fully consumed (and tables staged)
        for (int i = tid; i < TILE_ELEMS; i += NT) {
            const int r = i / PITCH, c = i - r * PITCH;
            int gx = reflect101(tx0 - 2 + c, W);
            gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);          // columns past a ragged tile: unused
            int gy = reflect101(y0 + ty0 - 2 + r, H);
            int br = gy - (y0 - 2);                                // row of the band buffer
            br = br < 0 ? 0 : (br > rows + 3 ? rows + 3 : br);     // rows past a ragged tile: unused
            const uint32_t raw = a.cfa[(size_t)br * W + gx];
            const int cell = ((gx + a.raw.xo) & 1) + 2 * ((gy + a.raw.yo) & 1);
            tile[i] = cfa_normalise(raw, cell, a.raw);
        }
        __syncthreads();

        // 6-row column of this lane: tile rows ly .. ly+5 <-> frame rows (top - 2) .. (top + 3)
        const float* col = tile + ly * PITCH + lx + 2;
        float v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = col[i * PITCH];

        float p1[6], p2[6];  // p1[i] = W + E at distance 1 on tile row ly+i, p2 at distance 2
        if (DPP) {
            // lanes 0 / 63 fetch the two samples beyond the wavefront's 64 columns, per row
            const int hoff = lane == 0 ? -2 : (lane == 63 ? 1 : -(lx & 1));
#pragma unroll
            for (int i = 1; i <= 4; ++i) {
                const float2 hv = *reinterpret_cast<const float2*>(col + i * PITCH + hoff);
                const float l1 = from_left(v[i], hv.y);    // x-1 (lane 0: column -1 of the wavefront)
                const float r1 = from_right(v[i], hv.x);   // x+1 (lane 63: column 64)
                p1[i] = l1 + r1;
                if (i == 2 || i == 3) {
                    const float l2 = from_left(l1, hv.x);   // x-2
                    const float r2 = from_right(r1, hv.y);  // x+2
                    p2[i] = l2 + r2;
                }
            }
        } else {
#pragma unroll
            for (int i = 1; i <= 4; ++i) {
                p1[i] = col[i * PITCH - 1] + col[i * PITCH + 1];
                if (i == 2 || i == 3) p2[i] = col[i * PITCH - 2] + col[i * PITCH + 2];
            }
        }

        const int gx = tx0 + lx;
        const int ry_top = ty0 + ly;               // band-relative row of the top pixel
        const int gy_top = y0 + ry_top;
        const int cx = (gx + a.raw.xo) & 1;
        const int cy_top = (gy_top + a.raw.yo) & 1;
        const bool topA = cx == cy_top;            // top pixel is the R/B site, bottom the G site

        // neighbourhood sums: top pixel = tile row ly+2, bottom = ly+3
        const float c_t = v[2], s1_t = v[1] + v[3], e1_t = p1[2], d_t = p1[1] + p1[3], v2_t = v[0] + v[4], h2_t = p2[2];
        const float c_b = v[3], s1_b = v[2] + v[4], e1_b = p1[3], d_b = p1[2] + p1[4], v2_b = v[1] + v[5], h2_b = p2[3];

        const float cA = topA ? c_t : c_b, s1A = topA ? s1_t : s1_b, e1A = topA ? e1_t : e1_b;
        const float dA = topA ? d_t : d_b, v2A = topA ? v2_t : v2_b, h2A = topA ? h2_t : h2_b;
        const float cG = topA ? c_b : c_t, s1G = topA ? s1_b : s1_t, e1G = topA ? e1_b : e1_t;
        const float dG = topA ? d_b : d_t, v2G = topA ? v2_b : v2_t, h2G = topA ? h2_b : h2_t;

        const float cross = mhc_cross(cA, s1A, e1A, v2A, h2A);
        const float diag = mhc_diag(cA, dA, v2A, h2A);
        const float horiz = mhc_axis(cG, e1G, dG, h2G, v2G);
        const float vert = mhc_axis(cG, s1G, dG, v2G, h2G);

        // A site: cx == cy -> R site when cx == 0, B site when cx == 1
        const float RA = cx == 0 ? cA : diag, GA = cross, BA = cx == 0 ? diag : cA;
        // G site: its row parity is the opposite of the A site's (which is cx)
        const int cyG = cx ^ 1;
        const float RG = cyG == 0 ? horiz : vert, GG = cG, BG = cyG == 0 ? vert : horiz;

        float px[2][3];
        cam_to_linear(a.raw.cam, topA ? RA : RG, topA ? GA : GG, topA ? BA : BG, px[0][0], px[0][1], px[0][2]);
        cam_to_linear(a.raw.cam, topA ? RG : RA, topA ? GG : GA, topA ? BG : BA, px[1][0], px[1][1], px[1][2]);

        const bool in_x = gx < W;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ry = ry_top + k;
            const bool live = in_x && ry < rows;
            const size_t pix = (size_t)(live ? ry : 0) * W + (live ? gx : 0);
            float r = px[k][0], g = px[k][1], b = px[k][2];
            if (MODE == MODE_DEVELOP) {
                if (a.ed.vign_strength != 0.0f) {  // wgsl:166-178, global coordinates
                    const float gain = vignette_gain(a.ed.vign_strength, gx, y0 + ry, (float)W, (float)H);
                    r *= gain; g *= gain; b *= gain;
                }
                const uint32_t active = active_masks(a.ed, pix, (size_t)rows * W);
                float o0, o1, o2;
                edit_chain(r, g, b, active, a.ed.mp, a.ed.num_masks, luts, *T, o0, o1, o2);
                r = o0; g = o1; b = o2;
            }
            if (live) store_px<OUTFMT>(a.out, pix, r, g, b);
        }
    }
}

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

template <int NT, bool LDS_LUT, bool LENS, int OUTFMT>
__global__ __launch_bounds__(NT) void k_apply(const ApplyArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    MathTables* T = reinterpret_cast<MathTables*>(smem);
    uint16_t* lut = reinterpret_cast<uint16_t*>(smem + sizeof(MathTables));
    const int tid = threadIdx.x;
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.ed.tabs);
        uint4* dst = reinterpret_cast<uint4*>(T);
        for (int i = tid; i < (int)(sizeof(MathTables) / 16); i += NT) dst[i] = src[i];
        if (LDS_LUT) {
            const uint4* ls = reinterpret_cast<const uint4*>(a.ed.bright);
            uint4* ld = reinterpret_cast<uint4*>(lut);
            for (int i = tid; i < RPF_CURVE_RESOLUTION * 2 / 16; i += NT) ld[i] = ls[i];
        }
    }
    __syncthreads();
    DevLuts<LDS_LUT> luts{lut, a.ed.bright, a.ed.hsl};
    const int W = (int)a.g.width, H = (int)a.g.height;
    const size_t npix = (size_t)a.g.rows * W;
    const size_t stride = (size_t)gridDim.x * NT * 2;
    for (size_t base = (size_t)blockIdx.x * NT * 2; base < npix; base += stride) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const size_t pix = base + (size_t)k * NT + tid;
            if (pix >= npix) continue;
            const int ry = (int)(pix / W), x = (int)(pix - (size_t)ry * W);
            const int gy = (int)a.g.y0 + ry;
            float r, g, b;
            if (LENS) {  // wgsl:123-163; a.in is the whole frame
                lens_sample(TexFetch{a.in, W}, W, H, x, gy, a.ed.lens_strength, r, g, b);
            } else {
                const float4 v = a.in[pix];
                r = v.x; g = v.y; b = v.z;
            }
            if (a.ed.vign_strength != 0.0f) {
                const float gain = vignette_gain(a.ed.vign_strength, x, gy, (float)W, (float)H);
                r *= gain; g *= gain; b *= gain;
            }
            const uint32_t active = active_masks(a.ed, pix, npix);
            float o0, o1, o2;
            edit_chain(r, g, b, active, a.ed.mp, a.ed.num_masks, luts, *T, o0, o1, o2);
            store_px<OUTFMT>(a.out, pix, o0, o1, o2);
        }
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

__global__ __launch_bounds__(256) void k_pad_rgb(const float* in, size_t npix, float4* out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256)
        out[i] = make_float4(in[i * 3], in[i * 3 + 1], in[i * 3 + 2], 1.0f);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
template <int NT>
constexpr size_t develop_lds(bool develop, bool lds_lut)
{
    return (size_t)((((2 * (NT / TILE_W) + 4) * PITCH * 4) + 15) & ~15) + (develop ? sizeof(MathTables) : 0) +
           (lds_lut ? (size_t)RPF_CURVE_RESOLUTION * 2 : 0);
}

template <class K>
static hipError_t set_lds(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

template <int NT, int MODE, bool LDS_LUT, bool DPP, int OUTFMT>
static hipError_t run_develop(const LaunchCfg& cfg, DevelopArgs& a, int blocks_per_cu, hipStream_t s)
{
    constexpr int TILE_H = 2 * (NT / TILE_W);
    a.tiles_x = (a.g.width + TILE_W - 1) / TILE_W;
    const uint32_t tiles_y = (a.g.rows + TILE_H - 1) / TILE_H;
    a.ntiles = a.tiles_x * tiles_y;
    if (a.ntiles == 0) return hipSuccess;
    const size_t lds = develop_lds<NT>(MODE == MODE_DEVELOP, LDS_LUT);
    auto kernel = k_develop<NT, MODE, LDS_LUT, DPP, OUTFMT>;
    hipError_t e = set_lds(kernel, lds);
    if (e != hipSuccess) return e;
    uint32_t grid = (uint32_t)cfg.num_cus * (uint32_t)blocks_per_cu;
    if (grid > a.ntiles) grid = a.ntiles;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(NT), lds, s, a);
    return hipGetLastError();
}

static bool needs_lds_lut(const DeviceEdit& ed) { return !(ed.mp[0].flags & MF_BRIGHT_IDENTITY); }

// Operand checks shared by the launchers: a tile must be non-empty, inside its frame and large
// enough for the reflect-101 border; a 128-KiB LDS curve needs one workgroup per CU.
static hipError_t check_geom(const TileGeom& g, bool bayer)
{
    if (g.width == 0 || g.height == 0 || g.rows == 0) return hipErrorInvalidValue;
    if ((uint64_t)g.y0 + g.rows > g.height) return hipErrorInvalidValue;
    if (bayer && (g.width < 3 || g.height < 3)) return hipErrorInvalidValue;
    if ((uint64_t)g.width * g.rows > (1ull << 31)) return hipErrorInvalidValue;
    return hipSuccess;
}

hipError_t launch_demosaic(const LaunchCfg& cfg, const uint16_t* d_cfa, const RawParamsDev& raw, const TileGeom& g,
                           float* d_rgba, hipStream_t s)
{
    hipError_t e = check_geom(g, true);
    if (e != hipSuccess) return e;
    if (!d_cfa || !d_rgba) return hipErrorInvalidValue;
    DevelopArgs a;
    memset(&a, 0, sizeof(a));
    a.cfa = d_cfa; a.g = g; a.out = d_rgba; a.raw = raw;
    if (cfg.variant == 1) return run_develop<512, MODE_DEMOSAIC, false, false, OUT_RGBA_F32>(cfg, a, 4, s);
    return run_develop<512, MODE_DEMOSAIC, false, true, OUT_RGBA_F32>(cfg, a, 4, s);
}

template <int OUTFMT>
static hipError_t develop_fmt(const LaunchCfg& cfg, DevelopArgs& a, hipStream_t s)
{
    const bool lut = needs_lds_lut(a.ed);
    // variant 1: no DPP (every stencil tap from LDS); variant 2: 1024-thread workgroups
    if (cfg.variant == 1)
        return lut ? run_develop<512, MODE_DEVELOP, true, false, OUTFMT>(cfg, a, 1, s)
                   : run_develop<512, MODE_DEVELOP, false, false, OUTFMT>(cfg, a, 2, s);
    if (cfg.variant == 2)
        return lut ? run_develop<1024, MODE_DEVELOP, true, true, OUTFMT>(cfg, a, 1, s)
                   : run_develop<1024, MODE_DEVELOP, false, true, OUTFMT>(cfg, a, 1, s);
    return lut ? run_develop<512, MODE_DEVELOP, true, true, OUTFMT>(cfg, a, 1, s)
               : run_develop<512, MODE_DEVELOP, false, true, OUTFMT>(cfg, a, 2, s);
}

hipError_t launch_develop(const LaunchCfg& cfg, const uint16_t* d_cfa, const RawParamsDev& raw, const TileGeom& g,
                          const DeviceEdit& ed, void* d_out, int out_format, hipStream_t s)
{
    hipError_t e = check_geom(g, true);
    if (e != hipSuccess) return e;
    if (!d_cfa || !d_out || !ed.tabs || !ed.bright || !ed.hsl || ed.num_masks < 1 || ed.num_masks > RPF_MAX_MASKS)
        return hipErrorInvalidValue;
    if (ed.num_masks > 1 && !ed.mask_planes) return hipErrorInvalidValue;
    if (ed.lens_strength != 0.0f) return hipErrorInvalidValue;  // needs the resampling pass (launch_apply)
    DevelopArgs a;
    memset(&a, 0, sizeof(a));
    a.cfa = d_cfa; a.g = g; a.out = d_out; a.raw = raw; a.ed = ed;
    switch (out_format) {
    case OUT_RGBA_F32: return develop_fmt<OUT_RGBA_F32>(cfg, a, s);
    case OUT_RGB_F32: return develop_fmt<OUT_RGB_F32>(cfg, a, s);
    case OUT_RGB_U8: return develop_fmt<OUT_RGB_U8>(cfg, a, s);
    default: return hipErrorInvalidValue;
    }
}

template <bool LDS_LUT, bool LENS, int OUTFMT>
static hipError_t run_apply(const LaunchCfg& cfg, const ApplyArgs& a, hipStream_t s)
{
    constexpr int NT = 512;
    const size_t lds = sizeof(MathTables) + (LDS_LUT ? (size_t)RPF_CURVE_RESOLUTION * 2 : 0);
    auto kernel = k_apply<NT, LDS_LUT, LENS, OUTFMT>;
    hipError_t e = set_lds(kernel, lds);
    if (e != hipSuccess) return e;
    const size_t npix = (size_t)a.g.rows * a.g.width;
    size_t grid = (size_t)cfg.num_cus * (LDS_LUT ? 1 : 2);
    const size_t need = (npix + NT * 2 - 1) / (NT * 2);
    if (grid > need) grid = need;
    hipLaunchKernelGGL(kernel, dim3((uint32_t)grid), dim3(NT), lds, s, a);
    return hipGetLastError();
}

template <int OUTFMT>
static hipError_t apply_fmt(const LaunchCfg& cfg, const ApplyArgs& a, hipStream_t s)
{
    const bool lut = needs_lds_lut(a.ed);
    const bool lens = a.ed.lens_strength != 0.0f;
    if (lens) return lut ? run_apply<true, true, OUTFMT>(cfg, a, s) : run_apply<false, true, OUTFMT>(cfg, a, s);
    return lut ? run_apply<true, false, OUTFMT>(cfg, a, s) : run_apply<false, false, OUTFMT>(cfg, a, s);
}

hipError_t launch_apply(const LaunchCfg& cfg, const float* d_in, const TileGeom& g, const DeviceEdit& ed, void* d_out,
                        int out_format, hipStream_t s)
{
    hipError_t e = check_geom(g, false);
    if (e != hipSuccess) return e;
    if (!d_in || !d_out || !ed.tabs || !ed.bright || !ed.hsl || ed.num_masks < 1 || ed.num_masks > RPF_MAX_MASKS)
        return hipErrorInvalidValue;
    if (ed.num_masks > 1 && !ed.mask_planes) return hipErrorInvalidValue;
    // the resampling branch reads anywhere in the frame: the tile must be the whole frame
    if (ed.lens_strength != 0.0f && (g.y0 != 0 || g.rows != g.height)) return hipErrorInvalidValue;
    if (ed.lens_strength != 0.0f && d_in == d_out) return hipErrorInvalidValue;
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

hipError_t launch_pad_rgb(const float* d_rgb, size_t npix, float* d_rgba, hipStream_t s)
{
    if (npix == 0) return hipSuccess;
    if (!d_rgb || !d_rgba) return hipErrorInvalidValue;
    const uint32_t grid = (uint32_t)((npix + 255) / 256 > 4096 ? 4096 : (npix + 255) / 256);
    hipLaunchKernelGGL(k_pad_rgb, dim3(grid), dim3(256), 0, s, d_rgb, npix, reinterpret_cast<float4*>(d_rgba));
    return hipGetLastError();
}

}  // namespace rpf
