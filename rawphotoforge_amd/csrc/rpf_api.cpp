// rpf_api.cpp — the extern "C" boundary (include/rpf.h): GpuProcessor / PhotoEditor of the
// reference's photo-editor crate (rust/photo-editor/src/lib.rs, gpu_image_processing.rs,
// image.rs) re-stated over HIP.  No torch, no CPU compute path: every pixel is produced by
// the gfx950 kernels in rpf_kernels.hip or not at all.
#include <hip/hip_runtime_api.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "rpf_internal.h"

namespace rpf {
void build_math_tables(MathTables& T);
void build_legacy_tables(LegacyTables& T);
void build_hue_bounds(HueBound* B, HueBoundCs* Cs, const MathTables& T);
void build_fast_tables(FastTables& T);
}

using namespace rpf;

static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}
static int fail(const Status& st) { return fail(st.code, st.msg); }
static int ok()
{
    return RPF_OK;
}
static int hip_fail(hipError_t e, const char* what, int code = RPF_ERR_GPU_COMPUTE)
{
    // errors.rs:35-39 "GPU compute error" / :41-45 "GPU initialization error"
    return fail(code, std::string(code == RPF_ERR_GPU_INITIALIZATION ? "GPU initialization error: "
                                                                      : "GPU compute error: ") +
                          what + ": " + hipGetErrorString(e));
}
// No C++ exception may cross the C boundary: host-side state (curves, masks, strings) allocates.
static int fail_exception() noexcept
{
    try {
        throw;
    } catch (const std::bad_alloc&) {
        try { g_last_error = "GPU compute error: out of host memory"; } catch (...) {}
        return RPF_ERR_GPU_COMPUTE;  // a resource failure, not a caller mistake
    } catch (const std::exception& ex) {
        try { g_last_error = std::string("internal error: ") + ex.what(); } catch (...) {}
        return RPF_ERR_GPU_COMPUTE;
    } catch (...) {
        return RPF_ERR_GPU_COMPUTE;
    }
}
namespace rpf {
int api_fail(int code, const std::string& msg) { return fail(code, msg); }
int api_hip_fail(hipError_t e, const char* what) { return hip_fail(e, what); }
}  // namespace rpf
#define RPF_GUARDED(...)   \
    try {                  \
        __VA_ARGS__        \
    } catch (...) {        \
        return fail_exception(); \
    }

#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return hip_fail(_e, #expr);    \
    } while (0)

static void processor_unref(rpf_processor* p)
{
    if (p->refs.fetch_sub(1) != 1) return;
    if (hipSetDevice(p->device) == hipSuccess) {
        if (p->d_tabs) (void)hipFree(p->d_tabs);
        if (p->d_legacy) (void)hipFree(p->d_legacy);
        if (p->d_fast_tabs) (void)hipFree(p->d_fast_tabs);
        if (p->d_hue_bounds) (void)hipFree(p->d_hue_bounds);
        if (p->d_hue_cs) (void)hipFree(p->d_hue_cs);
    }
    delete p;
}

namespace rpf {
void api_processor_unref(rpf_processor* p) { processor_unref(p); }
}  // namespace rpf

static size_t npix_of(const rpf_editor* e) { return (size_t)e->geom.rows * e->geom.width; }

namespace rpf {
int api_wait_timeout_ms()
{
    static const int ms = [] {
        const char* v = getenv("RPF_WAIT_TIMEOUT_MS");
        return v && *v ? atoi(v) : 30000;
    }();
    return ms;
}

int api_editor_quiesce(rpf_editor* e)
{
    {   // steps a shard's thread still holds (with copies of this editor's table pointers): enqueued first
        std::unique_lock<std::mutex> lock(e->qmu);
        const int ms = api_wait_timeout_ms();
        auto idle = [&] { return e->queued == 0; };
        if (ms > 0) {
            if (!e->qcv.wait_for(lock, std::chrono::milliseconds(ms), idle))
                return fail(RPF_ERR_GPU_COMPUTE, "GPU compute error: " + std::to_string(e->queued) + " step(s) handed to rpf_shard_develop_async "
                            "were not enqueued within " + std::to_string(ms) + " ms: the editor's tables are still referenced "
                            "(rpf_shard_flush reports what the shard's thread is waiting for)");
        } else {
            e->qcv.wait(lock, idle);
        }
    }
    if (e->foreign_streams) HIP_TRY(hipDeviceSynchronize());   // launches went to streams of the caller's
    else if (e->stream) HIP_TRY(hipStreamSynchronize(e->stream));
    return ok();
}
}  // namespace rpf

// Refresh the device copy of params / curves / masks if a setter ran since the last upload.
static int sync_edit_state_impl(rpf_editor* e);
static int sync_edit_state(rpf_editor* e)
{
    RPF_GUARDED(return sync_edit_state_impl(e);)
}
namespace rpf {
int api_sync_edit_state(rpf_editor* e) { return sync_edit_state(e); }
}  // namespace rpf
static int sync_edit_state_impl(rpf_editor* e)
{
    const uint32_t n = (uint32_t)e->state.masks.size();
    e->dev.precision = (uint32_t)e->precision;  // not versioned state: taken at every launch, like the shader flavour
    e->dev.fast_tabs = e->proc->d_fast_tabs;
    // The vignette's falloff depends on the pixel's position alone: one map per editor (4 B per pixel), computed when the
    // slider first leaves 0 (active shader flavour: the Godot-legacy falloff is another function and is evaluated in
    // place), read by the kernels instead of a square root, a quotient and a pow(t, 1.5) per pixel.
    if (e->state.masks[0].params.vignette != 0 && e->flavour == RPF_FLAVOUR_ACTIVE && !e->d_vign_falloff && npix_of(e) > 0 &&
        npix_of(e) * sizeof(float) <= 0xffffffffull) {
        float* map = nullptr;  // the editor's pointer is set only once the map is complete: a failed build leaves no map behind
        HIP_TRY(hipMalloc((void**)&map, npix_of(e) * sizeof(float)));
        hipError_t he = launch_vignette_falloff(e->geom, map, e->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);  // rpf_dev_* callers launch on their own streams
        if (he != hipSuccess) { (void)hipFree(map); return hip_fail(he, "vignette falloff map"); }
        e->d_vign_falloff = map;
        e->dev.vign_falloff = e->d_vign_falloff;   // (vign_strength != 0 was or is being uploaded with this very state)
    }
    if (e->uploaded_params == e->state.params_version && e->uploaded_masks == e->state.masks_version) return ok();
    const bool masks_changed = e->uploaded_masks != e->state.masks_version;
    bool synced = false;  // kernels launched earlier (on this editor's stream or a caller's) may still read the old tables
    auto quiesce = [&]() -> int {
        if (!synced)
            if (int qrc = api_editor_quiesce(e)) return qrc;
        synced = true;
        return ok();
    };
    int rc;
    if (e->lut_capacity < n) {
        if ((rc = quiesce())) return rc;
        if (e->d_bright) HIP_TRY(hipFree(e->d_bright));
        if (e->d_hsl) HIP_TRY(hipFree(e->d_hsl));
        e->d_bright = nullptr; e->d_hsl = nullptr;
        HIP_TRY(hipMalloc((void**)&e->d_bright, (size_t)n * RPF_CURVE_RESOLUTION * sizeof(uint16_t)));
        HIP_TRY(hipMalloc((void**)&e->d_hsl, (size_t)n * RPF_CURVE_RESOLUTION * sizeof(ushort4)));
        e->lut_capacity = n;
        e->uploaded_curves.clear();
    }
    // a mask list that changed (add / remove / reset) may have moved every slot: upload all of them again
    if (masks_changed) e->uploaded_curves.clear();
    e->uploaded_curves.resize(n, CurveStamps{{~0ull, ~0ull, ~0ull, ~0ull}});
    // only the curves a setter really changed go to the device (the reference re-uploads 1 MiB per mask per apply,
    // gpu_image_processing.rs:257-307, and its callers re-send all four curves every frame, web/main.ts:809-826)
    std::vector<uint16_t> bright;
    std::vector<ushort4> hsl;
    std::vector<MaskParams> all_mp(n);
    for (uint32_t m = 0; m < n; ++m) {
        const EditParameters& p = e->state.masks[m].params;
        all_mp[m] = make_mask_params(p);
        if (m < RPF_MASKS_PER_WORD) e->dev.mp[m] = all_mp[m];
        CurveStamps& up = e->uploaded_curves[m];
        // an identity brightness curve / trivial hue-saturation-lightness curves are never read by the kernels
        // (make_mask_params flags them): nothing to pack or upload for them
        const bool bright_read = !(all_mp[m].flags & MF_BRIGHT_IDENTITY);
        const bool hsl_read = (all_mp[m].flags & (MF_HUE_IDENTITY | MF_SAT_CONST | MF_LIGHT_CONST)) !=
                              (MF_HUE_IDENTITY | MF_SAT_CONST | MF_LIGHT_CONST);
        if (bright_read && up.v[RPF_CURVE_BRIGHTNESS] != p.curve_ver[RPF_CURVE_BRIGHTNESS]) {
            bright.resize(RPF_CURVE_RESOLUTION);
            for (int i = 0; i < RPF_CURVE_RESOLUTION; ++i) bright[i] = (uint16_t)lut_clamp(p.curve[RPF_CURVE_BRIGHTNESS][i]);  // wgsl:188
            if ((rc = quiesce())) return rc;
            HIP_TRY(hipMemcpy(e->d_bright + (size_t)m * RPF_CURVE_RESOLUTION, bright.data(),
                              bright.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            up.v[RPF_CURVE_BRIGHTNESS] = p.curve_ver[RPF_CURVE_BRIGHTNESS];
        }
        if (hsl_read &&
            (up.v[RPF_CURVE_HUE] != p.curve_ver[RPF_CURVE_HUE] || up.v[RPF_CURVE_SATURATION] != p.curve_ver[RPF_CURVE_SATURATION] ||
             up.v[RPF_CURVE_LIGHTNESS] != p.curve_ver[RPF_CURVE_LIGHTNESS])) {
            hsl.resize(RPF_CURVE_RESOLUTION);
            for (int i = 0; i < RPF_CURVE_RESOLUTION; ++i) {
                hsl[i].x = (uint16_t)lut_clamp(p.curve[RPF_CURVE_HUE][i]);
                hsl[i].y = (uint16_t)lut_clamp(p.curve[RPF_CURVE_SATURATION][i]);
                hsl[i].z = (uint16_t)lut_clamp(p.curve[RPF_CURVE_LIGHTNESS][i]);
                hsl[i].w = 0;
            }
            if ((rc = quiesce())) return rc;
            HIP_TRY(hipMemcpy(e->d_hsl + (size_t)m * RPF_CURVE_RESOLUTION, hsl.data(), hsl.size() * sizeof(ushort4),
                              hipMemcpyHostToDevice));
            for (int c = RPF_CURVE_HUE; c <= RPF_CURVE_LIGHTNESS; ++c) up.v[c] = p.curve_ver[c];
        }
    }
    // More masks than the kernel-argument segment holds parameters for (the reference has no limit, lib.rs:481-499): all
    // of them in device memory, read by the RGBA pass's general kernel (rpf_launch.h DeviceEdit::mp_wide).
    if (n > RPF_MASKS_PER_WORD) {
        if ((rc = quiesce())) return rc;
        if (e->mp_wide_capacity < n) {
            if (e->d_mp_wide) HIP_TRY(hipFree(e->d_mp_wide));
            e->d_mp_wide = nullptr; e->mp_wide_capacity = 0;
            HIP_TRY(hipMalloc((void**)&e->d_mp_wide, (size_t)n * sizeof(MaskParams)));
            e->mp_wide_capacity = n;
        }
        HIP_TRY(hipMemcpy(e->d_mp_wide, all_mp.data(), (size_t)n * sizeof(MaskParams), hipMemcpyHostToDevice));
    }
    e->dev.mp_wide = n > RPF_MASKS_PER_WORD ? e->d_mp_wide : nullptr;
    // One mask whose hue / saturation / lightness curves are not all trivial: fold them into the editor's fused hue table
    // (one gather per pixel instead of two and sincos_hue); rebuilt only when what it is made of changed.
    {
        const MaskParams& P0 = e->dev.mp[0];
        const EditParameters& p0 = e->state.masks[0].params;
        const bool trivial = (P0.flags & (MF_HUE_IDENTITY | MF_SAT_CONST | MF_LIGHT_CONST)) == (MF_HUE_IDENTITY | MF_SAT_CONST | MF_LIGHT_CONST);
        bool use = n == 1 && !trivial && e->d_hsl;
        if (use) {
            const bool same = e->hue_fused_valid && e->hue_fused_flags == P0.flags && e->hue_fused_gains[0] == P0.sat_gain &&
                              e->hue_fused_gains[1] == P0.light_gain && e->hue_fused_stamp[0] == p0.curve_ver[RPF_CURVE_HUE] &&
                              e->hue_fused_stamp[1] == p0.curve_ver[RPF_CURVE_SATURATION] &&
                              e->hue_fused_stamp[2] == p0.curve_ver[RPF_CURVE_LIGHTNESS];
            if (!same) {
                if ((rc = quiesce())) return rc;
                if (!e->d_hue_fused_alloc)
                    HIP_TRY(hipMalloc(&e->d_hue_fused_alloc, (size_t)RPF_CURVE_RESOLUTION * sizeof(HueFused)));
                HueFused* t = static_cast<HueFused*>(e->d_hue_fused_alloc);
                HIP_TRY(launch_build_hue_fused(e->proc->d_hue_bounds, e->d_hsl, P0, e->proc->d_tabs, t, e->stream));
                HIP_TRY(hipStreamSynchronize(e->stream));  // rpf_dev_* callers launch on their own streams
                e->hue_fused_valid = true;
                e->hue_fused_flags = P0.flags;
                e->hue_fused_gains[0] = P0.sat_gain; e->hue_fused_gains[1] = P0.light_gain;
                e->hue_fused_stamp[0] = p0.curve_ver[RPF_CURVE_HUE];
                e->hue_fused_stamp[1] = p0.curve_ver[RPF_CURVE_SATURATION];
                e->hue_fused_stamp[2] = p0.curve_ver[RPF_CURVE_LIGHTNESS];
            }
        }
        e->dev.hue_fused = use ? static_cast<const HueFused*>(e->d_hue_fused_alloc) : nullptr;
    }
    const rpf_gpu_edit_parameters g0 = convert_params(e->state.masks[0].params);
    e->dev.vign_strength = (-g0.vignette / 100.0f) * 2.0f;     // wgsl:167 (main mask only, :276)
    e->dev.lens_strength = -0.5f * (g0.lens_distortion / 100.0f);  // wgsl:110 (main mask only, :273)
    e->dev.vign_falloff = e->dev.vign_strength != 0.0f ? e->d_vign_falloff : nullptr;
    e->dev.num_masks = n;
    e->dev.bright = e->d_bright;
    e->dev.hsl = e->d_hsl;
    e->dev.tabs = e->proc->d_tabs;
    e->dev.legacy = e->proc->d_legacy;
    e->dev.hue_bounds = e->proc->d_hue_bounds;
    e->dev.hue_cs = e->proc->d_hue_cs;
    e->uploaded_params = e->state.params_version;
    if (masks_changed) {
        const uint32_t planes = n - 1;
        const size_t plane = npix_of(e);
        if (planes > e->mask_capacity) {
            if ((rc = quiesce())) return rc;
            if (e->d_masks) HIP_TRY(hipFree(e->d_masks));
            e->d_masks = nullptr;
            HIP_TRY(hipMalloc((void**)&e->d_masks, plane * planes));
            e->mask_capacity = planes;
        }
        // one u16 of coverage bits per pixel and 16 masks: plane 0 = user masks 1..15 (bit j: mask j + 1), plane g >= 1 =
        // masks 16 g .. 16 g + 15 (rpf_chain.h WideMasks)
        const uint32_t bit_planes = planes ? (n + RPF_MASKS_PER_WORD - 1) / RPF_MASKS_PER_WORD : 0;
        if (planes && (rc = quiesce())) return rc;
        if (bit_planes > e->mask_bit_planes) {
            if (e->d_mask_bits) HIP_TRY(hipFree(e->d_mask_bits));
            e->d_mask_bits = nullptr; e->mask_bit_planes = 0;
            HIP_TRY(hipMalloc((void**)&e->d_mask_bits, plane * sizeof(uint16_t) * bit_planes));
            e->mask_bit_planes = bit_planes;
        }
        for (uint32_t m = 1; m < n; ++m)
            HIP_TRY(hipMemcpy(e->d_masks + (size_t)(m - 1) * plane, e->state.masks[m].binary.data(), plane,
                              hipMemcpyHostToDevice));
        // (after quiesce(): nothing is in flight; the editor's stream orders it before the next apply)
        for (uint32_t gp = 0; gp < bit_planes; ++gp) {
            const uint32_t first = gp ? gp * RPF_MASKS_PER_WORD : 1u;  // first mask of the plane
            const uint32_t last = (gp + 1) * RPF_MASKS_PER_WORD < n ? (gp + 1) * RPF_MASKS_PER_WORD : n;
            HIP_TRY(launch_combine_masks(e->d_masks + (size_t)(first - 1) * plane, last - first, plane,
                                         e->d_mask_bits + (size_t)gp * plane, e->stream));
        }
        if (planes) HIP_TRY(hipStreamSynchronize(e->stream));  // rpf_dev_* callers launch on their own streams
        e->dev.mask_bits = planes ? e->d_mask_bits : nullptr;
        e->uploaded_masks = e->state.masks_version;
    }
    return ok();
}

static uint32_t flavour_flags(const rpf_editor* e)
{
    if (e->flavour != RPF_FLAVOUR_GODOT_LEGACY) return 0u;
    return EF_LEGACY_VIGNETTE | (e->is_bayer ? 0u : EF_DECODE_INPUT);
}

static int need_device(const rpf_editor* e)
{
    if (!e) return fail(RPF_ERR_INVALID_ARGUMENT, "null editor");
    if (!e->proc)
        return fail(RPF_ERR_GPU_COMPUTE, "GPU compute error: headless editor has no device (no CPU fallback exists)");
    return ok();
}

static int ensure_image_buffer(rpf_editor* e)
{
    if (!e->d_image) HIP_TRY(hipMalloc((void**)&e->d_image, npix_of(e) * 4 * sizeof(float)));
    return ok();
}

// linear RGBA of a bayer editor (demosaic-only pass), cached
static int ensure_linear(rpf_editor* e)
{
    if (e->original_valid) return ok();
    if (!e->d_original) HIP_TRY(hipMalloc((void**)&e->d_original, npix_of(e) * 4 * sizeof(float)));
    HIP_TRY(launch_demosaic(cfg_of(e->proc), e->d_cfa, e->raw, e->geom, e->d_original, e->stream));
    e->original_valid = true;
    return ok();
}

extern "C" {

const char* rpf_last_error(void) { return g_last_error.c_str(); }
const char* rpf_version(void) { return "rawphotoforge_amd 0.1 (gfx950)"; }

int rpf_device_count(int* count)
{
    if (!count) return fail(RPF_ERR_INVALID_ARGUMENT, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return hip_fail(e, "hipGetDeviceCount", RPF_ERR_GPU_INITIALIZATION); }
    *count = n;
    return ok();
}

int rpf_device_name(int device, char* buf, size_t buf_len)
{
    if (!buf || buf_len == 0) return fail(RPF_ERR_INVALID_ARGUMENT, "null buffer");
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceProperties", RPF_ERR_GPU_INITIALIZATION);
    std::string s = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    strncpy(buf, s.c_str(), buf_len - 1);
    buf[buf_len - 1] = 0;
    return ok();
}

int rpf_processor_create(int device, rpf_processor** out)
{
    if (!out) return fail(RPF_ERR_INVALID_ARGUMENT, "null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount", RPF_ERR_GPU_INITIALIZATION);
    if (device < 0 || device >= n)  // gpu_image_processing.rs:48-51: adapters.nth(index) is None
        return fail(RPF_ERR_GPU_INITIALIZATION, "GPU initialization error: no device with index " +
                                                    std::to_string(device) + " (" + std::to_string(n) + " visible)");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceProperties", RPF_ERR_GPU_INITIALIZATION);
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(RPF_ERR_GPU_INITIALIZATION, std::string("GPU initialization error: kernels are built for gfx950, "
                                                            "device is ") + prop.gcnArchName);
    e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice", RPF_ERR_GPU_INITIALIZATION);
    std::unique_ptr<rpf_processor> p(new (std::nothrow) rpf_processor);
    if (!p) return fail(RPF_ERR_GPU_INITIALIZATION, "out of host memory");
    p->device = device;
    p->num_cus = prop.multiProcessorCount;
    std::unique_ptr<MathTables> tabs(new MathTables);
    build_math_tables(*tabs);
    e = hipMalloc((void**)&p->d_tabs, sizeof(MathTables));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(tables)", RPF_ERR_GPU_INITIALIZATION);
    e = hipMemcpy(p->d_tabs, tabs.get(), sizeof(MathTables), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(p->d_tabs); return hip_fail(e, "hipMemcpy(tables)", RPF_ERR_GPU_INITIALIZATION); }
    std::unique_ptr<LegacyTables> legacy(new LegacyTables);
    build_legacy_tables(*legacy);
    std::vector<HueBound> bounds(RPF_CURVE_RESOLUTION);
    std::vector<HueBoundCs> hcs(RPF_CURVE_RESOLUTION);
    build_hue_bounds(bounds.data(), hcs.data(), *tabs);
    e = hipMalloc((void**)&p->d_legacy, sizeof(LegacyTables));
    if (e == hipSuccess) e = hipMemcpy(p->d_legacy, legacy.get(), sizeof(LegacyTables), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_hue_bounds, bounds.size() * sizeof(HueBound));
    if (e == hipSuccess)
        e = hipMemcpy(p->d_hue_bounds, bounds.data(), bounds.size() * sizeof(HueBound), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_hue_cs, hcs.size() * sizeof(HueBoundCs));
    if (e == hipSuccess) e = hipMemcpy(p->d_hue_cs, hcs.data(), hcs.size() * sizeof(HueBoundCs), hipMemcpyHostToDevice);
    std::unique_ptr<FastTables> fast(new FastTables);
    build_fast_tables(*fast);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_fast_tabs, sizeof(FastTables));
    if (e == hipSuccess) e = hipMemcpy(p->d_fast_tabs, fast.get(), sizeof(FastTables), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(p->d_tabs);
        if (p->d_legacy) (void)hipFree(p->d_legacy);
        if (p->d_fast_tabs) (void)hipFree(p->d_fast_tabs);
        if (p->d_hue_bounds) (void)hipFree(p->d_hue_bounds);
        if (p->d_hue_cs) (void)hipFree(p->d_hue_cs);
        return hip_fail(e, "tables", RPF_ERR_GPU_INITIALIZATION);
    }
    *out = p.release();
    return ok();
}

void rpf_processor_destroy(rpf_processor* p)
{
    if (!p) return;
    processor_unref(p);  // editors still alive keep the tables until the last of them is destroyed
}

int rpf_processor_device(const rpf_processor* p, int* device)
{
    if (!p || !device) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *device = p->device;
    return ok();
}

int rpf_processor_compute_units(const rpf_processor* p, int* cus)
{
    if (!p || !cus) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *cus = p->num_cus;
    return ok();
}

int rpf_set_kernel_variant(rpf_processor* p, int variant)
{
    if (!p) return fail(RPF_ERR_INVALID_ARGUMENT, "null processor");
    p->variant = variant;
    return ok();
}

// ---- editors --------------------------------------------------------------------------------

static int editor_device_init(rpf_editor* e, rpf_processor* p)
{
    e->proc = p;
    p->refs.fetch_add(1);  // dropped by rpf_editor_destroy, also on the failure paths below
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    return ok();
}

void rpf_editor_destroy(rpf_editor* e)
{
    if (!e) return;
    if (e->proc && hipSetDevice(e->proc->device) == hipSuccess) {
        // Steps still queued in a shard's thread hold pointers into the tables freed below, and to this object.  Should they
        // not get enqueued within the deadline (a stalled exchange), the editor is LEAKED — tables, stream and all — rather
        // than freed under a thread that will still touch it; the caller has been told (rpf_shard_flush / _drain failed).
        if (api_editor_quiesce(e) != RPF_OK) return;
        if (e->stream) (void)hipStreamSynchronize(e->stream);
        void* bufs[] = {e->d_cfa, e->d_original, e->d_image, e->d_scratch, e->d_bright, e->d_hsl, e->d_masks, e->d_mask_bits, e->d_mp_wide, e->d_linear_tile,
                        e->d_vign_falloff, e->d_hue_fused_alloc};
        for (void* b : bufs)
            if (b) (void)hipFree(b);
        if (e->stream) (void)hipStreamDestroy(e->stream);
    }
    if (e->proc) processor_unref(e->proc);
    delete e;
}

int rpf_editor_create_headless(uint32_t width, uint32_t height, rpf_editor** out)
{
    if (!out) return fail(RPF_ERR_INVALID_ARGUMENT, "null out");
    *out = nullptr;
    RPF_GUARDED(
        rpf_editor* e = new rpf_editor(width, height);
        e->geom = TileGeom{width, height, 0, height};
        *out = e;
        return ok();)
}

int rpf_editor_create_for_tile(rpf_processor* p, const rpf_tile* tile, rpf_editor** out)
{
    if (!p || !tile || !out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (tile->frame_width == 0 || tile->rows == 0 || (uint64_t)tile->y0 + tile->rows > tile->frame_height)
        return fail(RPF_ERR_INVALID_ARGUMENT, "tile outside its frame");
    // masks and the state machine are tile-sized: rows x frame_width
    RPF_GUARDED(
        std::unique_ptr<rpf_editor> e(new rpf_editor(tile->frame_width, tile->rows));
        e->geom = TileGeom{tile->frame_width, tile->frame_height, tile->y0, tile->rows};
        int rc = editor_device_init(e.get(), p);
        if (rc) { rpf_editor_destroy(e.release()); return rc; }
        *out = e.release();
        return ok();)
}

int rpf_editor_create_from_rgb_f32(rpf_processor* p, const float* rgb, uint32_t width, uint32_t height,
                                   rpf_editor** out)
{
    if (!p || !out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (!rgb || width == 0 || height == 0)  // errors.rs:9-13
        return fail(RPF_ERR_FAILED_TO_READ_IMAGE, "Failed to read image: empty image");
    rpf_tile t{width, height, 0, height};
    rpf_editor* e = nullptr;
    int rc = rpf_editor_create_for_tile(p, &t, &e);
    if (rc) return rc;
    auto cleanup = [&](int code) { rpf_editor_destroy(e); return code; };
    const size_t npix = (size_t)width * height;
    float* d_rgb = nullptr;
    hipError_t he = hipMalloc((void**)&d_rgb, npix * 3 * sizeof(float));
    if (he != hipSuccess) return cleanup(hip_fail(he, "hipMalloc(rgb)"));
    he = hipMalloc((void**)&e->d_original, npix * 4 * sizeof(float));
    if (he == hipSuccess) he = hipMemcpy(d_rgb, rgb, npix * 3 * sizeof(float), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = launch_pad_rgb(d_rgb, npix, e->d_original, e->stream);  // image.rs:135-139
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    (void)hipFree(d_rgb);
    if (he != hipSuccess) return cleanup(hip_fail(he, "upload"));
    e->original_valid = true;
    e->owns_image = true;
    *out = e;
    return ok();
}

int rpf_ppm16_parse_header(const uint8_t* file, size_t len, uint32_t* width, uint32_t* height, size_t* data_offset)
{
    if (!width || !height || !data_offset) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    RPF_GUARDED(const Status st = ppm16_parse_header(file, len, width, height, data_offset);
                return st.code ? fail(st) : ok();)
}

int rpf_editor_create_from_ppm16(rpf_processor* p, const uint8_t* file, size_t len, rpf_editor** out)
{
    if (!p || !out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    uint32_t width = 0, height = 0;
    size_t off = 0;
    if (int rc = rpf_ppm16_parse_header(file, len, &width, &height, &off)) return rc;
    rpf_tile t{width, height, 0, height};
    rpf_editor* e = nullptr;
    int rc = rpf_editor_create_for_tile(p, &t, &e);
    if (rc) return rc;
    const size_t npix = (size_t)width * height;
    uint8_t* d_raw = nullptr;  // the payload lands at an aligned device address whatever the header length was
    hipError_t he = hipMalloc((void**)&d_raw, npix * 6);
    if (he == hipSuccess) he = hipMalloc((void**)&e->d_original, npix * 4 * sizeof(float));
    if (he == hipSuccess) he = hipMemcpy(d_raw, file + off, npix * 6, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = launch_rgb16be(d_raw, npix, e->d_original, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (d_raw) (void)hipFree(d_raw);
    if (he != hipSuccess) { rpf_editor_destroy(e); return hip_fail(he, "upload PPM"); }
    e->original_valid = true;
    e->owns_image = true;
    *out = e;
    return ok();
}

// decoded 8- / 16-bit RGB -> editor (image.rs:412-447 behind the decoder): staged, converted on the device
static int create_from_rgb_int(rpf_processor* p, const void* rgb, int bits, uint32_t width, uint32_t height,
                               const float* lut, rpf_editor** out)
{
    if (!p || !out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (!rgb || width == 0 || height == 0)
        return fail(RPF_ERR_FAILED_TO_READ_IMAGE, "Failed to read image: empty image");
    rpf_tile t{width, height, 0, height};
    rpf_editor* e = nullptr;
    int rc = rpf_editor_create_for_tile(p, &t, &e);
    if (rc) return rc;
    const size_t npix = (size_t)width * height, in_bytes = npix * 3 * (size_t)(bits / 8);
    const size_t lut_bytes = lut ? sizeof(float) * ((size_t)1 << bits) : 0;
    uint8_t* d_raw = nullptr;
    float* d_lut = nullptr;
    hipError_t he = hipMalloc((void**)&d_raw, in_bytes);
    if (he == hipSuccess && lut) he = hipMalloc((void**)&d_lut, lut_bytes);
    if (he == hipSuccess) he = hipMalloc((void**)&e->d_original, npix * 4 * sizeof(float));
    if (he == hipSuccess) he = hipMemcpy(d_raw, rgb, in_bytes, hipMemcpyHostToDevice);
    if (he == hipSuccess && lut) he = hipMemcpy(d_lut, lut, lut_bytes, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = launch_rgb_int(d_raw, bits, npix, d_lut, e->d_original, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (d_raw) (void)hipFree(d_raw);
    if (d_lut) (void)hipFree(d_lut);
    if (he != hipSuccess) { rpf_editor_destroy(e); return hip_fail(he, "upload decoded image"); }
    e->original_valid = true;
    e->owns_image = true;
    *out = e;
    return ok();
}

int rpf_editor_create_from_rgb_u8(rpf_processor* p, const uint8_t* rgb, uint32_t width, uint32_t height,
                                  const float* lut256, rpf_editor** out)
{
    return create_from_rgb_int(p, rgb, 8, width, height, lut256, out);
}

int rpf_editor_create_from_rgb_u16(rpf_processor* p, const uint16_t* rgb, uint32_t width, uint32_t height,
                                   const float* lut65536, rpf_editor** out)
{
    return create_from_rgb_int(p, rgb, 16, width, height, lut65536, out);
}

int rpf_editor_create_from_bayer_u16(rpf_processor* p, const uint16_t* cfa, uint32_t width, uint32_t height,
                                     const rpf_raw_params* raw, rpf_editor** out)
{
    if (!p || !out || !raw) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (!cfa || width < 3 || height < 3)
        return fail(RPF_ERR_FAILED_TO_READ_IMAGE, "Failed to read image: a Bayer frame needs at least 3x3 samples");
    for (int c = 0; c < 4; ++c)
        if (!(raw->white > raw->black[c]))
            return fail(RPF_ERR_FAILED_TO_READ_IMAGE, "Failed to read image: white level must exceed black level");
    rpf_tile t{width, height, 0, height};
    rpf_editor* e = nullptr;
    int rc = rpf_editor_create_for_tile(p, &t, &e);
    if (rc) return rc;
    const size_t row_bytes = (size_t)width * sizeof(uint16_t);
    hipError_t he = hipMalloc((void**)&e->d_cfa, row_bytes * (height + 4));
    if (he == hipSuccess) he = hipMemset(e->d_cfa, 0, row_bytes * (height + 4));
    if (he == hipSuccess)
        he = hipMemcpy(reinterpret_cast<unsigned char*>(e->d_cfa) + 2 * row_bytes, cfa, row_bytes * height,
                       hipMemcpyHostToDevice);
    if (he != hipSuccess) { rpf_editor_destroy(e); return hip_fail(he, "upload CFA"); }
    e->is_bayer = true;
    e->raw = make_raw_params(*raw);
    e->owns_image = true;
    *out = e;
    return ok();
}

int rpf_width(const rpf_editor* e, uint32_t* width)
{
    if (!e || !width) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *width = e->geom.width;
    return ok();
}
int rpf_height(const rpf_editor* e, uint32_t* height)
{
    if (!e || !height) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *height = e->geom.rows;
    return ok();
}

#define STATE_CALL(expr)                                   \
    do {                                                   \
        if (!e) return fail(RPF_ERR_INVALID_ARGUMENT, "null editor"); \
        RPF_GUARDED(Status _st = (expr); return _st ? ok() : fail(_st);) \
    } while (0)

int rpf_set_whitebalance(rpf_editor* e, int32_t temperature, int32_t tint, const char* mask_name)
{
    STATE_CALL(e->state.set_whitebalance(temperature, tint, mask_name));
}
int rpf_set_vignette(rpf_editor* e, int32_t value) { STATE_CALL(e->state.set_vignette(value)); }
int rpf_set_lens_distortion_correction(rpf_editor* e, int32_t value) { STATE_CALL(e->state.set_lens_distortion(value)); }
int rpf_set_tone(rpf_editor* e, float exposure, int32_t contrast, int32_t shadow, int32_t highlight, int32_t black,
                 int32_t white, const char* mask_name)
{
    STATE_CALL(e->state.set_tone(exposure, contrast, shadow, highlight, black, white, mask_name));
}
int rpf_set_brightness_tone_curve(rpf_editor* e, const int32_t* curve, size_t curve_len, const int32_t* xs,
                                  size_t n_x, const int32_t* ys, size_t n_y, const char* mask_name)
{
    STATE_CALL(e->state.set_curve(RPF_CURVE_BRIGHTNESS, curve, curve_len, xs, n_x, ys, n_y, mask_name));
}
int rpf_set_oklch_hue_curve(rpf_editor* e, const int32_t* curve, size_t curve_len, const int32_t* xs, size_t n_x,
                            const int32_t* ys, size_t n_y, const char* mask_name)
{
    STATE_CALL(e->state.set_curve(RPF_CURVE_HUE, curve, curve_len, xs, n_x, ys, n_y, mask_name));
}
int rpf_set_oklch_saturation_curve(rpf_editor* e, const int32_t* curve, size_t curve_len, const int32_t* xs,
                                   size_t n_x, const int32_t* ys, size_t n_y, const char* mask_name)
{
    STATE_CALL(e->state.set_curve(RPF_CURVE_SATURATION, curve, curve_len, xs, n_x, ys, n_y, mask_name));
}
int rpf_set_oklch_lightness_curve(rpf_editor* e, const int32_t* curve, size_t curve_len, const int32_t* xs,
                                  size_t n_x, const int32_t* ys, size_t n_y, const char* mask_name)
{
    STATE_CALL(e->state.set_curve(RPF_CURVE_LIGHTNESS, curve, curve_len, xs, n_x, ys, n_y, mask_name));
}
int rpf_add_mask(rpf_editor* e, const char* name, const float* mask) { STATE_CALL(e->state.add_mask(name, mask)); }
int rpf_remove_mask(rpf_editor* e, const char* name) { STATE_CALL(e->state.remove_mask(name)); }

int rpf_set_mask_range(rpf_editor* e, float mask_range)
{
    if (!e) return fail(RPF_ERR_INVALID_ARGUMENT, "null editor");
    e->state.masks[0].params.mask_range = mask_range;
    return ok();
}

int rpf_set_shader_flavour(rpf_editor* e, int flavour)
{
    if (!e) return fail(RPF_ERR_INVALID_ARGUMENT, "null editor");
    if (flavour != RPF_FLAVOUR_ACTIVE && flavour != RPF_FLAVOUR_GODOT_LEGACY)
        return fail(RPF_ERR_INVALID_ARGUMENT, "unknown shader flavour");
    e->flavour = flavour;
    return ok();
}

int rpf_set_precision(rpf_editor* e, int precision)
{
    if (!e) return fail(RPF_ERR_INVALID_ARGUMENT, "null editor");
    if (precision != RPF_PRECISION_EXACT && precision != RPF_PRECISION_FAST)
        return fail(RPF_ERR_INVALID_ARGUMENT, "unknown precision");
    e->precision = precision;
    return ok();
}

int rpf_get_precision(const rpf_editor* e, int* precision)
{
    if (!e || !precision) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *precision = e->precision;
    return ok();
}

int rpf_reset(rpf_editor* e)
{
    if (!e) return fail(RPF_ERR_INVALID_ARGUMENT, "null editor");
    RPF_GUARDED(e->state.reset();)
    e->image_valid = false;  // lib.rs:228: image = original_image.clone()
    return ok();
}

int rpf_num_masks(const rpf_editor* e, uint32_t* n)
{
    if (!e || !n) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *n = (uint32_t)e->state.masks.size();
    return ok();
}

int rpf_editor_get_gpu_params(const rpf_editor* e, const char* mask_name, rpf_gpu_edit_parameters* out)
{
    if (!e || !out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    RPF_GUARDED(
        Status st;
        const EditParameters* p = e->state.find(mask_name, st);
        if (!p) return fail(st);
        *out = convert_params(*p);
        return ok();)
}

int rpf_editor_get_curve(const rpf_editor* e, int which, const char* mask_name, int32_t* out65536)
{
    if (!e || !out65536 || which < 0 || which > 3) return fail(RPF_ERR_INVALID_ARGUMENT, "bad argument");
    RPF_GUARDED(
        Status st;
        const EditParameters* p = e->state.find(mask_name, st);
        if (!p) return fail(st);
        memcpy(out65536, p->curve[which].data(), sizeof(int32_t) * RPF_CURVE_RESOLUTION);
        return ok();)
}

int rpf_pchip_interpolate_i32(const int32_t* x_pts, size_t n_x, const int32_t* y_pts, size_t n_y,
                              const int32_t* x_eval, size_t n_eval, int32_t* y_eval, int32_t* bad_index)
{
    if ((n_x && !x_pts) || (n_y && !y_pts) || (n_eval && (!x_eval || !y_eval)))
        return fail(RPF_ERR_INVALID_ARGUMENT, "null array");
    RPF_GUARDED(Status st = pchip_interpolate_i32(x_pts, n_x, y_pts, n_y, x_eval, n_eval, y_eval, bad_index);
                return st ? ok() : fail(st);)
}

// ---- preview pyramid (web/main.ts:652-695, 958-1026) -----------------------------------------

int rpf_preview_size(uint32_t src_width, uint32_t src_height, uint32_t target_long_edge, uint32_t* width, uint32_t* height)
{
    if (!width || !height || src_width == 0 || src_height == 0 || target_long_edge == 0)
        return fail(RPF_ERR_INVALID_ARGUMENT, "bad preview size arguments");
    if (src_width >= src_height) {  // Math.round(x) == floor(x + 0.5) for x >= 0
        *width = target_long_edge;
        *height = (uint32_t)floor((double)src_height * ((double)target_long_edge / (double)src_width) + 0.5);
    } else {
        *height = target_long_edge;
        *width = (uint32_t)floor((double)src_width * ((double)target_long_edge / (double)src_height) + 0.5);
    }
    return ok();
}

int rpf_editor_create_preview(rpf_editor* src, uint32_t target_long_edge, rpf_editor** out)
{
    if (!out) return fail(RPF_ERR_INVALID_ARGUMENT, "null out");
    *out = nullptr;
    int rc = need_device(src);
    if (rc) return rc;
    if (!src->owns_image) return fail(RPF_ERR_GPU_COMPUTE, "GPU compute error: tile editor has no image to resize");
    uint32_t dw = 0, dh = 0;
    if ((rc = rpf_preview_size(src->geom.width, src->geom.rows, target_long_edge, &dw, &dh))) return rc;
    if (dw == 0 || dh == 0) return fail(RPF_ERR_INVALID_ARGUMENT, "preview would be empty");
    HIP_TRY(hipSetDevice(src->proc->device));
    if (src->is_bayer && (rc = ensure_linear(src))) return rc;  // the pyramid is built from the linear RGB frame
    rpf_tile t{dw, dh, 0, dh};
    rpf_editor* e = nullptr;
    if ((rc = rpf_editor_create_for_tile(src->proc, &t, &e))) return rc;
    hipError_t he = hipMalloc((void**)&e->d_original, (size_t)dw * dh * 4 * sizeof(float));
    if (he == hipSuccess) he = launch_resize(src->d_original, src->geom.width, src->geom.rows, e->d_original, dw, dh, src->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(src->stream);
    if (he != hipSuccess) { rpf_editor_destroy(e); return hip_fail(he, "preview resize"); }
    e->original_valid = true;
    e->owns_image = true;
    *out = e;
    return ok();
}

// ---- apply / readback -----------------------------------------------------------------------

int rpf_apply_adjustments(rpf_editor* e)
{
    int rc = need_device(e);
    if (rc) return rc;
    if (!e->owns_image) return fail(RPF_ERR_GPU_COMPUTE, "GPU compute error: tile editor has no image; use rpf_dev_*");
    HIP_TRY(hipSetDevice(e->proc->device));
    if ((rc = sync_edit_state(e))) return rc;
    if ((rc = ensure_image_buffer(e))) return rc;
    const LaunchCfg cfg = cfg_of(e->proc);
    // Godot-legacy flavour: its vignette, and — for images that were handed over sRGB-encoded, which is how that
    // crate stores them (rust-godot-legacy/photo-editor/src/image.rs) — srgb_to_linear on the sampled input.
    // A Bayer editor's frame is linear already (the legacy crate let rawler encode it, then decoded it here).
    e->dev.flags = flavour_flags(e);
    if (e->is_bayer && e->dev.lens_strength == 0.0f && e->dev.flags == 0 && e->dev.num_masks <= RPF_MASKS_PER_WORD) {
        HIP_TRY(launch_develop(cfg, e->d_cfa, e->raw, e->geom, e->dev, e->d_image, OUT_RGBA_F32, e->stream));
    } else {
        // resampling needs the demosaiced frame; more than 16 masks run the RGBA pass's general kernel (same bits: the
        // fused kernel is the demosaic-only kernel's arithmetic followed by the pass's)
        if (e->is_bayer && (rc = ensure_linear(e))) return rc;
        HIP_TRY(launch_apply(cfg, e->d_original, e->geom, e->dev, e->d_image, OUT_RGBA_F32, e->stream));
    }
    e->image_valid = true;
    return ok();  // asynchronous like queue.submit (gpu_image_processing.rs:391-399)
}

int rpf_editor_sync(rpf_editor* e)
{
    int rc = need_device(e);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->proc->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ok();
}

static int current_image(rpf_editor* e, const float** d_img)
{
    int rc = need_device(e);
    if (rc) return rc;
    if (!e->owns_image) return fail(RPF_ERR_GPU_COMPUTE, "GPU compute error: tile editor has no image; use rpf_dev_*");
    HIP_TRY(hipSetDevice(e->proc->device));
    if (e->image_valid) {
        *d_img = e->d_image;
    } else {  // PhotoEditor.image starts as the original (lib.rs:113-114)
        if (e->is_bayer && (rc = ensure_linear(e))) return rc;
        *d_img = e->d_original;
    }
    return ok();
}

int rpf_get_rgba_f32(rpf_editor* e, float* out)
{
    if (!out) return fail(RPF_ERR_INVALID_ARGUMENT, "null out");
    const float* img = nullptr;
    int rc = current_image(e, &img);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(out, img, npix_of(e) * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return ok();
}

static int readback_repacked(rpf_editor* e, void* out, int fmt, size_t bytes_per_px)
{
    if (!out) return fail(RPF_ERR_INVALID_ARGUMENT, "null out");
    const float* img = nullptr;
    int rc = current_image(e, &img);
    if (rc) return rc;
    const size_t bytes = npix_of(e) * bytes_per_px;
    if (e->scratch_bytes < bytes) {
        if (e->d_scratch) HIP_TRY(hipFree(e->d_scratch));
        e->d_scratch = nullptr;
        HIP_TRY(hipMalloc(&e->d_scratch, bytes));
        e->scratch_bytes = bytes;
    }
    HIP_TRY(launch_repack(img, npix_of(e), e->d_scratch, fmt, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(out, e->d_scratch, bytes, hipMemcpyDeviceToHost));
    return ok();
}

int rpf_get_rgb_f32(rpf_editor* e, float* out) { return readback_repacked(e, out, OUT_RGB_F32, 3 * sizeof(float)); }
int rpf_get_rgb_u8(rpf_editor* e, uint8_t* out) { return readback_repacked(e, out, OUT_RGB_U8, 3); }

// ---- device-pointer layer -------------------------------------------------------------------

int rpf_dev_demosaic(rpf_processor* p, const uint16_t* d_cfa, const rpf_raw_params* raw, const rpf_tile* tile,
                     float* d_rgba_out, void* stream)
{
    if (!p || !d_cfa || !raw || !tile || !d_rgba_out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    const TileGeom g{tile->frame_width, tile->frame_height, tile->y0, tile->rows};
    HIP_TRY(launch_demosaic(cfg_of(p), d_cfa, make_raw_params(*raw), g, d_rgba_out, (hipStream_t)stream));
    return ok();
}

int rpf_dev_unpack_bayer(rpf_processor* p, const uint8_t* d_packed, size_t row_stride, uint32_t width, uint32_t rows,
                         uint32_t bits, uint32_t order, uint16_t* d_cfa_out, void* stream)
{
    if (!p || !d_packed || !d_cfa_out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    if (order > RPF_PACK_MIPI) return fail(RPF_ERR_INVALID_ARGUMENT, "unknown bit order");
    if (bits != 10 && bits != 12 && bits != 14 && (bits != 16 || order == RPF_PACK_MIPI))
        return fail(RPF_ERR_READ_IMAGE_UNSUPPORTED_FORMAT,
                    order == RPF_PACK_MIPI ? "Unsupported read image format: MIPI-packed samples must be 10, 12 or 14 bits"
                                           : "Unsupported read image format: packed samples must be 10, 12, 14 or 16 bits");
    size_t need = ((size_t)width * bits + 7) / 8;
    if (order == RPF_PACK_MIPI) {  // whole CSI-2 groups: 4 px / 5 B, 2 px / 3 B, 4 px / 7 B
        const size_t n = bits == 12 ? 2 : 4;
        need = ((size_t)width + n - 1) / n * (n + n * (bits - 8) / 8);
    }
    if (row_stride < need) return fail(RPF_ERR_INVALID_ARGUMENT, "row_stride shorter than a packed row");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(launch_unpack(d_packed, row_stride, width, rows, bits, order, d_cfa_out, (hipStream_t)stream));
    return ok();
}

int rpf_dev_prefetch(rpf_processor* p, const void* d_buf, size_t bytes, void* stream)
{
    if (!p || (!d_buf && bytes)) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(launch_touch(d_buf, bytes, p->num_cus, (hipStream_t)stream));
    return ok();
}

int rpf_dev_rgb_u8_to_rgba_f32(rpf_processor* p, const uint8_t* d_rgb, size_t npix, const float* d_lut256, float* d_rgba_out,
                               void* stream)
{
    if (!p || !d_rgb || !d_rgba_out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(launch_rgb_int(d_rgb, 8, npix, d_lut256, d_rgba_out, (hipStream_t)stream));
    return ok();
}

int rpf_dev_rgb_u16_to_rgba_f32(rpf_processor* p, const uint16_t* d_rgb, size_t npix, const float* d_lut65536,
                                float* d_rgba_out, void* stream)
{
    if (!p || !d_rgb || !d_rgba_out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    if (reinterpret_cast<uintptr_t>(d_rgb) & 1) return fail(RPF_ERR_INVALID_ARGUMENT, "samples must be 2-byte aligned");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(launch_rgb_int(d_rgb, 16, npix, d_lut65536, d_rgba_out, (hipStream_t)stream));
    return ok();
}

int rpf_dev_rgb16be_to_rgba_f32(rpf_processor* p, const uint8_t* d_rgb16be, size_t npix, float* d_rgba_out, void* stream)
{
    if (!p || !d_rgb16be || !d_rgba_out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    if (reinterpret_cast<uintptr_t>(d_rgb16be) & 1) return fail(RPF_ERR_INVALID_ARGUMENT, "samples must be 2-byte aligned");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(launch_rgb16be(d_rgb16be, npix, d_rgba_out, (hipStream_t)stream));
    return ok();
}

int rpf_dev_develop(rpf_editor* e, const uint16_t* d_cfa, const rpf_raw_params* raw, void* d_out, int out_format,
                    void* stream)
{
    int rc = need_device(e);
    if (rc) return rc;
    if (!d_cfa || !raw || !d_out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(e->proc->device));
    if ((rc = sync_edit_state(e))) return rc;
    if (e->dev.lens_strength != 0.0f)
        return fail(RPF_ERR_INVALID_ARGUMENT,
                    "lens distortion resamples the whole frame: demosaic first, then rpf_dev_apply on the full frame");
    if (e->flavour != RPF_FLAVOUR_ACTIVE)
        return fail(RPF_ERR_INVALID_ARGUMENT, "the fused kernel implements the active shader only: use rpf_dev_demosaic, "
                                              "then rpf_dev_apply, for the Godot-legacy flavour");
    if ((hipStream_t)stream != e->stream) e->foreign_streams = true;
    e->dev.flags = 0;
    if (e->dev.num_masks > RPF_MASKS_PER_WORD) {
        // More masks than the fused kernel takes: the demosaic-only kernel into the editor's own linear tile, then the RGBA
        // pass — the same bits (rpf_apply_adjustments and a sharded step do the same).  That tile is the editor's: with so many
        // masks it serves one stream at a time.
        if (!e->d_linear_tile) HIP_TRY(hipMalloc((void**)&e->d_linear_tile, npix_of(e) * 4 * sizeof(float)));
        HIP_TRY(launch_demosaic(cfg_of(e->proc), d_cfa, make_raw_params(*raw), e->geom, e->d_linear_tile, (hipStream_t)stream));
        HIP_TRY(launch_apply(cfg_of(e->proc), e->d_linear_tile, e->geom, e->dev, d_out, out_format, (hipStream_t)stream));
        return ok();
    }
    HIP_TRY(launch_develop(cfg_of(e->proc), d_cfa, make_raw_params(*raw), e->geom, e->dev, d_out, out_format,
                           (hipStream_t)stream));
    return ok();
}

int rpf_dev_develop_packed(rpf_editor* e, const uint8_t* d_packed, size_t row_stride, uint32_t bits, uint32_t order,
                           const rpf_raw_params* raw, void* d_out, int out_format, void* stream)
{
    int rc = need_device(e);
    if (rc) return rc;
    if (order > RPF_PACK_MIPI) return fail(RPF_ERR_INVALID_ARGUMENT, "unknown bit order");
    if (!d_packed || !raw || !d_out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    if (bits != 10 && bits != 12 && bits != 14 && (bits != 16 || order == RPF_PACK_MIPI))
        return fail(RPF_ERR_READ_IMAGE_UNSUPPORTED_FORMAT,
                    order == RPF_PACK_MIPI ? "Unsupported read image format: MIPI-packed samples must be 10, 12 or 14 bits"
                                           : "Unsupported read image format: packed samples must be 10, 12, 14 or 16 bits");
    size_t need = ((size_t)e->geom.width * bits + 7) / 8;
    if (order == RPF_PACK_MIPI) {  // whole CSI-2 groups; the kernel reads a group through an 8-byte window
        const size_t n = bits == 12 ? 2 : 4;
        need = ((size_t)e->geom.width + n - 1) / n * (n + n * (bits - 8) / 8);
        if (need < 8) need = 8;
    }
    if (row_stride < need || row_stride > 0xffffffffu)
        return fail(RPF_ERR_INVALID_ARGUMENT, "row_stride shorter than a packed row");
    HIP_TRY(hipSetDevice(e->proc->device));
    if ((rc = sync_edit_state(e))) return rc;
    if (e->dev.lens_strength != 0.0f)
        return fail(RPF_ERR_INVALID_ARGUMENT,
                    "lens distortion resamples the whole frame: unpack + demosaic first, then rpf_dev_apply on the full frame");
    if (e->flavour != RPF_FLAVOUR_ACTIVE)
        return fail(RPF_ERR_INVALID_ARGUMENT, "the fused kernel implements the active shader only");
    if (e->dev.num_masks > RPF_MASKS_PER_WORD)
        return fail(RPF_ERR_INVALID_ARGUMENT, "packed rows with more than \"main\" + 15 user masks: rpf_dev_unpack_bayer first, then "
                                              "rpf_dev_develop on the u16 band (the same bits)");
    if ((hipStream_t)stream != e->stream) e->foreign_streams = true;
    e->dev.flags = 0;
    HIP_TRY(launch_develop_packed(cfg_of(e->proc), d_packed, (uint32_t)row_stride, bits, order, make_raw_params(*raw), e->geom,
                                  e->dev, d_out, out_format, (hipStream_t)stream));
    return ok();
}

int rpf_dev_apply(rpf_editor* e, const float* d_rgba_in, void* d_out, int out_format, void* stream)
{
    int rc = need_device(e);
    if (rc) return rc;
    if (!d_rgba_in || !d_out) return fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(e->proc->device));
    if ((rc = sync_edit_state(e))) return rc;
    if (e->dev.lens_strength != 0.0f && (e->geom.y0 != 0 || e->geom.rows != e->geom.height))
        return fail(RPF_ERR_INVALID_ARGUMENT, "lens distortion resamples the whole frame: the tile must be the frame");
    if ((hipStream_t)stream != e->stream) e->foreign_streams = true;
    e->dev.flags = flavour_flags(e);  // legacy: d_rgba_in is taken as sRGB-encoded, as that shader's input is
    HIP_TRY(launch_apply(cfg_of(e->proc), d_rgba_in, e->geom, e->dev, d_out, out_format, (hipStream_t)stream));
    return ok();
}

}  // extern "C"
