// rpf_internal.h — what the translation units behind include/rpf.h share: the two handle types and the error plumbing.
// Not installed, not part of the ABI.
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "rpf_host.h"
#include "rpf_launch.h"

struct rpf_processor {
    int device = 0;
    int num_cus = 0;
    int variant = 0;
    rpf::MathTables* d_tabs = nullptr;
    rpf::LegacyTables* d_legacy = nullptr;  // Godot-legacy flavour only
    rpf::FastTables* d_fast_tabs = nullptr; // RPF_PRECISION_FAST (rpf_fast.h), 17.5 KiB
    rpf::HueBound* d_hue_bounds = nullptr;  // 65536 entries, 1 MiB
    rpf::HueBoundCs* d_hue_cs = nullptr;    // 65536 entries, 2 MiB (HUE_DIRECT launches)
    // Arc<GpuProcessor> (lib.rs:83): one reference for the caller's handle plus one per editor created on it.
    // rpf_processor_destroy drops the caller's; the device tables are freed by whoever drops the last one.
    std::atomic<int> refs{1};
};

struct CurveStamps {
    uint64_t v[4];
};

struct rpf_editor {
    rpf_processor* proc = nullptr;  // null: headless (state only)
    rpf::EditState state;
    rpf::TileGeom geom{};
    bool owns_image = false;      // created from host pixels (vs. for_tile)
    bool is_bayer = false;
    int flavour = RPF_FLAVOUR_ACTIVE;
    int precision = RPF_PRECISION_EXACT;
    rpf::RawParamsDev raw{};
    uint16_t* d_cfa = nullptr;    // (rows + 4) x width, rows 0,1 and rows+2,rows+3 are halo
    float* d_original = nullptr;  // RGBA32F linear (rgb editors; bayer editors: demosaic cache)
    bool original_valid = false;
    float* d_image = nullptr;     // RGBA32F result of the last apply
    bool image_valid = false;
    void* d_scratch = nullptr;    // readback repack
    size_t scratch_bytes = 0;
    // device copy of the edit state
    uint16_t* d_bright = nullptr;
    ushort4* d_hsl = nullptr;
    uint8_t* d_masks = nullptr;       // user-mask planes as uploaded (u8, binarised)
    uint16_t* d_mask_bits = nullptr;  // what the kernels read: one u16 of mask bits per pixel, combined on the device
                                      // (more than 16 masks: one such plane per 16 masks, rpf_launch.h DeviceEdit::mask_bits)
    uint32_t mask_bit_planes = 0;     // planes d_mask_bits holds room for
    // rpf_dev_develop / a sharded step with more than 16 masks: the band's linear RGBA32F between the demosaic-only kernel and
    // the RGBA pass (rows x width; a buffer of its own — d_original is an RGB editor's IMAGE, not a scratch)
    float* d_linear_tile = nullptr;
    rpf::MaskParams* d_mp_wide = nullptr;  // every mask's parameters, for editors with more than 16 masks (DeviceEdit::mp_wide)
    uint32_t mp_wide_capacity = 0;
    // one mask with non-trivial hue / saturation / lightness curves: the fused hue table and what it was built from
    void* d_hue_fused_alloc = nullptr;
    uint64_t hue_fused_stamp[3] = {~0ull, ~0ull, ~0ull};  // curve versions (hue, saturation, lightness) of "main"
    uint32_t hue_fused_flags = 0;
    float hue_fused_gains[2] = {0.0f, 0.0f};
    bool hue_fused_valid = false;
    float* d_vign_falloff = nullptr;  // rows x width: the vignette's falloff per pixel, built when the slider first leaves 0
    uint32_t lut_capacity = 0;    // masks worth of LUT storage
    uint32_t mask_capacity = 0;   // user-mask planes allocated
    uint64_t uploaded_params = 0, uploaded_masks = 0;
    std::vector<CurveStamps> uploaded_curves;  // per mask slot: EditParameters::curve_ver of what the device holds
    rpf::DeviceEdit dev{};
    hipStream_t stream = nullptr;
    // Steps handed to a shard's thread (rpf_shard_develop_async) that the thread has not enqueued yet.  Each holds a COPY of
    // `dev` — raw pointers to d_bright / d_hsl / d_mask_bits / the fused hue table / the falloff map — so nothing may free or
    // overwrite those tables, and the editor may not go away, before the count is back at zero (api_editor_quiesce).
    std::mutex qmu;
    std::condition_variable qcv;
    int queued = 0;
    // a launch that reads this editor's tables went to a stream other than `stream` (rpf_dev_*, rpf_shard_*): quiescing
    // then means the whole device; an editor driven through the operator layer alone waits for its own stream only
    bool foreign_streams = false;

    rpf_editor(uint32_t w, uint32_t h) : state(w, h) {}
};


namespace rpf {
// thread-local rpf_last_error() text + the status code, for every translation unit of the library
int api_fail(int code, const std::string& msg);
int api_hip_fail(hipError_t e, const char* what);
// refresh the device copy of params / curves / masks if a setter ran since the last upload (rpf_api.cpp)
int api_sync_edit_state(rpf_editor* e);
// Nothing reads the editor's device tables any more: steps still queued in a shard's thread have been enqueued (bounded
// wait, api_wait_timeout_ms; an error names what did not happen) and every launch that could read them has completed.
int api_editor_quiesce(rpf_editor* e);
// RPF_WAIT_TIMEOUT_MS (default 30000; 0 = no bound): the longest any call of this library waits for another thread or rank
int api_wait_timeout_ms();
// Arc<GpuProcessor>: drop one reference (the device tables go with the last one)
void api_processor_unref(rpf_processor* p);
inline LaunchCfg cfg_of(const rpf_processor* p) { return LaunchCfg{p->num_cus, p->variant}; }
}  // namespace rpf
