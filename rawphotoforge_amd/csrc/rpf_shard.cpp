// rpf_shard.cpp — one rank's end of the row-tile sharding of ONE frame over the GPUs of a node (include/rpf.h,
// "row-tile sharding").  Replaces the single dispatch of gpu_image_processing.rs:385-391 where the
// frame is cut into row bands, one process and GPU per band.
//
// MI355X-first: xGMI is point-to-point and a halo is 2 rows (24-48 KB), so the exchange is latency, not bandwidth.  It is
// one ncclGroup of neighbour sends / receives on the step's stream in front of ONE launch over the band; everything a step
// needs is enqueued by one C call: no Python, no per-step allocation, no host synchronisation (rpf_shard_develop).  Every
// cross-stream dependency costs this runtime 4-13 us per step — round 3 measured the two schedules that hide the exchange
// behind such dependencies (interior / edge rows on a side stream; the exchange one frame ahead) as losers at both band sizes
// and round 4 removed them — so the latency-hiding schedule is the one that has none:
// rpf_shard_develop_async hands the whole step (exchange, then launch, on the step's stream) to the shard's thread, and a
// rank runs TWO shards — two communicators, two streams, two band buffers — with frames alternating between them: one
// frame's exchange, launch latency and tail beside the other frame's kernel (a 6000 x 500 band in 26 us per step, what the
// kernel alone takes; 39 with the exchange in line).  Every wait is bounded (rpf_shard_flush / rpf_shard_drain).
//
// librccl is loaded with dlopen at rpf_shard_create (world > 1, or a self-exchange rehearsal): librpf_hip.so does not
// link it, so a single-GPU host (the reference's only configuration) never maps the 300-MB library.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enumerators only: every entry point is resolved with dlsym
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "rpf_internal.h"

using namespace rpf;

namespace {

struct Rccl {
    void* handle = nullptr;
    std::string path, error;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// One RCCL per process.  A library that is already mapped under the SONAME librccl.so.1 (a ROCm PyTorch wheel brings
// its own) is what dlopen returns for that name, so the process never holds two.
const Rccl* rccl()
{
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("RPF_RCCL_LIB");
        const char* cands[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* c : cands) {
            if (!c || !*c) continue;
            R.handle = dlopen(c, RTLD_NOW | RTLD_LOCAL);
            if (R.handle) { R.path = c; break; }
            const char* de = dlerror();
            R.error += std::string(R.error.empty() ? "" : "; ") + c + ": " + (de ? de : "?");
        }
        if (!R.handle) return;
        bool all = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(R.handle, name);
            if (!p) { all = false; R.error += std::string(" missing symbol ") + name; }
            return p;
        };
        R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
        R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
        R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
        R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
        R.Send = reinterpret_cast<decltype(R.Send)>(sym("ncclSend"));
        R.Recv = reinterpret_cast<decltype(R.Recv)>(sym("ncclRecv"));
        R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
        if (!all) { dlclose(R.handle); R.handle = nullptr; }
    });
    return R.handle ? &R : nullptr;
}

int rccl_missing()
{
    return api_fail(RPF_ERR_GPU_INITIALIZATION, "GPU initialization error: librccl could not be loaded for the halo exchange");
}

int nccl_fail(const Rccl* R, ncclResult_t r, const char* what, int code = RPF_ERR_GPU_COMPUTE)
{
    return api_fail(code, std::string(code == RPF_ERR_GPU_INITIALIZATION ? "GPU initialization error: " : "GPU compute error: ") +
                              what + ": " + (R && R->GetErrorString ? R->GetErrorString(r) : "RCCL error"));
}

#define HIP_TRY(expr)                                         \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return api_hip_fail(_e, #expr); \
    } while (0)
#define NCCL_TRY(R, expr)                                        \
    do {                                                         \
        ncclResult_t _r = (expr);                                \
        if (_r != ncclSuccess) return nccl_fail(R, _r, #expr);   \
    } while (0)

}  // namespace

namespace {
// Everything a band's launch needs, by value: a step handed to the shard's thread outlives the call that described it.
struct BandLaunch {
    rpf_processor* proc;
    TileGeom geom;
    DeviceEdit dev;
    const unsigned char* band;
    uint32_t row_stride, bits, order;
    RawParamsDev raw;
    void* out;
    int out_format;
    // More masks than the fused band kernel takes (rpf_launch.h DeviceEdit::mp_wide): the band's linear RGBA32F, an
    // intermediate of the EDITOR's (rows x width), between the demosaic-only kernel and the RGBA pass.  Null otherwise.
    float* linear;
};
}  // namespace

struct rpf_shard {
    rpf_processor* proc = nullptr;
    int rank = 0, world = 1;
    int up = -1, down = -1;  // ranks that hold the rows above / below this band; -1 = frame edge
    ncclComm_t comm = nullptr;
    char bus[32] = {0};

    // Whole steps handed to the shard's thread (rpf_shard_develop_async): the exchange and the launch, enqueued on the
    // step's stream by that thread.  Two shards (two communicators, two streams, two threads) alternating frames overlap
    // one frame's exchange, launch latency and tail with the other frame's kernel without one cross-stream dependency.
    struct Step {
        BandLaunch b;
        size_t row_bytes;
        uint32_t rows;
        bool exchange, timed;
        hipStream_t stream;
        rpf_editor* ed;  // its `queued` count is dropped when the step has been enqueued (rpf_internal.h)
    };
    struct Work { Step* step; };
    int in_flight = 0;           // steps handed over and not yet enqueued
    // progress of the shard's thread, for the message of a wait that ran into its deadline (rpf_shard_flush)
    uint64_t steps_handed = 0, steps_enqueued = 0;
    std::atomic<const char*> phase{"idle"};
    hipStream_t last_stream = nullptr;  // of the most recent step handed over (rpf_shard_drain)
    int timeout_ms = -1;                // rpf_shard_set_timeout; -1: api_wait_timeout_ms()
    int async_rc = 0;            // the first failure of such a step, for rpf_shard_flush
    std::string async_error;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timing;  // start / stop of the timed launches since the last rpf_shard_timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timing_free;
    // `rccl_mu` serialises every use of the communicator (the shard's thread's and the calling thread's: rpf_shard_exchange,
    // rpf_shard_develop); `mu` guards the queue and the counters
    std::mutex rccl_mu, mu;
    std::condition_variable cv;
    std::deque<Work> work;
    std::thread worker;
    bool stop = false;
};

namespace {

int need_comm(rpf_shard* s)
{
    if (s->comm) return RPF_OK;
    // a self-exchange on a shard created without an id (world = 1): a one-rank communicator of its own
    const Rccl* R = rccl();
    if (!R) return rccl_missing();
    if (s->world != 1) return api_fail(RPF_ERR_INVALID_ARGUMENT, "shard has no communicator");
    ncclUniqueId id;
    ncclResult_t r = R->GetUniqueId(&id);
    if (r == ncclSuccess) r = R->CommInitRank(&s->comm, 1, id, 0);
    if (r != ncclSuccess) return nccl_fail(R, r, "ncclCommInitRank(1 rank)", RPF_ERR_GPU_INITIALIZATION);
    return RPF_OK;
}

// The 2-row halos of `band` with the ranks above and below, as one group on `stream`.
int enqueue_exchange(rpf_shard* s, unsigned char* band, size_t row_bytes, uint32_t rows, hipStream_t stream)
{
    if (s->up < 0 && s->down < 0) return RPF_OK;
    if (rows < 2) return api_fail(RPF_ERR_INVALID_ARGUMENT, "a band must have at least 2 rows to feed its neighbours' halos");
    std::lock_guard<std::mutex> lock(s->rccl_mu);
    int rc = need_comm(s);
    if (rc) return rc;
    const Rccl* R = rccl();
    const size_t n = 2 * row_bytes;  // RCCL has no 16-bit integer type: the rows travel as bytes
    NCCL_TRY(R, R->GroupStart());
    ncclResult_t r = ncclSuccess;
    if (s->up >= 0) {  // my first two rows are the bottom halo of the rank above
        r = R->Send(band + 2 * row_bytes, n, ncclUint8, s->up, s->comm, stream);
        if (r == ncclSuccess) r = R->Recv(band, n, ncclUint8, s->up, s->comm, stream);
    }
    if (r == ncclSuccess && s->down >= 0) {
        r = R->Send(band + (size_t)rows * row_bytes, n, ncclUint8, s->down, s->comm, stream);
        if (r == ncclSuccess) r = R->Recv(band + ((size_t)rows + 2) * row_bytes, n, ncclUint8, s->down, s->comm, stream);
    }
    const ncclResult_t rg = R->GroupEnd();  // always closed, also after a failed call inside it
    if (r != ncclSuccess) return nccl_fail(R, r, "ncclSend/ncclRecv(halo rows)");
    if (rg != ncclSuccess) return nccl_fail(R, rg, "ncclGroupEnd(halo rows)");
    return RPF_OK;
}

hipError_t launch_band(const BandLaunch& b, hipStream_t stream, hipEvent_t stop_event = nullptr,
                       hipEvent_t start_event = nullptr, int reserve_cus = 0)
{
    LaunchCfg cfg = cfg_of(b.proc);
    if (reserve_cus > 0 && cfg.num_cus > 4 * reserve_cus) cfg.num_cus -= reserve_cus;
    cfg.stop_event = stop_event;
    cfg.start_event = start_event;
    if (b.linear) {  // two launches, the same bits (rpf_api.cpp rpf_apply_adjustments does the same for a whole frame)
        cfg.stop_event = cfg.start_event = nullptr;
        hipError_t e = start_event ? hipEventRecord(start_event, stream) : hipSuccess;
        if (e == hipSuccess) e = launch_demosaic(cfg, reinterpret_cast<const uint16_t*>(b.band), b.raw, b.geom, b.linear, stream);
        if (e == hipSuccess) e = launch_apply(cfg, b.linear, b.geom, b.dev, b.out, b.out_format, stream);
        if (e == hipSuccess && stop_event) e = hipEventRecord(stop_event, stream);
        return e;
    }
    if (b.bits)
        return launch_develop_packed(cfg, b.band, b.row_stride, b.bits, b.order, b.raw, b.geom, b.dev, b.out, b.out_format, stream);
    return launch_develop(cfg, reinterpret_cast<const uint16_t*>(b.band), b.raw, b.geom, b.dev, b.out, b.out_format, stream);
}

// The shard's thread: enqueues the steps handed over by rpf_shard_develop_async, in order.  Errors travel to rpf_shard_flush
// (rpf_last_error is per thread).
int run_step(rpf_shard* s, const rpf_shard::Step& t);

void exchange_worker(rpf_shard* s)
{
    (void)hipSetDevice(s->proc->device);
    for (;;) {
        rpf_shard::Work w{nullptr};
        {
            std::unique_lock<std::mutex> lock(s->mu);
            s->cv.wait(lock, [&] { return s->stop || !s->work.empty(); });
            if (s->work.empty()) return;  // stop, and nothing left to enqueue
            w = s->work.front();
            s->work.pop_front();
        }
        {   // a whole step: exchange + launch on its stream
            const int rc = run_step(s, *w.step);
            s->phase.store("idle");
            {   // the step's copy of the editor's table pointers is no longer needed: setters / destroy may proceed
                rpf_editor* ed = w.step->ed;
                std::lock_guard<std::mutex> qlock(ed->qmu);
                --ed->queued;
                ed->qcv.notify_all();
            }
            delete w.step;
            std::lock_guard<std::mutex> lock(s->mu);
            if (rc && !s->async_rc) { s->async_rc = rc; s->async_error = rpf_last_error(); }
            --s->in_flight;
            ++s->steps_enqueued;
        }
        s->cv.notify_all();
    }
}

// One step on its stream, by the shard's thread: the halo exchange, then the band's launch (timed: with a pair of events
// as the launch's own start / completion signals — no marker packets around it).
int run_step(rpf_shard* s, const rpf_shard::Step& t)
{
    int rc = RPF_OK;
    s->phase.store("halo exchange (ncclGroupStart ... ncclGroupEnd)");
    if (t.exchange && (rc = enqueue_exchange(s, const_cast<unsigned char*>(t.b.band), t.row_bytes, t.rows, t.stream))) return rc;
    s->phase.store("launch");
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    bool timed = t.timed;
    if (timed) {
        std::lock_guard<std::mutex> lock(s->mu);
        if (s->timing.size() >= 4096) timed = false;  // nobody has collected them (rpf_shard_timing): stop adding events
        else if (!s->timing_free.empty()) { ev = s->timing_free.back(); s->timing_free.pop_back(); }
    }
    if (timed && !ev.first) {
        hipError_t he = hipEventCreate(&ev.first);
        if (he == hipSuccess && (he = hipEventCreate(&ev.second)) != hipSuccess) { (void)hipEventDestroy(ev.first); ev.first = nullptr; }
        if (he != hipSuccess) return api_hip_fail(he, "hipEventCreate(timed launch)");
    }
    // A step that exchanges is one of two pipelines: its launch leaves a CU per XCD to the OTHER pipeline's exchange, which
    // starts while this kernel runs — RCCL's workgroup needs a CU of its own (the fused kernel's persistent workgroups
    // fill theirs), and a persistent workgroup that has to wait for that CU makes its whole launch late.
    static const int reserve = getenv("RPF_SHARD_ASYNC_RESERVE") ? atoi(getenv("RPF_SHARD_ASYNC_RESERVE")) : 8;
    const hipError_t le = launch_band(t.b, t.stream, ev.second, ev.first, t.exchange ? reserve : 0);
    if (timed) {  // the pair goes back to the pool whether or not the launch took it
        std::lock_guard<std::mutex> lock(s->mu);
        (le == hipSuccess ? s->timing : s->timing_free).push_back(ev);
    }
    if (le != hipSuccess) return api_hip_fail(le, "fused launch of an async step");
    return RPF_OK;
}

int shard_deadline_ms(const rpf_shard* s) { return s->timeout_ms >= 0 ? s->timeout_ms : api_wait_timeout_ms(); }

std::string shard_who(const rpf_shard* s)
{
    return "rank " + std::to_string(s->rank) + " of " + std::to_string(s->world) + " (neighbour above " + std::to_string(s->up) +
           ", below " + std::to_string(s->down) + ", shard " + std::to_string((unsigned long long)(uintptr_t)s % 100000) + ")";
}

}  // namespace

extern "C" {

int rpf_shard_unique_id(uint8_t id[RPF_SHARD_ID_BYTES])
{
    static_assert(RPF_SHARD_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is an ncclUniqueId");
    if (!id) return api_fail(RPF_ERR_INVALID_ARGUMENT, "null id");
    const Rccl* R = rccl();
    if (!R) return rccl_missing();
    ncclUniqueId u;
    const ncclResult_t r = R->GetUniqueId(&u);
    if (r != ncclSuccess) return nccl_fail(R, r, "ncclGetUniqueId", RPF_ERR_GPU_INITIALIZATION);
    memcpy(id, u.internal, RPF_SHARD_ID_BYTES);
    return RPF_OK;
}

void rpf_shard_destroy(rpf_shard* s)
{
    if (!s) return;
    if (s->worker.joinable()) {
        bool idle = true;
        {
            std::unique_lock<std::mutex> lock(s->mu);
            const int ms = shard_deadline_ms(s);
            auto done = [&] { return s->in_flight == 0; };
            if (ms > 0) idle = s->cv.wait_for(lock, std::chrono::milliseconds(ms), done);
            else s->cv.wait(lock, done);
            s->stop = true;
        }
        s->cv.notify_all();
        if (!idle) {  // the thread sits in a call that does not return (a stalled exchange): leave it and the shard behind
            s->worker.detach();
            return;
        }
        s->worker.join();
    }
    if (s->proc && hipSetDevice(s->proc->device) == hipSuccess) {
        const Rccl* R = s->comm ? rccl() : nullptr;
        if (R) (void)R->CommDestroy(s->comm);
        for (auto* v : {&s->timing, &s->timing_free})
            for (auto& pr : *v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    }
    if (s->proc) api_processor_unref(s->proc);
    delete s;
}

int rpf_shard_create(rpf_processor* p, const uint8_t* id, int rank, int world, rpf_shard** out)
{
    if (!p || !out) return api_fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return api_fail(RPF_ERR_INVALID_ARGUMENT, "rank outside [0, world)");
    if (world > 1 && !id) return api_fail(RPF_ERR_INVALID_ARGUMENT, "a shard of more than one rank needs the unique id");
    rpf_shard* s = new (std::nothrow) rpf_shard;
    if (!s) return api_fail(RPF_ERR_GPU_INITIALIZATION, "out of host memory");
    s->proc = p;
    p->refs.fetch_add(1);
    s->rank = rank;
    s->world = world;
    s->up = rank > 0 ? rank - 1 : -1;
    s->down = rank < world - 1 ? rank + 1 : -1;
    auto bail = [&](int rc) { rpf_shard_destroy(s); return rc; };
    hipError_t he = hipSetDevice(p->device);
    if (he == hipSuccess) he = hipDeviceGetPCIBusId(s->bus, (int)sizeof(s->bus), p->device);
    if (he != hipSuccess) return bail(api_hip_fail(he, "shard device"));
    if (world > 1) {
        const Rccl* R = rccl();
        if (!R) return bail(rccl_missing());
        ncclUniqueId u;
        memcpy(u.internal, id, RPF_SHARD_ID_BYTES);
        const ncclResult_t r = R->CommInitRank(&s->comm, world, u, rank);
        if (r != ncclSuccess) return bail(nccl_fail(R, r, "ncclCommInitRank", RPF_ERR_GPU_INITIALIZATION));
    }
    *out = s;
    return RPF_OK;
}

int rpf_shard_set_neighbours(rpf_shard* s, int up, int down)
{
    if (!s) return api_fail(RPF_ERR_INVALID_ARGUMENT, "null shard");
    if (up < -1 || up >= s->world || down < -1 || down >= s->world)
        return api_fail(RPF_ERR_INVALID_ARGUMENT, "neighbour outside [-1, world)");
    s->up = up;
    s->down = down;
    return RPF_OK;
}

int rpf_shard_info(const rpf_shard* s, int* rank, int* world, int* up, int* down, char* pci_bus_id, size_t len)
{
    if (!s) return api_fail(RPF_ERR_INVALID_ARGUMENT, "null shard");
    if (rank) *rank = s->rank;
    if (world) *world = s->world;
    if (up) *up = s->up;
    if (down) *down = s->down;
    if (pci_bus_id && len) {
        strncpy(pci_bus_id, s->bus, len - 1);
        pci_bus_id[len - 1] = 0;
    }
    return RPF_OK;
}

int rpf_shard_exchange(rpf_shard* s, void* d_band, size_t row_bytes, uint32_t rows, void* stream)
{
    if (!s || !d_band || row_bytes == 0) return api_fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(s->proc->device));
    return enqueue_exchange(s, static_cast<unsigned char*>(d_band), row_bytes, rows, (hipStream_t)stream);
}

namespace {
// Argument checks and edit-state upload shared by rpf_shard_develop and rpf_shard_develop_async; fills the launch.
int prepare_band_launch(rpf_shard* s, rpf_editor* e, void* d_band, size_t row_stride, uint32_t bits, uint32_t order,
                        const rpf_raw_params* raw, void* d_out, int out_format, BandLaunch& b, size_t& row_bytes)
{
    if (!s || !e || !d_band || !raw || !d_out) return api_fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    if (!e->proc) return api_fail(RPF_ERR_GPU_COMPUTE, "GPU compute error: headless editor has no device (no CPU fallback exists)");
    if (e->proc != s->proc) return api_fail(RPF_ERR_INVALID_ARGUMENT, "editor and shard belong to different processors");
    if (order > RPF_PACK_MIPI) return api_fail(RPF_ERR_INVALID_ARGUMENT, "unknown bit order");
    if (bits != 0 && bits != 10 && bits != 12 && bits != 14 && (bits != 16 || order == RPF_PACK_MIPI))
        return api_fail(RPF_ERR_READ_IMAGE_UNSUPPORTED_FORMAT, "Unsupported read image format: packed samples must be 10, 12, 14 or 16 bits");
    row_bytes = bits ? row_stride : (size_t)e->geom.width * sizeof(uint16_t);
    if (bits && (row_stride < ((size_t)e->geom.width * bits + 7) / 8 || row_stride > 0xffffffffu))
        return api_fail(RPF_ERR_INVALID_ARGUMENT, "row_stride shorter than a packed row");
    HIP_TRY(hipSetDevice(s->proc->device));
    int rc = api_sync_edit_state(e);
    if (rc) return rc;
    if (e->dev.lens_strength != 0.0f)
        return api_fail(RPF_ERR_INVALID_ARGUMENT, "lens distortion resamples the whole frame: it does not shard by rows");
    if (e->flavour != RPF_FLAVOUR_ACTIVE)
        return api_fail(RPF_ERR_INVALID_ARGUMENT, "the fused kernel implements the active shader only");
    float* linear = nullptr;
    if (e->dev.num_masks > RPF_MASKS_PER_WORD) {  // the band's steps become demosaic + RGBA pass (launch_band)
        if (bits)
            return api_fail(RPF_ERR_INVALID_ARGUMENT, "a packed band with more than \"main\" + 15 user masks: unpack it first "
                                                      "(rpf_dev_unpack_bayer), then the u16 band");
        if (!e->d_linear_tile)
            HIP_TRY(hipMalloc((void**)&e->d_linear_tile, (size_t)e->geom.rows * e->geom.width * 4 * sizeof(float)));
        linear = e->d_linear_tile;
    }
    e->dev.flags = 0;
    e->foreign_streams = true;  // the launch goes to a stream of the caller's (api_editor_quiesce)
    b = BandLaunch{e->proc, e->geom, e->dev, static_cast<const unsigned char*>(d_band), (uint32_t)row_stride, bits, order,
                   make_raw_params(*raw), d_out, out_format, linear};
    return RPF_OK;
}
}  // namespace

int rpf_shard_develop(rpf_shard* s, rpf_editor* e, void* d_band, size_t row_stride, uint32_t bits, uint32_t order,
                      const rpf_raw_params* raw, void* d_out, int out_format, void* stream, uint32_t flags)
{
    BandLaunch b;
    size_t row_bytes = 0;
    int rc = prepare_band_launch(s, e, d_band, row_stride, bits, order, raw, d_out, out_format, b, row_bytes);
    if (rc) return rc;
    if (flags & ~(uint32_t)RPF_SHARD_NO_EXCHANGE) return api_fail(RPF_ERR_INVALID_ARGUMENT, "rpf_shard_develop takes RPF_SHARD_NO_EXCHANGE only");
    hipStream_t cur = (hipStream_t)stream;
    unsigned char* band = static_cast<unsigned char*>(d_band);
    const bool exchange = !(flags & RPF_SHARD_NO_EXCHANGE) && (s->up >= 0 || s->down >= 0);
    if (exchange && (rc = enqueue_exchange(s, band, row_bytes, b.geom.rows, cur))) return rc;
    HIP_TRY(launch_band(b, cur));
    return RPF_OK;
}

int rpf_shard_develop_async(rpf_shard* s, rpf_editor* e, void* d_band, size_t row_stride, uint32_t bits, uint32_t order,
                            const rpf_raw_params* raw, void* d_out, int out_format, void* stream, uint32_t flags)
{
    if (flags & ~(uint32_t)(RPF_SHARD_NO_EXCHANGE | RPF_SHARD_TIMED))
        return api_fail(RPF_ERR_INVALID_ARGUMENT, "rpf_shard_develop_async takes RPF_SHARD_NO_EXCHANGE and RPF_SHARD_TIMED only");
    BandLaunch b;
    size_t row_bytes = 0;
    const int rc = prepare_band_launch(s, e, d_band, row_stride, bits, order, raw, d_out, out_format, b, row_bytes);
    if (rc) return rc;
    const bool exchange = !(flags & RPF_SHARD_NO_EXCHANGE) && (s->up >= 0 || s->down >= 0);
    rpf_shard::Step* t = new (std::nothrow) rpf_shard::Step{b, row_bytes, b.geom.rows, exchange, (flags & RPF_SHARD_TIMED) != 0,
                                                           (hipStream_t)stream, e};
    if (!t) return api_fail(RPF_ERR_GPU_COMPUTE, "GPU compute error: out of host memory");
    {   // from here until the shard's thread has enqueued the step, the editor's tables are pinned (api_editor_quiesce)
        std::lock_guard<std::mutex> qlock(e->qmu);
        ++e->queued;
    }
    try {
        std::lock_guard<std::mutex> lock(s->mu);
        s->work.push_back(rpf_shard::Work{t});
        ++s->in_flight;
        ++s->steps_handed;
        s->last_stream = (hipStream_t)stream;
        if (!s->worker.joinable()) s->worker = std::thread(exchange_worker, s);
    } catch (...) {
        {
            std::lock_guard<std::mutex> lock(s->mu);
            if (!s->work.empty() && s->work.back().step == t) { s->work.pop_back(); --s->in_flight; --s->steps_handed; }
        }
        {
            std::lock_guard<std::mutex> qlock(e->qmu);
            --e->queued;
            e->qcv.notify_all();
        }
        delete t;
        return api_fail(RPF_ERR_GPU_COMPUTE, "GPU compute error: could not start the shard's thread");
    }
    s->cv.notify_all();
    return RPF_OK;
}

int rpf_shard_set_timeout(rpf_shard* s, int timeout_ms)
{
    if (!s) return api_fail(RPF_ERR_INVALID_ARGUMENT, "null shard");
    s->timeout_ms = timeout_ms;
    return RPF_OK;
}

int rpf_shard_flush(rpf_shard* s)
{
    if (!s) return api_fail(RPF_ERR_INVALID_ARGUMENT, "null shard");
    std::unique_lock<std::mutex> lock(s->mu);
    const int ms = shard_deadline_ms(s);
    auto idle = [&] { return s->in_flight == 0; };
    if (ms > 0) {
        if (!s->cv.wait_for(lock, std::chrono::milliseconds(ms), idle)) {
            // the thread is stuck in a call that does not return — a neighbour that never posted its half of the exchange,
            // a launch behind a full queue: say where, and leave the steps where they are (they may still complete)
            const std::string msg = "GPU compute error: " + shard_who(s) + ": step " + std::to_string(s->steps_enqueued + 1) + " of the " +
                                    std::to_string(s->steps_handed) + " handed to rpf_shard_develop_async was not enqueued within " +
                                    std::to_string(ms) + " ms; the shard's thread is in: " + s->phase.load();
            lock.unlock();
            return api_fail(RPF_ERR_GPU_COMPUTE, msg);
        }
    } else {
        s->cv.wait(lock, idle);
    }
    if (s->async_rc) {
        const int rc = s->async_rc;
        const std::string msg = shard_who(s) + ": " + s->async_error;
        s->async_rc = 0;
        s->async_error.clear();
        lock.unlock();
        return api_fail(rc, msg);
    }
    return RPF_OK;
}

int rpf_shard_drain(rpf_shard* s, void* stream)
{
    if (!s) return api_fail(RPF_ERR_INVALID_ARGUMENT, "null shard");
    int rc = rpf_shard_flush(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->proc->device));
    hipStream_t st = (hipStream_t)stream;
    const int ms = shard_deadline_ms(s);
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(ms);
    for (;;) {  // polled: a wait that can be given up (hipStreamSynchronize cannot)
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return RPF_OK;
        if (q != hipErrorNotReady) return api_hip_fail(q, "hipStreamQuery(step stream)");
        if (ms > 0 && std::chrono::steady_clock::now() > t_end) {
            uint64_t enq, handed;
            {
                std::lock_guard<std::mutex> lock(s->mu);
                enq = s->steps_enqueued; handed = s->steps_handed;
            }
            return api_fail(RPF_ERR_GPU_COMPUTE, "GPU compute error: " + shard_who(s) + ": all " + std::to_string(enq) + " of " +
                            std::to_string(handed) + " steps are enqueued but their stream has not drained within " + std::to_string(ms) +
                            " ms: a halo exchange is waiting on the device for a neighbour that has not reached the same step");
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

int rpf_shard_timing(rpf_shard* s, uint32_t* launches, double* total_ms)
{
    if (!s || !launches || !total_ms) return api_fail(RPF_ERR_INVALID_ARGUMENT, "null argument");
    int rc = rpf_shard_flush(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->proc->device));
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;
    {
        std::lock_guard<std::mutex> lock(s->mu);
        pairs.swap(s->timing);
    }
    *launches = 0;
    *total_ms = 0.0;
    hipError_t first = hipSuccess;
    for (auto& pr : pairs) {
        float ms = 0.0f;
        hipError_t he = hipEventSynchronize(pr.second);
        if (he == hipSuccess) he = hipEventElapsedTime(&ms, pr.first, pr.second);
        if (he == hipSuccess) { ++*launches; *total_ms += ms; }
        else if (first == hipSuccess) first = he;
    }
    {
        std::lock_guard<std::mutex> lock(s->mu);
        for (auto& pr : pairs) s->timing_free.push_back(pr);
    }
    if (first != hipSuccess) return api_hip_fail(first, "hipEventElapsedTime(timed launch)");
    return RPF_OK;
}

}  // extern "C"
