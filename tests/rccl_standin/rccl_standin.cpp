// rccl_standin.cpp — TEST INFRASTRUCTURE ONLY (never shipped, never linked: loaded through the RPF_RCCL_LIB hook of
// rpf_shard.cpp by tests/test_shard_multiproc.py).  The eight RCCL entry points librpf_hip.so resolves, re-stated so that
// SEVERAL PROCESSES ON ONE GPU can be the ranks of a communicator — which RCCL itself refuses ("duplicate GPU") — and the
// library's N > 1 path (rpf_shard_create(world > 1), both communicators of the two-pipeline schedule, rpf_shard_develop_async,
// bench.py's library branch) meets a second process before it meets the driver's 8-GPU node.
//
// Transport: a POSIX shared-memory segment per communicator (named by the unique id), one ring of four slots per ordered
// pair of ranks.  ncclSend copies the rows device -> slot on the call's stream, waits for that stream, then publishes the
// slot; ncclRecv waits for the slot (bounded: RPF_STANDIN_TIMEOUT_MS, default 20 s -> ncclSystemError), copies slot ->
// device on the stream and waits again.  Host-synchronous on purpose: order and data are what RCCL's would be, timings are
// not.  ncclGroupEnd runs the group's sends before its receives, so a rank that is its own neighbour works too.
// RPF_STANDIN_DROP_SENDS_FROM=<rank>: that rank's sends are silently dropped — its neighbours' receives never complete
// (the wedge tests/test_shard_multiproc.py uses to see the library's own deadline fire).
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <new>
#include <vector>

namespace {

constexpr int MAXW = 8, SLOTS = 4;
constexpr size_t SLOT_BYTES = 128 * 1024;

struct Chan {
    std::atomic<uint64_t> sent, consumed;
};
struct Ctl {
    std::atomic<uint32_t> magic;  // set by the creator once the segment is sized and zeroed
    std::atomic<int> alive;
    int world;
    Chan chan[MAXW][MAXW];  // [src][dst]
};
constexpr uint32_t MAGIC = 0x52504653u;

size_t segment_bytes() { return sizeof(Ctl) + (size_t)MAXW * MAXW * SLOTS * SLOT_BYTES; }

struct Comm {
    Ctl* ctl;
    unsigned char* data;
    int rank, world;
    char name[64];
    uint64_t recv_seq[MAXW];  // next sequence number expected from each source
    uint64_t send_seq[MAXW];
};
unsigned char* slot(Comm* c, int src, int dst, uint64_t seq)
{
    return c->data + (((size_t)src * MAXW + dst) * SLOTS + (seq % SLOTS)) * SLOT_BYTES;
}

struct Op {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    Comm* comm;
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

double now_ms()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}
double timeout_ms()
{
    const char* v = getenv("RPF_STANDIN_TIMEOUT_MS");
    return v ? atof(v) : 20000.0;
}
int drop_rank()
{
    const char* v = getenv("RPF_STANDIN_DROP_SENDS_FROM");
    return v ? atoi(v) : -1;
}
size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

ncclResult_t run(const Op& o)
{
    Comm* c = o.comm;
    if (o.bytes > SLOT_BYTES) return ncclInvalidArgument;
    const double t_end = now_ms() + timeout_ms();
    if (o.send) {
        if (c->rank == drop_rank()) return ncclSuccess;  // the wedge: nothing ever arrives
        Chan& ch = c->ctl->chan[c->rank][o.peer];
        const uint64_t seq = c->send_seq[o.peer]++;
        while (seq - ch.consumed.load(std::memory_order_acquire) >= (uint64_t)SLOTS) {  // ring full: the receiver is behind
            if (now_ms() > t_end) return ncclSystemError;
            usleep(20);
        }
        if (hipMemcpyAsync(slot(c, c->rank, o.peer, seq), o.buf, o.bytes, hipMemcpyDeviceToHost, o.stream) != hipSuccess ||
            hipStreamSynchronize(o.stream) != hipSuccess)
            return ncclUnhandledCudaError;
        ch.sent.store(seq + 1, std::memory_order_release);
        return ncclSuccess;
    }
    Chan& ch = c->ctl->chan[o.peer][c->rank];
    const uint64_t seq = c->recv_seq[o.peer]++;
    while (ch.sent.load(std::memory_order_acquire) <= seq) {
        if (now_ms() > t_end) return ncclSystemError;
        usleep(20);
    }
    if (hipMemcpyAsync(o.buf, slot(c, o.peer, c->rank, seq), o.bytes, hipMemcpyHostToDevice, o.stream) != hipSuccess ||
        hipStreamSynchronize(o.stream) != hipSuccess)
        return ncclUnhandledCudaError;
    ch.consumed.store(seq + 1, std::memory_order_release);
    return ncclSuccess;
}

ncclResult_t submit(const Op& o)
{
    if (g_depth > 0) {
        g_ops.push_back(o);
        return ncclSuccess;
    }
    return run(o);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    unsigned long long r = 0;
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) { (void)!fread(&r, sizeof(r), 1, f); fclose(f); }
    snprintf(id->internal, sizeof(id->internal), "/rpf_standin_%d_%016llx", (int)getpid(), r);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAXW || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (strncmp(id.internal, "/rpf_standin_", 13) != 0) return ncclInvalidArgument;
    const size_t bytes = segment_bytes();
    bool creator = true;
    int fd = shm_open(id.internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) {
        creator = false;
        fd = shm_open(id.internal, O_RDWR, 0600);
    }
    if (fd < 0) return ncclSystemError;
    if (creator && ftruncate(fd, (off_t)bytes) != 0) { close(fd); return ncclSystemError; }
    const double t_end = now_ms() + timeout_ms();
    if (!creator) {  // until the creator has sized it
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < bytes) {
            if (now_ms() > t_end) { close(fd); return ncclSystemError; }
            usleep(100);
        }
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    Ctl* ctl = static_cast<Ctl*>(p);
    if (creator) {  // (a fresh segment is zero-filled: counters start at 0)
        ctl->world = nranks;
        ctl->magic.store(MAGIC, std::memory_order_release);
    }
    while (ctl->magic.load(std::memory_order_acquire) != MAGIC) {
        if (now_ms() > t_end) { munmap(p, bytes); return ncclSystemError; }
        usleep(100);
    }
    if (ctl->world != nranks) { munmap(p, bytes); return ncclInvalidArgument; }
    Comm* c = new (std::nothrow) Comm;
    if (!c) { munmap(p, bytes); return ncclSystemError; }
    memset(c, 0, sizeof(*c));
    c->ctl = ctl;
    c->data = static_cast<unsigned char*>(p) + sizeof(Ctl);
    c->rank = rank;
    c->world = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    const int arrived = ctl->alive.fetch_add(1) + 1;
    (void)arrived;
    // collective like the real one: return when every rank has joined
    while (ctl->alive.load(std::memory_order_acquire) < nranks) {
        if (now_ms() > t_end) { munmap(p, bytes); delete c; return ncclSystemError; }
        usleep(100);
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    shm_unlink(c->name);  // the mappings of the other ranks stay valid; the name goes with the first rank to leave
    munmap(c->ctl, segment_bytes());
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    ncclResult_t first = ncclSuccess;
    for (int pass = 0; pass < 2; ++pass)  // sends, then receives
        for (const Op& o : ops)
            if (o.send == (pass == 0)) {
                const ncclResult_t r = run(o);
                if (r != ncclSuccess && first == ncclSuccess) first = r;
            }
    return first;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || !buf || peer < 0 || peer >= c->world) return ncclInvalidArgument;
    return submit(Op{true, const_cast<void*>(buf), count * type_bytes(type), peer, c, stream});
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || !buf || peer < 0 || peer >= c->world) return ncclInvalidArgument;
    return submit(Op{false, buf, count * type_bytes(type), peer, c, stream});
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "stand-in: timed out waiting for a peer (unhandled system error)";
    case ncclInvalidArgument: return "stand-in: invalid argument";
    case ncclInvalidUsage: return "stand-in: invalid usage";
    case ncclUnhandledCudaError: return "stand-in: HIP call failed";
    default: return "stand-in: error";
    }
}

}  // extern "C"
