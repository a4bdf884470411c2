// akz_comm.hip — the one exchange step of the frame-sharded front-end, reachable from the C ABI:
// descriptor blocks of the frames a rank has just extracted travel to the ranks that match against them.
//
//   frame g lives on rank g mod N (akaze::Akaze is Copy and stateless, akaze/src/lib.rs:108: extraction shards by frame);
//   a new frame is matched against its recent predecessors g-1 .. g-k (cv-sfm/src/settings.rs:449-450
//   tracking_recent_frames = 32; the loop at cv-sfm/src/lib.rs:1462-1486), which live on the other ranks:
//     k = 1        one ring shift: rank r -> r + 1 (a rank needs its predecessor's block only)
//     k >= N - 1   an all-gather of the fixed-capacity blocks {count, cap x 64 B} (what SURVEY §8e / north_star name)
//
// RCCL is used directly (ncclSend / ncclRecv / ncclAllGather on a stream of this module), through dlopen of
// librccl.so.1 — the library has no link-time dependency on it, a single-GPU user never loads it, and a host that
// already carries a copy (PyTorch bundles one under the same soname) shares that copy.  The unique id is created by
// rank 0 (akz_comm_unique_id) and handed to the other ranks by the host application's own means (MPI, a file, a
// socket; bench.py: torch.distributed's store), then every rank calls akz_comm_create.
//
// Ordering is by events, like every other stage: a call waits for `stream_to_wait` (the stream that produced the
// send rows and/or last read the receive rows), enqueues the transfer on akz_comm_stream() and returns; consumers
// take akz_comm_stream() as their stream_to_wait.
#include <dlfcn.h>

#include "akz_common.h"

namespace {

typedef struct {
    char internal[128];
} rccl_unique_id;                       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm;                // ncclComm_t
constexpr int kRcclUint8 = 1;           // ncclUint8
constexpr int kRcclUint32 = 3;          // ncclUint32

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(rccl_unique_id*) = nullptr;
    int (*CommInitRank)(rccl_comm*, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(rccl_comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

// Loaded once, on first use, by whichever thread gets there first (function-local static: the C++11 runtime serialises
// the initialisation, every other caller waits and then sees the finished table or the recorded failure).
RcclApi* rccl()
{
    static const RcclApi api = []() {
        RcclApi a;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
        }
        if (!a.handle) return a;
        bool all = true;
#define RCCL_SYM(field, sym)                                               \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, sym));   \
    all = all && a.field != nullptr;
        RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
        RCCL_SYM(CommInitRank, "ncclCommInitRank")
        RCCL_SYM(CommDestroy, "ncclCommDestroy")
        RCCL_SYM(GroupStart, "ncclGroupStart")
        RCCL_SYM(GroupEnd, "ncclGroupEnd")
        RCCL_SYM(Send, "ncclSend")
        RCCL_SYM(Recv, "ncclRecv")
        RCCL_SYM(AllGather, "ncclAllGather")
        RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
        a.ok = all;
        return a;
    }();
    return api.ok ? const_cast<RcclApi*>(&api) : nullptr;
}

thread_local int g_last_rccl = 0;

}  // namespace

#define AKZ_RCCL(call)               \
    do {                             \
        int r_ = (call);             \
        if (r_ != 0) {               \
            g_last_rccl = r_;        \
            return AKZ_E_COMM;       \
        }                            \
    } while (0)

// Between ncclGroupStart and ncclGroupEnd (inside a transfer function: c, e0, e1 in scope): a failing call must not
// leave the group open (every later RCCL call of the process would join it) — close it, keep the first error, hand the
// timing events back.
#define AKZ_RCCL_IN_GROUP(R, call)                 \
    do {                                           \
        int r_ = (call);                           \
        if (r_ != 0) {                             \
            g_last_rccl = r_;                      \
            (R)->GroupEnd();                       \
            if (e0) c->pool.push_back(e0);         \
            if (e1) c->pool.push_back(e1);         \
            return AKZ_E_COMM;                     \
        }                                          \
    } while (0)

// ncclGroupStart / ncclGroupEnd themselves (same scope): no group is open after either fails, but the timing events taken by
// comm_begin still go back to the pool.
#define AKZ_RCCL_GROUP_EDGE(call)                  \
    do {                                           \
        int r_ = (call);                           \
        if (r_ != 0) {                             \
            g_last_rccl = r_;                      \
            if (e0) c->pool.push_back(e0);         \
            if (e1) c->pool.push_back(e1);         \
            return AKZ_E_COMM;                     \
        }                                          \
    } while (0)

// most event pairs kept un-resolved when timing is on and akz_comm_timing() is never called: beyond it the oldest are
// resolved (their transfers are long finished) before another pair is added
constexpr size_t kCommMaxPending = 256;

struct akz_comm {
    int device = 0, rank = 0, world = 1;
    rccl_comm comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    // exposed time of the exchange: events around every transfer (akz_comm_timing)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<hipEvent_t> pool;
    double ms = 0.0;
    uint64_t calls = 0, bytes = 0;
    bool timing = false;
};

extern "C" int32_t akz_comm_unique_id(uint8_t* id128)
{
    return akz_guard([&]() -> int32_t {
        if (!id128) return AKZ_E_INVALID;
        RcclApi* R = rccl();
        if (!R) return AKZ_E_COMM;
        rccl_unique_id id;
        AKZ_RCCL(R->GetUniqueId(&id));
        memcpy(id128, id.internal, 128);
        return AKZ_OK;
    });
}

extern "C" int32_t akz_comm_create(const uint8_t* id128, int32_t rank, int32_t world, int32_t device, akz_comm** out)
{
    return akz_guard([&]() -> int32_t {
        if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return AKZ_E_INVALID;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return AKZ_E_NO_DEVICE;
        RcclApi* R = rccl();
        if (!R) return AKZ_E_COMM;
        AKZ_HIP(hipSetDevice(device));
        // (owned until the communicator is complete: every failure below frees the half-built object, its stream and event)
        struct Undo {
            akz_comm* c;
            ~Undo()
            {
                if (!c) return;
                if (c->ev) hipEventDestroy(c->ev);
                if (c->stream) hipStreamDestroy(c->stream);
                delete c;
            }
        } undo{new akz_comm()};
        akz_comm* c = undo.c;
        c->device = device;
        c->rank = rank;
        c->world = world;
        AKZ_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        AKZ_HIP(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
        rccl_unique_id id;
        memcpy(id.internal, id128, 128);
        AKZ_RCCL(R->CommInitRank(&c->comm, world, id, rank));
        undo.c = nullptr;
        *out = c;
        return AKZ_OK;
    });
}

extern "C" int32_t akz_comm_destroy(akz_comm* c)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_OK;
        hipSetDevice(c->device);
        if (c->stream) hipStreamSynchronize(c->stream);
        RcclApi* R = rccl();
        if (R && c->comm) R->CommDestroy(c->comm);
        for (auto& pr : c->pending) {
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
        for (hipEvent_t e : c->pool) hipEventDestroy(e);
        if (c->ev) hipEventDestroy(c->ev);
        if (c->stream) hipStreamDestroy(c->stream);
        delete c;
        return AKZ_OK;
    });
}

extern "C" void* akz_comm_stream(akz_comm* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int32_t akz_comm_rank(akz_comm* c) { return c ? c->rank : -1; }
extern "C" int32_t akz_comm_world(akz_comm* c) { return c ? c->world : 0; }

extern "C" int32_t akz_comm_sync(akz_comm* c)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        AKZ_HIP(hipStreamSynchronize(c->stream));
        return AKZ_OK;
    });
}

static void comm_resolve(akz_comm* c, size_t keep)
{
    size_t n = c->pending.size() > keep ? c->pending.size() - keep : 0;
    for (size_t i = 0; i < n; ++i) {
        auto& pr = c->pending[i];
        float t = 0.0f;
        hipEventSynchronize(pr.second);
        if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) c->ms += (double)t;
        c->pool.push_back(pr.first);
        c->pool.push_back(pr.second);
    }
    c->pending.erase(c->pending.begin(), c->pending.begin() + (long)n);
}
static int32_t comm_begin(akz_comm* c, void* stream_to_wait, hipEvent_t* e0, hipEvent_t* e1)
{
    AKZ_HIP(hipSetDevice(c->device));
    if (stream_to_wait) {
        AKZ_HIP(hipEventRecord(c->ev, akz_wait_stream(stream_to_wait)));
        AKZ_HIP(hipStreamWaitEvent(c->stream, c->ev, 0));
    }
    *e0 = *e1 = nullptr;
    if (c->timing) {
        if (c->pending.size() >= kCommMaxPending) comm_resolve(c, kCommMaxPending / 2);
        auto take = [&]() {
            hipEvent_t e = nullptr;
            if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        };
        *e0 = take();
        *e1 = take();
        if (*e0 && *e1) AKZ_HIP(hipEventRecord(*e0, c->stream));
    }
    return AKZ_OK;
}
static int32_t comm_end(akz_comm* c, hipEvent_t e0, hipEvent_t e1, uint64_t bytes)
{
    if (e0 && e1) {
        AKZ_HIP(hipEventRecord(e1, c->stream));
        c->pending.emplace_back(e0, e1);
    }
    c->calls += 1;
    c->bytes += bytes;
    return AKZ_OK;
}

// Ring shift of descriptor blocks: this rank's n_frames blocks ([n_frames][cap_per_img][64] bytes + [n_frames] u32 counts)
// go to rank + 1, the blocks of rank - 1 arrive in d_recv_*; one grouped send/recv pair per buffer.
extern "C" int32_t akz_comm_shift_blocks(akz_comm* c, const void* d_descs, const void* d_counts, uint32_t n_frames, uint32_t cap_per_img,
                                         void* d_recv_descs, void* d_recv_counts, void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_descs || !d_counts || !d_recv_descs || !d_recv_counts || cap_per_img == 0) return AKZ_E_INVALID;
        if (n_frames == 0) return AKZ_OK;
        RcclApi* R = rccl();
        if (!R) return AKZ_E_COMM;
        hipEvent_t e0, e1;
        AKZ_TRY(comm_begin(c, stream_to_wait, &e0, &e1));
        const size_t db = (size_t)n_frames * cap_per_img * 64, cb = (size_t)n_frames;
        const int nxt = (c->rank + 1) % c->world, prv = (c->rank + c->world - 1) % c->world;
        AKZ_RCCL_GROUP_EDGE(R->GroupStart());
        AKZ_RCCL_IN_GROUP(R, R->Send(d_descs, db, kRcclUint8, nxt, c->comm, c->stream));
        AKZ_RCCL_IN_GROUP(R, R->Send(d_counts, cb, kRcclUint32, nxt, c->comm, c->stream));
        AKZ_RCCL_IN_GROUP(R, R->Recv(d_recv_descs, db, kRcclUint8, prv, c->comm, c->stream));
        AKZ_RCCL_IN_GROUP(R, R->Recv(d_recv_counts, cb, kRcclUint32, prv, c->comm, c->stream));
        AKZ_RCCL_GROUP_EDGE(R->GroupEnd());
        return comm_end(c, e0, e1, db + 4 * cb);
    });
}

// All-gather of descriptor blocks: every rank contributes n_frames blocks; d_all_descs [world][n_frames][cap][64] and
// d_all_counts [world][n_frames] hold every rank's (rank-major) afterwards — the owner of frame g then has the
// descriptors of g-1 .. g-k for any k.
extern "C" int32_t akz_comm_allgather_blocks(akz_comm* c, const void* d_descs, const void* d_counts, uint32_t n_frames, uint32_t cap_per_img,
                                             void* d_all_descs, void* d_all_counts, void* stream_to_wait)
{
    return akz_guard([&]() -> int32_t {
        if (!c || !d_descs || !d_counts || !d_all_descs || !d_all_counts || cap_per_img == 0) return AKZ_E_INVALID;
        if (n_frames == 0) return AKZ_OK;
        RcclApi* R = rccl();
        if (!R) return AKZ_E_COMM;
        hipEvent_t e0, e1;
        AKZ_TRY(comm_begin(c, stream_to_wait, &e0, &e1));
        const size_t db = (size_t)n_frames * cap_per_img * 64, cb = (size_t)n_frames;
        AKZ_RCCL_GROUP_EDGE(R->GroupStart());
        AKZ_RCCL_IN_GROUP(R, R->AllGather(d_descs, d_all_descs, db, kRcclUint8, c->comm, c->stream));
        AKZ_RCCL_IN_GROUP(R, R->AllGather(d_counts, d_all_counts, cb, kRcclUint32, c->comm, c->stream));
        AKZ_RCCL_GROUP_EDGE(R->GroupEnd());
        return comm_end(c, e0, e1, (db + 4 * cb) * (size_t)c->world);
    });
}

extern "C" int32_t akz_comm_timing(akz_comm* c, int32_t enable, double* ms, uint64_t* calls, uint64_t* bytes, int32_t reset)
{
    return akz_guard([&]() -> int32_t {
        if (!c) return AKZ_E_INVALID;
        AKZ_HIP(hipSetDevice(c->device));
        comm_resolve(c, 0);
        if (ms) *ms = c->ms;
        if (calls) *calls = c->calls;
        if (bytes) *bytes = c->bytes;
        if (reset) {
            c->ms = 0.0;
            c->calls = c->bytes = 0;
        }
        c->timing = enable != 0;
        return AKZ_OK;
    });
}

extern "C" const char* akz_comm_last_error_string(void)
{
    RcclApi* R = rccl();
    if (!R) return "librccl.so.1 could not be loaded";
    return g_last_rccl ? R->GetErrorString(g_last_rccl) : "no error";
}
