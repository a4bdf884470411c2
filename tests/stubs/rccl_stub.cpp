// rccl_stub.cpp — a stand-in for librccl.so.1 that lets ONE process on ONE GPU play every rank of a communicator.
//
// TEST INFRASTRUCTURE ONLY (tests/test_comm_stub.py builds it into tests/stubs/build/librccl.so.1 and puts that
// directory first on LD_LIBRARY_PATH of a subprocess that has not loaded PyTorch's bundled RCCL).  It implements the nine
// symbols cv_amd/csrc/akz_comm.hip resolves with dlsym:
//   * every call is appended to a log the test reads back (rccl_stub_log): which rank sent / received how many
//     elements of which type to / from which peer, in which group — the argument layout of akz_comm_shift_blocks and
//     akz_comm_allgather_blocks for world = 2 and 3 is asserted from it;
//   * the data really moves: sends and receives of all communicators created with the same unique id are matched in
//     posting order per (source rank, destination rank) and executed as device-to-device copies once both sides are
//     posted; an all-gather completes when every rank has posted its contribution.  The test can therefore hold the
//     received rows to the sent rows, byte for byte, without a second GPU;
//   * rccl_stub_fail_after(n) makes the n-th following data call fail, so the test sees that a failure inside a group
//     still closes the group (ncclGroupEnd is called) and frees nothing twice.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <vector>

namespace {
struct Comm {
    int rank, world;
    char id[128];
};
struct Post {
    int kind;   // 0 send, 1 recv, 2 all-gather
    Comm* comm;
    const void* src;
    void* dst;
    size_t count;
    int type, peer;
    hipStream_t stream;
    bool done;
};
struct LogRec {
    int32_t kind, rank, peer, type, in_group, world;   // kind: 0 send, 1 recv, 2 allgather, 3 group start, 4 group end, 5 init, 6 destroy
    uint64_t count;
};
std::mutex g_mu;
std::vector<Post> g_posts;
std::vector<LogRec> g_log;
int g_group_depth = 0, g_fail_in = -1, g_next_id = 1;

size_t type_size(int t) { return t == 1 ? 1 : t == 3 ? 4 : 0; }   // ncclUint8 = 1, ncclUint32 = 3 (the two akz_comm uses)

bool same_id(const Comm* a, const Comm* b) { return memcmp(a->id, b->id, 128) == 0; }

// executes every transfer whose partners are all posted (called with the lock held, outside any group)
void progress()
{
    for (size_t i = 0; i < g_posts.size(); ++i) {
        Post& s = g_posts[i];
        if (s.done || s.kind != 0) continue;
        for (size_t j = 0; j < g_posts.size(); ++j) {
            Post& r = g_posts[j];
            if (r.done || r.kind != 1 || !same_id(s.comm, r.comm)) continue;
            if (r.comm->rank != s.peer || r.peer != s.comm->rank || r.type != s.type || r.count != s.count) continue;
            hipStreamSynchronize(s.stream);
            hipStreamSynchronize(r.stream);
            hipMemcpy(r.dst, s.src, s.count * type_size(s.type), hipMemcpyDeviceToDevice);
            s.done = r.done = true;
            break;
        }
    }
    // all-gathers: the k-th all-gather post of every rank of one id belongs together
    for (size_t i = 0; i < g_posts.size(); ++i) {
        Post& a = g_posts[i];
        if (a.done || a.kind != 2) continue;
        std::vector<Post*> set(a.comm->world, nullptr);
        for (size_t j = 0; j < g_posts.size(); ++j) {
            Post& b = g_posts[j];
            if (b.done || b.kind != 2 || !same_id(a.comm, b.comm) || b.count != a.count || b.type != a.type) continue;
            if (!set[b.comm->rank]) set[b.comm->rank] = &b;
        }
        bool all = true;
        for (Post* p : set) all = all && p;
        if (!all) continue;
        for (Post* p : set) hipStreamSynchronize(p->stream);
        const size_t bytes = a.count * type_size(a.type);
        for (Post* d : set)
            for (Post* s : set) hipMemcpy((char*)d->dst + (size_t)s->comm->rank * bytes, s->src, bytes, hipMemcpyDeviceToDevice);
        for (Post* p : set) p->done = true;
    }
}
bool should_fail()
{
    if (g_fail_in < 0) return false;
    if (g_fail_in == 0) {
        g_fail_in = -1;
        return true;
    }
    --g_fail_in;
    return false;
}
void log(int kind, const Comm* c, int peer, int type, uint64_t count)
{
    g_log.push_back(LogRec{kind, c ? c->rank : -1, peer, type, g_group_depth, c ? c->world : 0, count});
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id->internal, 0, 128);
    memcpy(id->internal, "akz-rccl-stub", 13);
    id->internal[16] = (char)g_next_id++;
    return 0;
}
int ncclCommInitRank(void** comm, int world, ncclUniqueId id, int rank)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (should_fail()) return 2;
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    memcpy(c->id, id.internal, 128);
    *comm = c;
    log(5, c, -1, 0, 0);
    return 0;
}
int ncclCommDestroy(void* comm)
{
    std::lock_guard<std::mutex> lk(g_mu);
    log(6, (Comm*)comm, -1, 0, 0);
    delete (Comm*)comm;
    return 0;
}
int ncclGroupStart()
{
    std::lock_guard<std::mutex> lk(g_mu);
    log(3, nullptr, -1, 0, 0);
    ++g_group_depth;
    return 0;
}
int ncclGroupEnd()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_group_depth > 0) --g_group_depth;
    log(4, nullptr, -1, 0, 0);
    if (g_group_depth == 0) progress();
    return 0;
}
int ncclSend(const void* buf, size_t count, int type, int peer, void* comm, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (should_fail()) return 3;
    log(0, (Comm*)comm, peer, type, count);
    g_posts.push_back(Post{0, (Comm*)comm, buf, nullptr, count, type, peer, stream, false});
    if (g_group_depth == 0) progress();
    return 0;
}
int ncclRecv(void* buf, size_t count, int type, int peer, void* comm, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (should_fail()) return 3;
    log(1, (Comm*)comm, peer, type, count);
    g_posts.push_back(Post{1, (Comm*)comm, nullptr, buf, count, type, peer, stream, false});
    if (g_group_depth == 0) progress();
    return 0;
}
int ncclAllGather(const void* src, void* dst, size_t count, int type, void* comm, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (should_fail()) return 3;
    log(2, (Comm*)comm, -1, type, count);
    g_posts.push_back(Post{2, (Comm*)comm, src, dst, count, type, -1, stream, false});
    if (g_group_depth == 0) progress();
    return 0;
}
const char* ncclGetErrorString(int r) { return r == 0 ? "stub: no error" : r == 2 ? "stub: injected init failure" : "stub: injected failure"; }

// ---- the test's side channel ----
int rccl_stub_log(int32_t* out, int cap_records)   // 8 x int32 per record: kind, rank, peer, type, in_group, world, count lo, count hi
{
    std::lock_guard<std::mutex> lk(g_mu);
    int n = (int)g_log.size() < cap_records ? (int)g_log.size() : cap_records;
    for (int i = 0; i < n; ++i) {
        const LogRec& r = g_log[i];
        int32_t* o = out + 8 * i;
        o[0] = r.kind; o[1] = r.rank; o[2] = r.peer; o[3] = r.type; o[4] = r.in_group; o[5] = r.world;
        o[6] = (int32_t)(r.count & 0xffffffffu); o[7] = (int32_t)(r.count >> 32);
    }
    return (int)g_log.size();
}
void rccl_stub_reset()
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_log.clear();
    g_posts.clear();
    g_group_depth = 0;
    g_fail_in = -1;
}
void rccl_stub_fail_after(int n) { std::lock_guard<std::mutex> lk(g_mu); g_fail_in = n; }
int rccl_stub_group_depth() { std::lock_guard<std::mutex> lk(g_mu); return g_group_depth; }
int rccl_stub_unfinished()
{
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (const Post& p : g_posts) n += p.done ? 0 : 1;
    return n;
}
}
