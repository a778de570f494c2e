// api_comm.cpp -- one frame of the chain sharded by row bands over the GPUs of a node, the exchanges below the C ABI (SURVEY.md 8e;
// no reference counterpart: DiligentFX is single-GPU).
//
//   mifx_comm                      one rank's endpoint: RCCL (ncclCommInitRank over an ncclUniqueId the caller distributes) or, for tests on a single GPU, an
//                                  in-process group of ranks driven by one host thread each (the same exchange code, device copies instead of RCCL)
//   mifx_chain_execute_sharded     the four phases of mifx_chain_execute_phase with the three exchanges between them:
//       after phase 0   the band rows of the shaded radiance go to every rank (the SSR ray march reads the whole frame): grouped ncclSend / ncclRecv of
//                       row slabs on a side stream, overlapping phase 1 (PostFX prep + SSAO do not read the radiance); joined before phase 2
//       after phase 2   the rows of Bloom level `gather_level` each rank owns go to every rank (the coarse tail is computed whole everywhere)
//       after phase 3   the first / last rows of the five history planes go to the upper / lower neighbour (the next frame reprojects across the band edge)
//   xGMI is point to point and fully connected: every transfer is a direct send to the rank that needs the rows, no ring, no staging copies (full-frame planes
//   in global coordinates make a block of rows one contiguous slab on both sides).  Every rank derives every other rank's rows from the shared cuts, so
//   both sides of a transfer agree on its size without a round of communication.
// RCCL is loaded with dlopen at the first mifx_comm call: libmifx.so itself does not depend on it.  Every failure of the communication layer is MIFX_ERR_COMM.
#include <cstdlib>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "mifx_objects.h"

using namespace mifx;

namespace
{
// ------------------------------------------------------------------------------------------------ RCCL entry points (dlopen)
struct Rccl
{
    void* lib = nullptr;
    decltype(&ncclGetUniqueId)    GetUniqueId    = nullptr;
    decltype(&ncclCommInitRank)   CommInitRank   = nullptr;
    decltype(&ncclCommDestroy)    CommDestroy    = nullptr;
    decltype(&ncclSend)           Send           = nullptr;
    decltype(&ncclRecv)           Recv           = nullptr;
    decltype(&ncclGroupStart)     GroupStart     = nullptr;
    decltype(&ncclGroupEnd)       GroupEnd       = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort)      CommAbort      = nullptr; // optional
    decltype(&ncclCommCount)      CommCount      = nullptr; // optional (mifx_comm_get_stats: how many ranks RCCL itself sees)
};
const Rccl* rccl()
{
    static Rccl        r;
    static std::mutex  m;
    std::lock_guard<std::mutex> lock(m);
    if (r.lib) return &r;
    // MIFX_RCCL_PATH names the library explicitly: a C++ host next to a Python package finds torch's private librccl that way (the bare soname only resolves in a
    // process that has mapped it already or through the loader path), and the tests load their multi-process stand-in (tests/fake_rccl) through it.
    void*       lib      = nullptr;
    const char* explicitPath = std::getenv("MIFX_RCCL_PATH");
    if (explicitPath != nullptr && explicitPath[0] != 0)
    {
        lib = dlopen(explicitPath, RTLD_NOW | RTLD_LOCAL);
        if (!lib)
        {
            set_error("RCCL is not available: MIFX_RCCL_PATH=%s: %s", explicitPath, dlerror());
            return nullptr;
        }
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
        if (lib == nullptr) lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!lib)
    {
        set_error("RCCL is not available: %s (MIFX_RCCL_PATH names the library explicitly)", dlerror());
        return nullptr;
    }
    Rccl t;
    t.lib = lib;
#define MIFX_SYM(n) t.n = reinterpret_cast<decltype(t.n)>(dlsym(lib, "nccl" #n))
    MIFX_SYM(GetUniqueId); MIFX_SYM(CommInitRank); MIFX_SYM(CommDestroy); MIFX_SYM(Send); MIFX_SYM(Recv); MIFX_SYM(GroupStart); MIFX_SYM(GroupEnd); MIFX_SYM(GetErrorString); MIFX_SYM(CommAbort); MIFX_SYM(CommCount);
#undef MIFX_SYM
    if (!t.GetUniqueId || !t.CommInitRank || !t.CommDestroy || !t.Send || !t.Recv || !t.GroupStart || !t.GroupEnd || !t.GetErrorString)
    {
        set_error("librccl lacks one of the entry points ncclGetUniqueId / CommInitRank / CommDestroy / Send / Recv / GroupStart / GroupEnd");
        dlclose(lib);
        return nullptr;
    }
    r = t;
    return &r;
}
#define MIFX_NCCL_CHECK(call)                                                                                     \
    do {                                                                                                          \
        const ncclResult_t mifx_r_ = (call);                                                                      \
        if (mifx_r_ != ncclSuccess)                                                                               \
        {                                                                                                         \
            set_error("%s failed: %s", #call, rccl() ? rccl()->GetErrorString(mifx_r_) : "RCCL unavailable");     \
            return MIFX_ERR_COMM;                                                                                 \
        }                                                                                                         \
    } while (0)

// ------------------------------------------------------------------------------------------------ in-process group (tests on one GPU)
// Every rank is driven by its own host thread.  A send posts {pointer, event recorded on the sender's stream}; the matching receive waits for
// the post, makes its stream wait for the event and copies device to device; the sender's stream then waits for the receiver's copy before it
// may overwrite the rows.  Posts of one (source, destination) pair are matched in order, like RCCL's.
struct LocalPost
{
    const void* src;
    size_t      bytes;
    hipEvent_t  ready  = nullptr; // recorded by the sender after the work that produced the rows
    hipEvent_t  copied = nullptr; // recorded by the receiver after its copy; the sender waits on it
    bool        done = false;
    LocalPost() = default;
    LocalPost(const LocalPost&) = delete;
    // the events belong to the post and go with its last holder (the sender's list, the box, a receiver that has popped it): nobody destroys an event another
    // thread may still record or wait on
    ~LocalPost()
    {
        if (ready) (void)hipEventDestroy(ready);
        if (copied) (void)hipEventDestroy(copied);
    }
};
struct LocalGroup
{
    int                     world;
    std::mutex              m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<LocalPost>>> box; // (source, destination) -> posts in order
};
struct PendingOp
{
    bool   send;
    void*  ptr;
    size_t bytes;
    int    peer;
};
} // namespace

struct mifx_comm
{
    int          device = 0, rank = 0, world = 1;
    ncclComm_t   nccl = nullptr;               // RCCL endpoint (null for the in-process group, and after an abort)
    bool         rccl_endpoint = false;        // created by mifx_comm_create (stays true when an abort has released `nccl`)
    std::shared_ptr<LocalGroup> group;         // in-process group (null for RCCL)
    std::vector<PendingOp> pending;            // operations of the open group
    bool         open = false;
    bool         broken = false;               // an RCCL call failed inside an open group with operations already queued: the communicator is not used again
    int          queuedInGroup = 0;
    hipStream_t  side = nullptr;               // the radiance all-gather runs here, beside phase 1
    std::vector<mifx_chain*> users;            // chains whose sharding borrows this communicator (mifx_chain_set_sharding): detached when either side goes away
    // what the endpoint has moved (mifx_comm_get_stats): host counters, bumped when an operation is handed to the transport
    uint64_t     groups = 0, bytesSent = 0, bytesReceived = 0;
    // mifx_comm_set_timing: every exchange group of a frame (SSAO halos, Bloom gather, TAA + SSR halos, the luminance rows) between two timing events on the stream it is
    // issued on -- start: the stream reaches the group, stop: all of its transfers are done on this rank.  Read and released by mifx_comm_get_stats.
    bool         timing = false;
    struct Timed { hipEvent_t start = nullptr, stop = nullptr; };
    std::vector<Timed> timed;
    static constexpr size_t kMaxTimed = 4096;
    void time_start(hipStream_t s)
    {
        if (!timing || timed.size() >= kMaxTimed) return;
        Timed t;
        if (hipEventCreate(&t.start) != hipSuccess || hipEventCreate(&t.stop) != hipSuccess || hipEventRecord(t.start, s) != hipSuccess)
        {
            if (t.start) (void)hipEventDestroy(t.start);
            if (t.stop) (void)hipEventDestroy(t.stop);
            return;
        }
        timed.push_back(t);
        timedOpen = true;
    }
    void time_stop(hipStream_t s)
    {
        if (!timedOpen) return;
        timedOpen = false;
        if (hipEventRecord(timed.back().stop, s) != hipSuccess) drop_timed(timed.size() - 1);
    }
    void drop_timed(size_t i)
    {
        (void)hipEventDestroy(timed[i].start);
        (void)hipEventDestroy(timed[i].stop);
        timed.erase(timed.begin() + long(i));
    }
    bool         timedOpen = false;

    // Closes a group that an error path left open (GroupGuard): RCCL must see its GroupEnd or every later call on this thread nests inside the abandoned group; the
    // in-process group just forgets what was queued (nothing has been posted before end()).
    // With operations already queued, ending the group launches a PARTIAL exchange whose matching operations the peers never post: its kernels would spin on the device
    // for ever.  So the group is closed first (the host call returns once the kernels are queued) and the communicator is then aborted -- ncclCommAbort is what makes
    // kernels that wait for a peer give up -- and forgotten: the abort has released it, mifx_comm_destroy must not hand it to ncclCommDestroy again (round 4 did both,
    // and closed the group AFTER the abort, on a communicator the library had already freed).  The endpoint stays `broken`: every later begin() is refused.
    void abort_group()
    {
        if (timedOpen) // (an exchange that did not complete has no duration)
        {
            timedOpen = false;
            drop_timed(timed.size() - 1);
        }
        if (!open) return;
        open = false;
        pending.clear();
        if (nccl && rccl())
        {
            (void)rccl()->GroupEnd();
            if (queuedInGroup > 0)
            {
                broken = true;
                if (rccl()->CommAbort)
                {
                    (void)rccl()->CommAbort(nccl);
                    nccl = nullptr; // released by the abort
                }
            }
        }
        queuedInGroup = 0;
    }

    mifx_status begin()
    {
        pending.clear();
        MIFX_CHECK(refuse_broken());
        if (nccl) MIFX_NCCL_CHECK(rccl()->GroupStart());
        open = true;
        queuedInGroup = 0;
        return MIFX_OK;
    }
    mifx_status refuse_broken() const
    {
        if (!broken) return MIFX_OK;
        set_error("the communicator was aborted after a failed exchange: create a new one (mifx_comm_create)");
        return MIFX_ERR_COMM;
    }
    mifx_status send(const void* p, size_t bytes, int peer, hipStream_t s)
    {
        MIFX_CHECK(refuse_broken());
        if (bytes == 0) return MIFX_OK;
        if (nccl)
        {
            MIFX_NCCL_CHECK(rccl()->Send(p, bytes, ncclInt8, peer, nccl, s));
            ++queuedInGroup;
        }
        else pending.push_back(PendingOp{true, const_cast<void*>(p), bytes, peer});
        bytesSent += bytes;
        return MIFX_OK;
    }
    mifx_status recv(void* p, size_t bytes, int peer, hipStream_t s)
    {
        MIFX_CHECK(refuse_broken());
        if (bytes == 0) return MIFX_OK;
        if (nccl)
        {
            MIFX_NCCL_CHECK(rccl()->Recv(p, bytes, ncclInt8, peer, nccl, s));
            ++queuedInGroup;
        }
        else pending.push_back(PendingOp{false, p, bytes, peer});
        bytesReceived += bytes;
        return MIFX_OK;
    }
    mifx_status end(hipStream_t s)
    {
        open = false;
        MIFX_CHECK(refuse_broken());
        ++groups;
        if (nccl)
        {
            queuedInGroup = 0;
            MIFX_NCCL_CHECK(rccl()->GroupEnd());
            return MIFX_OK;
        }
        // in-process group: post every send, then serve every receive, then order the stream behind the receivers' copies
        std::vector<std::shared_ptr<LocalPost>> mine;
        // on a failure below: withdraw the posts of this rank that nobody has taken yet (a later frame must not pair them with its receives).  A post a receiver has
        // already popped stays alive through the receiver's reference; its events are freed by ~LocalPost when the last holder lets go.
        struct Withdraw
        {
            LocalGroup* g; int rank; std::vector<std::shared_ptr<LocalPost>>* mine; bool armed = true;
            ~Withdraw()
            {
                if (!armed) return;
                {
                    std::lock_guard<std::mutex> lock(g->m);
                    for (auto& kv : g->box)
                        if (kv.first.first == rank)
                            for (auto it = kv.second.begin(); it != kv.second.end();)
                                it = std::find(mine->begin(), mine->end(), *it) != mine->end() ? kv.second.erase(it) : it + 1;
                }
                g->cv.notify_all();
                mine->clear();
            }
        } withdraw{group.get(), rank, &mine};
        std::vector<PendingOp> ops;
        ops.swap(pending); // (the queue is empty again whatever happens below)
        for (const PendingOp& op : ops)
            if (op.send)
            {
                auto post = std::make_shared<LocalPost>();
                post->src = op.ptr; post->bytes = op.bytes;
                MIFX_HIP_CHECK(hipEventCreateWithFlags(&post->ready, hipEventDisableTiming));
                MIFX_HIP_CHECK(hipEventCreateWithFlags(&post->copied, hipEventDisableTiming));
                MIFX_HIP_CHECK(hipEventRecord(post->ready, s));
                {
                    std::lock_guard<std::mutex> lock(group->m);
                    group->box[{rank, op.peer}].push_back(post);
                }
                group->cv.notify_all();
                mine.push_back(post);
            }
        for (const PendingOp& op : ops)
            if (!op.send)
            {
                std::shared_ptr<LocalPost> post;
                {
                    std::unique_lock<std::mutex> lock(group->m);
                    auto& q = group->box[{op.peer, rank}];
                    if (!group->cv.wait_for(lock, std::chrono::seconds(60), [&] { return !q.empty(); }))
                    {
                        set_error("in-process group: rank %d waited 60 s for a send of rank %d", rank, op.peer);
                        return MIFX_ERR_COMM;
                    }
                    post = q.front();
                    q.pop_front();
                }
                if (post->bytes != op.bytes)
                {
                    set_error("in-process group: rank %d expects %zu bytes from rank %d, which sends %zu", rank, op.bytes, op.peer, post->bytes);
                    return MIFX_ERR_COMM;
                }
                MIFX_HIP_CHECK(hipStreamWaitEvent(s, post->ready, 0));
                MIFX_HIP_CHECK(hipMemcpyAsync(op.ptr, post->src, op.bytes, hipMemcpyDeviceToDevice, s));
                MIFX_HIP_CHECK(hipEventRecord(post->copied, s));
                {
                    std::lock_guard<std::mutex> lock(group->m);
                    post->done = true;
                }
                group->cv.notify_all();
            }
        for (auto& post : mine)
        {
            std::unique_lock<std::mutex> lock(group->m);
            if (!group->cv.wait_for(lock, std::chrono::seconds(60), [&] { return post->done; }))
            {
                set_error("in-process group: rank %d waited 60 s for a receiver", rank);
                return MIFX_ERR_COMM;
            }
            lock.unlock();
            MIFX_HIP_CHECK(hipStreamWaitEvent(s, post->copied, 0));
            // (the receiver has recorded `copied` -- done is set after the record -- and this stream's wait on it is enqueued; the events go with the post)
        }
        withdraw.armed = false;
        return MIFX_OK;
    }
};

namespace
{
// begin() ... end() with every early return in between covered: the destructor closes a group that is still open (mifx_comm::abort_group)
struct GroupGuard
{
    mifx_comm* c;
    explicit GroupGuard(mifx_comm* comm) : c(comm) {}
    ~GroupGuard() { c->abort_group(); }
    GroupGuard(const GroupGuard&) = delete;
    GroupGuard& operator=(const GroupGuard&) = delete;
};
// rows [b, e) of a pitched plane: one contiguous slab
inline unsigned char* row_ptr(const Plane& p, int row) { return static_cast<unsigned char*>(p.data) + size_t(row) * p.pitch; }
inline size_t         row_bytes(const Plane& p, int b, int e) { return e > b ? size_t(e - b) * p.pitch : 0; }

// every rank's rows [b_r, e_r) of every plane of `planes` (planes of one height) go to every other rank: an all-gather of uneven row slabs as direct sends, one group
mifx_status allgather_rows(mifx_comm* c, std::initializer_list<const Plane*> planes, const std::vector<Rows>& rows, hipStream_t s)
{
    MIFX_CHECK(c->begin());
    GroupGuard guard(c); // (closes the group and drops the open time bracket on every early return)
    c->time_start(s);
    for (const Plane* plane : planes)
        for (int r = 0; r < c->world; ++r)
        {
            if (r == c->rank) continue;
            // (a rank may own no rows of a small plane: both sides skip the transfer, the ranges are known to all)
            if (!rows[c->rank].empty()) MIFX_CHECK(c->send(row_ptr(*plane, rows[c->rank].b), row_bytes(*plane, rows[c->rank].b, rows[c->rank].e), r, s));
            if (!rows[r].empty()) MIFX_CHECK(c->recv(row_ptr(*plane, rows[r].b), row_bytes(*plane, rows[r].b, rows[r].e), r, s));
        }
    MIFX_CHECK(c->end(s));
    c->time_stop(s);
    return MIFX_OK;
}
mifx_status allgather_rows(mifx_comm* c, const Plane& plane, const std::vector<Rows>& rows, hipStream_t s) { return allgather_rows(c, {&plane}, rows, s); }
} // namespace

extern "C" {

mifx_status mifx_comm_get_unique_id(uint8_t out_id[MIFX_COMM_ID_BYTES])
{
    MIFX_REQUIRE(out_id != nullptr, "mifx_comm_get_unique_id: null argument");
    static_assert(MIFX_COMM_ID_BYTES == sizeof(ncclUniqueId), "MIFX_COMM_ID_BYTES must be the size of ncclUniqueId");
    const Rccl* r = rccl();
    if (!r) return MIFX_ERR_COMM;
    ncclUniqueId id;
    MIFX_NCCL_CHECK(r->GetUniqueId(&id));
    std::memcpy(out_id, &id, sizeof(id));
    return MIFX_OK;
}

mifx_status mifx_comm_create(mifx_postfx* ctx, const uint8_t id[MIFX_COMM_ID_BYTES], int32_t rank, int32_t world, mifx_comm** out)
{
    MIFX_REQUIRE(ctx != nullptr && id != nullptr && out != nullptr && world >= 1 && rank >= 0 && rank < world, "mifx_comm_create: bad argument (rank %d of %d)", rank, world);
    *out = nullptr;
    const Rccl* r = rccl();
    if (!r) return MIFX_ERR_COMM;
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    std::unique_ptr<mifx_comm> c(new mifx_comm());
    c->device = ctx->device; c->rank = rank; c->world = world;
    MIFX_NCCL_CHECK(r->CommInitRank(&c->nccl, world, uid, rank));
    c->rccl_endpoint = true;
    MIFX_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    *out = c.release();
    return MIFX_OK;
}

mifx_status mifx_comm_create_local_group(mifx_postfx* ctx, int32_t world, mifx_comm** out_comms)
{
    MIFX_REQUIRE(ctx != nullptr && out_comms != nullptr && world >= 1 && world <= 64, "mifx_comm_create_local_group: bad argument");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    auto g = std::make_shared<LocalGroup>();
    g->world = world;
    for (int r = 0; r < world; ++r)
    {
        mifx_comm* c = new mifx_comm();
        c->device = ctx->device; c->rank = r; c->world = world; c->group = g;
        if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess)
        {
            delete c;
            for (int k = 0; k < r; ++k) mifx_comm_destroy(out_comms[k]);
            set_error("mifx_comm_create_local_group: hipStreamCreate failed");
            return MIFX_ERR_HIP;
        }
        out_comms[r] = c;
    }
    return MIFX_OK;
}

void mifx_comm_destroy(mifx_comm* c)
{
    if (!c) return;
    // chains that borrowed the communicator fall back to unsharded frames instead of keeping a dangling pointer
    for (mifx_chain* chain : std::vector<mifx_chain*>(c->users))
    {
        chain->comm = nullptr;
        chain->cuts.clear();
        (void)mifx_chain_set_row_band(chain, 0, 0, 0);
    }
    c->users.clear();
    (void)hipSetDevice(c->device);
    if (c->nccl && rccl()) (void)rccl()->CommDestroy(c->nccl);
    if (c->side) (void)hipStreamDestroy(c->side);
    while (!c->timed.empty()) c->drop_timed(c->timed.size() - 1);
    delete c;
}

// First contact with the transport before a frame depends on it: every rank sends every peer a slab of `bytes_per_peer` bytes whose words name (sender, receiver, word)
// and checks what it receives from every peer -- one group of the same begin / send / recv / end calls the frames use, on the context's stream.  A peer that never posts
// would leave the group's kernels spinning for ever: the host waits at most `timeout_ms` for the stream, then aborts the communicator (ncclCommAbort) and reports it.
// Collective: every rank of the communicator calls it.  MIFX_ERR_COMM with the transport's message in mifx_last_error() on any failure.
mifx_status mifx_comm_self_test(mifx_comm* c, mifx_postfx* ctx, uint32_t bytes_per_peer, uint32_t timeout_ms)
{
    MIFX_REQUIRE(c != nullptr && ctx != nullptr && bytes_per_peer >= 4u && bytes_per_peer % 4u == 0u, "mifx_comm_self_test: bad argument");
    if (c->world == 1) return MIFX_OK;
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t words = bytes_per_peer / 4u, total = words * size_t(c->world);
    std::vector<uint32_t> host(total), back(total, 0u);
    auto word = [](int from, int to, size_t i) { return uint32_t(0x9E3779B9u * uint32_t(from + 1) + 0x85EBCA6Bu * uint32_t(to + 1) + uint32_t(i)); };
    for (int q = 0; q < c->world; ++q)
        for (size_t i = 0; i < words; ++i) host[size_t(q) * words + i] = word(c->rank, q, i);
    DeviceScratch out, in;
    MIFX_CHECK(out.reserve(total * 4u));
    MIFX_CHECK(in.reserve(total * 4u));
    hipStream_t s = ctx->stream;
    MIFX_HIP_CHECK(hipMemcpyAsync(out.data, host.data(), total * 4u, hipMemcpyHostToDevice, s));
    MIFX_HIP_CHECK(hipMemsetAsync(in.data, 0, total * 4u, s));
    {
        MIFX_CHECK(c->begin());
        GroupGuard guard(c);
        for (int q = 0; q < c->world; ++q)
        {
            if (q == c->rank) continue;
            MIFX_CHECK(c->send(static_cast<const unsigned char*>(out.data) + size_t(q) * bytes_per_peer, bytes_per_peer, q, s));
            MIFX_CHECK(c->recv(static_cast<unsigned char*>(in.data) + size_t(q) * bytes_per_peer, bytes_per_peer, q, s));
        }
        MIFX_CHECK(c->end(s));
    }
    hipEvent_t done = nullptr;
    MIFX_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    struct EventGuard { hipEvent_t e; ~EventGuard() { (void)hipEventDestroy(e); } } eg{done};
    MIFX_HIP_CHECK(hipEventRecord(done, s));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;)
    {
        const hipError_t q = hipEventQuery(done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady)
        {
            set_error("mifx_comm_self_test: the exchange failed on the device: %s", hipGetErrorString(q));
            return MIFX_ERR_COMM;
        }
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > int64_t(timeout_ms))
        {
            // (the kernels of the group wait for a peer that does not answer: the abort is what ends them)
            c->broken = true;
            const bool canAbort = c->nccl && rccl() && rccl()->CommAbort;
            if (canAbort)
            {
                (void)rccl()->CommAbort(c->nccl);
                c->nccl = nullptr;
            }
            else if (c->nccl)
            {
                // (a transport without ncclCommAbort: the exchange's kernels keep waiting for the peer and keep the two slabs in use -- freeing them would wait for the
                //  device for ever, so they are left allocated; the endpoint is `broken` and refuses every later exchange.  The limit of this time-out: ncclGroupEnd above
                //  runs on the host before the wait starts, and a transport that blocks THERE is not covered)
                out.abandon();
                in.abandon();
            }
            set_error("mifx_comm_self_test: rank %d of %d: no answer from the peers within %u ms (%s)", c->rank, c->world, timeout_ms,
                      canAbort ? "communicator aborted" : "the transport has no ncclCommAbort: the endpoint is unusable and its two test slabs stay allocated");
            return MIFX_ERR_COMM;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    MIFX_HIP_CHECK(hipMemcpy(back.data(), in.data, total * 4u, hipMemcpyDeviceToHost));
    for (int q = 0; q < c->world; ++q)
    {
        if (q == c->rank) continue;
        for (size_t i = 0; i < words; ++i)
            if (back[size_t(q) * words + i] != word(q, c->rank, i))
            {
                set_error("mifx_comm_self_test: rank %d received word %zu of rank %d's slab as 0x%08x, expected 0x%08x", c->rank, i, q, back[size_t(q) * words + i], word(q, c->rank, i));
                return MIFX_ERR_COMM;
            }
    }
    return MIFX_OK;
}

mifx_status mifx_comm_get_info(const mifx_comm* c, int32_t* out_rank, int32_t* out_world, int32_t* out_is_rccl)
{
    MIFX_REQUIRE(c != nullptr, "mifx_comm_get_info: null argument");
    if (out_rank) *out_rank = c->rank;
    if (out_world) *out_world = c->world;
    if (out_is_rccl) *out_is_rccl = c->rccl_endpoint ? 1 : 0;
    return MIFX_OK;
}

// Timing of the exchange groups of the frames that follow (off by default: two event records per group).  Switching it off keeps what was recorded for mifx_comm_get_stats.
mifx_status mifx_comm_set_timing(mifx_comm* c, int32_t enable)
{
    MIFX_REQUIRE(c != nullptr, "mifx_comm_set_timing: null argument");
    c->timing = enable != 0;
    return MIFX_OK;
}

// What this endpoint is and what it has moved.  ranks_in_communicator is RCCL's own answer (ncclCommCount) -- the check that the library's communicator really spans the
// ranks the caller started, not `world` echoed back; the in-process group reports its size, a transport without the entry point -1.  The exchange durations recorded since
// mifx_comm_set_timing(comm, 1) are read here (the call waits for the events of those groups, i.e. for the frames that issued them) and then forgotten.
mifx_status mifx_comm_get_stats(mifx_comm* c, mifx_comm_stats* out)
{
    MIFX_REQUIRE(c != nullptr && out != nullptr, "mifx_comm_get_stats: null argument");
    *out = mifx_comm_stats{};
    out->rank = c->rank; out->world = c->world; out->is_rccl = c->rccl_endpoint ? 1 : 0;
    out->ranks_in_communicator = c->group ? c->group->world : -1;
    if (c->nccl && rccl() && rccl()->CommCount)
    {
        int n = -1;
        if (rccl()->CommCount(c->nccl, &n) == ncclSuccess) out->ranks_in_communicator = n;
    }
    out->groups = c->groups; out->bytes_sent = c->bytesSent; out->bytes_received = c->bytesReceived;
    (void)hipSetDevice(c->device);
    while (!c->timed.empty())
    {
        float ms = 0.0f;
        if (hipEventSynchronize(c->timed.back().stop) == hipSuccess && hipEventElapsedTime(&ms, c->timed.back().start, c->timed.back().stop) == hipSuccess)
        {
            ++out->timed_groups;
            out->exchange_ms_total += ms;
            out->exchange_ms_max = ms > out->exchange_ms_max ? ms : out->exchange_ms_max;
        }
        c->drop_timed(c->timed.size() - 1);
    }
    return MIFX_OK;
}

// The bands of all ranks: row_cuts[0] = 0 < row_cuts[1] < ... < row_cuts[world] = frame height; this rank owns [row_cuts[rank], row_cuts[rank + 1]).
mifx_status mifx_chain_set_sharding(mifx_chain* chain, mifx_comm* comm, const int32_t* row_cuts, int32_t max_motion_rows)
{
    MIFX_REQUIRE(chain != nullptr, "mifx_chain_set_sharding: null chain");
    chain->join_halos(); // (an exchange of the previous frame still in flight belongs to the old bands)
    if (comm == nullptr) // off
    {
        mifx::chain_detach_comm(chain);
        return mifx_chain_set_row_band(chain, 0, 0, 0);
    }
    MIFX_REQUIRE(row_cuts != nullptr && max_motion_rows >= 0, "mifx_chain_set_sharding: bad argument");
    MIFX_REQUIRE(comm->device == chain->ctx->device, "mifx_chain_set_sharding: the communicator lives on device %d, the chain on %d", comm->device, chain->ctx->device);
    MIFX_REQUIRE(row_cuts[0] == 0, "mifx_chain_set_sharding: row_cuts[0] must be 0");
    for (int r = 0; r < comm->world; ++r) MIFX_REQUIRE(row_cuts[r + 1] > row_cuts[r], "mifx_chain_set_sharding: row_cuts must increase (band %d is empty)", r);
    // the band first (since round 3 every option of the chain runs inside the row-band phases: only bad rows are refused), then the communicator
    if (comm->world == 1) MIFX_CHECK(mifx_chain_set_row_band(chain, 0, 0, 0)); // one rank: the whole frame, no phases
    else MIFX_CHECK(mifx_chain_set_row_band(chain, row_cuts[comm->rank], row_cuts[comm->rank + 1], max_motion_rows));
    mifx::chain_detach_comm(chain);
    chain->comm = comm;
    chain->cuts.assign(row_cuts, row_cuts + comm->world + 1);
    comm->users.push_back(chain);
    return MIFX_OK;
}

// the chain stops borrowing its communicator (mifx_chain_set_sharding(NULL), a new communicator, ~mifx_chain)
extern "C++" void mifx::chain_detach_comm(mifx_chain* chain)
{
    if (chain->comm)
    {
        auto& u = chain->comm->users;
        u.erase(std::remove(u.begin(), u.end(), chain), u.end());
    }
    chain->comm = nullptr;
    chain->cuts.clear();
}

// One frame of a rank's band: the phases of mifx_chain_execute_phase with the exchanges between them -- or, comm == nullptr (mifx_chain_execute_band), without them.
//
// Two lanes across frames (round 5; chain->overlap >= 2 and asynchronous halos, the input contract of mifx_chain_set_overlap 2): phases 0 - 2 -- shade, prep, SSAO, SSR,
// composite, TAA, Bloom's fine levels, the Bloom gather -- run on the chain's side stream L, phase 3 -- Bloom's coarse levels and the final pass, ~15 launches of a few
// microseconds each that leave the GPU idle -- on the context's stream M behind them.  The next frame's L does not wait for M until its own phase 2 (which overwrites the
// Bloom levels and the depth-of-field output phase 3 reads), so this frame's Bloom tail runs beside the next frame's shade and SSAO -- what the unsharded chain's lanes do
// for the whole frame, for a band whose fixed per-rank work weighs eight times as much.  When the call returns, M is ordered behind everything of the frame.
// (chain->overlap >= 3: a third lane for the PostFX prep and SSAO, below.)
static mifx_status execute_sharded_impl(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* out_ldr, mifx_comm* c)
{
    const int H = int(f->frame.Height);
    const int world = c ? c->world : 1, rank = c ? c->rank : 0;
    mifx_postfx* ctx = chain->ctx;
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    const hipStream_t M = ctx->stream;

    // Everything that can be refused is checked before the first kernel and before any group is opened: what every rank owns and needs follows from the cuts and the
    // per-frame attributes alone (the resources are prepared first: the Bloom plan reads the level sizes).
    MIFX_CHECK(mifx::chain_prepare_resources(chain, f));
    std::vector<Rows> bands(world);
    std::vector<mifx_shard_info> info(world);
    int halos[3] = {0, 0, 0}; // TAA, SSR, SSAO: both neighbours of an edge move the same number of rows = the largest need of any rank
    // Round 6: Bloom's level 0 with halos (mifx_bloom::halo_level0) -- the row windows of everything in front of Bloom follow from it, so the request is in place before
    // the needs of the ranks are derived; it ends with the call (HaloLevel0 below).
    struct HaloLevel0
    {
        mifx_bloom* b;
        ~HaloLevel0() { b->halo_level0 = false; b->after_level0 = nullptr; }
    } haloLevel0{chain->bloom};
    chain->bloom->halo_level0 = mifx::shard_bloom_halo_enabled();
    std::vector<mifx_bloom::Plan> plans(world);
    if (c)
    {
        for (int r = 0; r < world; ++r) bands[r] = Rows{chain->cuts[r], chain->cuts[r + 1]};
        for (int r = 0; r < world; ++r) info[r] = chain_shard_info(chain, f, bands[r]);
        for (int r = 0; r < world; ++r) plans[r] = mifx::chain_bloom_plan(chain, f, bands[r]);
        for (int r = 0; r < world; ++r)
        {
            MIFX_REQUIRE(info[r].gather_level == info[rank].gather_level, "mifx_chain_execute_sharded: ranks disagree on the Bloom gather level");
            halos[0] = std::max(halos[0], int(info[r].halo_taa)); halos[1] = std::max(halos[1], int(info[r].halo_ssr)); halos[2] = std::max(halos[2], int(info[r].halo_ssao));
        }
    }
    const mifx_shard_info me = c ? info[rank] : mifx_shard_info{};
    // (a halo taller than a neighbour's band reaches into the band beyond it: the exchange below sends every rank the rows of its ghost zones from whichever
    //  ranks own them -- "multi-hop" in one step, since all ranks are peers over xGMI)

    const bool async = chain->async_halos;
    const bool lanes = chain->overlap >= 2 && async && !chain->profiling;
    // Three lanes (chain->overlap >= 3): prep + SSAO (phase 1) on a lane A of their own -- they read the G-buffer and the PostFX planes only -- beside the shade (phase 0) and
    // SSR (the first half of phase 2) on L.  L waits for the prep (SSR's temporal pass reads its planes) before phase 2 and for the end of SSAO in front of the composite
    // (mifx_chain::sig_after_prep / wait_before_composite); the next frame's A waits for this frame's phase 2 on L, the last reader of what the prep and SSAO overwrite.
    const bool lanes3 = lanes && chain->overlap >= 3;
    // ... and a stream H for SSR's depth hierarchy (mifx_ssr::hiz_stream): whole-frame streaming work on every rank that depends on the depth buffer alone -- beside the
    // shade instead of between it and the march.  Its last reader is the previous frame's march (phase 2 on L); the march of this frame waits for it inside mifx_ssr_execute.
    hipStream_t L = M; // the stream of phases 0 - 2
    hipStream_t A = M; // the stream of phase 1
    hipStream_t Hs = nullptr;
    if (lanes)
    {
        MIFX_CHECK(mifx::chain_make_lanes(chain, true));
        L = chain->side;
        A = lanes3 ? chain->lane_x : L;
        if (lanes3 && !chain->lane_h)
        {
            MIFX_HIP_CHECK(hipStreamCreateWithFlags(&chain->lane_h, hipStreamNonBlocking));
            for (hipEvent_t* e : {&chain->evHiz, &chain->evJoinH}) MIFX_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
        Hs = lanes3 ? chain->lane_h : nullptr;
        if (!mifx::chain_lanes_continue(chain)) // first frame, or the library queued work on M since the last one (resets, imports, re-allocations): the lanes behind M once
        {
            MIFX_HIP_CHECK(hipEventRecord(chain->evFork, M));
            MIFX_HIP_CHECK(hipStreamWaitEvent(L, chain->evFork, 0));
            if (lanes3) MIFX_HIP_CHECK(hipStreamWaitEvent(A, chain->evFork, 0));
            if (Hs) MIFX_HIP_CHECK(hipStreamWaitEvent(Hs, chain->evFork, 0));
        }
        else if (lanes3)
        {
            MIFX_HIP_CHECK(hipStreamWaitEvent(A, chain->evPrepConsumed, 0)); // (recorded on L behind the previous frame's phase 2)
            if (Hs) MIFX_HIP_CHECK(hipStreamWaitEvent(Hs, chain->evPrepConsumed, 0));
        }
    }
    struct Restore // whatever happens, the context's stream is M again and ends behind the lanes, and no event request is left on the chain
    {
        mifx_chain* ch;
        hipStream_t m, l, a, h;
        bool        joined = false;
        ~Restore()
        {
            ch->ctx->stream = m;
            ch->sig_after_prep = ch->wait_before_composite = nullptr;
            if (ch->ssr) { ch->ssr->hiz_stream = nullptr; ch->ssr->hiz_done = nullptr; }
            if (joined) return;
            if (l != m && hipEventRecord(ch->evJoinS, l) == hipSuccess) (void)hipStreamWaitEvent(m, ch->evJoinS, 0);
            if (a != l && hipEventRecord(ch->evJoinX, a) == hipSuccess) (void)hipStreamWaitEvent(m, ch->evJoinX, 0);
            if (h != nullptr && hipEventRecord(ch->evJoinH, h) == hipSuccess) (void)hipStreamWaitEvent(m, ch->evJoinH, 0);
        }
    } restore{chain, M, L, A, Hs};

    // History halos for the next frame: every rank receives the rows of its two ghost zones from whichever ranks own them.  Both sides of a transfer derive its rows from
    // the cuts and the halo sizes = the largest need of any rank, recomputed every frame (the needs follow the per-frame attributes: SSAO reconstruction radius, Bloom radius).
    const uint32_t ci = f->frame.Index & 1u;
    struct HistoryPlane { const Plane* p; int halo; };
    auto exchange_halos = [&](std::initializer_list<HistoryPlane> planes, hipStream_t s) -> mifx_status {
        if (!c) return MIFX_OK;
        MIFX_CHECK(c->begin());
        GroupGuard guard(c);
        c->time_start(s);
        auto meet = [](Rows a, Rows b) { return Rows{a.b > b.b ? a.b : b.b, a.e < b.e ? a.e : b.e}; };
        for (const HistoryPlane& hp : planes)
        {
            const int  halo = hp.halo;
            const Rows mine = bands[rank];
            // the ghost zone of rank r on the side of rank q: the `halo` rows above its band when q lies above it, below otherwise
            auto ghost = [&](int r, int q) { return rows_clip(q < r ? Rows{bands[r].b - halo, bands[r].b} : Rows{bands[r].e, bands[r].e + halo}, H); };
            for (int q = 0; q < world; ++q)
            {
                if (q == rank) continue;
                const Rows out = meet(mine, ghost(q, rank)), in = meet(bands[q], ghost(rank, q)); // both follow from the cuts: the peer computes the same two ranges
                if (!out.empty()) MIFX_CHECK(c->send(row_ptr(*hp.p, out.b), row_bytes(*hp.p, out.b, out.e), q, s));
                if (!in.empty()) MIFX_CHECK(c->recv(row_ptr(*hp.p, in.b), row_bytes(*hp.p, in.b, in.e), q, s));
            }
        }
        MIFX_CHECK(c->end(s));
        c->time_stop(s);
        return MIFX_OK;
    };
    // Asynchronous halos (mifx_chain::async_halos): a plane's halo is sent on `halo_stream` as soon as the pass that writes it is done, and the stream of the phases waits
    // for it where the next frame first reads that plane.  Every rank issues its groups in the same order (SSAO halos, Bloom gather, TAA + SSR halos), as the transports require.
    if (c && async && chain->halo_stream == nullptr)
    {
        MIFX_HIP_CHECK(hipStreamCreateWithFlags(&chain->halo_stream, hipStreamNonBlocking));
        for (hipEvent_t* e : {&chain->evAfterP1, &chain->evAfterP2, &chain->evHaloSsao, &chain->evHaloRest}) MIFX_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    auto halos_after = [&](hipStream_t producer, hipEvent_t produced, hipEvent_t exchanged, bool& pending, std::initializer_list<HistoryPlane> planes) -> mifx_status {
        if (!c) return MIFX_OK;
        MIFX_HIP_CHECK(hipEventRecord(produced, producer));
        MIFX_HIP_CHECK(hipStreamWaitEvent(chain->halo_stream, produced, 0));
        MIFX_CHECK(exchange_halos(planes, chain->halo_stream));
        MIFX_HIP_CHECK(hipEventRecord(exchanged, chain->halo_stream));
        pending = true;
        // (work queued on the context's stream outside this function is ordered behind the exchange: mifx_postfx::queued_outside_execute; the event is re-recorded every frame)
        if (std::find(ctx->pending_joins.begin(), ctx->pending_joins.end(), exchanged) == ctx->pending_joins.end()) ctx->pending_joins.push_back(exchanged);
        return MIFX_OK;
    };

    // phases 0 and 1: shade, prep, SSAO.  (Until round 3 the band rows of the shaded radiance were all-gathered here -- 465 MB per GPU and frame at 8K / 8 ranks; the
    // ray march now records where it hit and phase 2 loads the colour there or re-shades it: api_chain.cpp.)  mifx_chain_execute_phase joins a pending halo exchange of
    // the previous frame where the phase first reads the plane.
    ctx->stream = L;
    MIFX_CHECK(mifx_chain_execute_phase(chain, f, out_ldr, 0));
    ctx->stream = A;
    if (lanes3) chain->sig_after_prep = chain->evPrep;
    // Round 6: the last level of SSAO's depth pyramid, which A3's far taps read anywhere, is reduced by the rank that owns its rows and all-gathered (two planes of
    // (H / 16) x (W / 16) texels: 1 MB per frame at 7680x4320) instead of reduced whole on every rank: the row of the last level that holds frame row y belongs to the
    // rank whose band holds its first frame row.  Every rank takes the same decision (it follows from the frame and the SSAO flags alone); MIFX_SHARD_GATHER_SSAO_LEVEL=0
    // keeps round 5's whole pyramids.  mifx_chain_execute_band (c == nullptr) reduces what a rank of the sharded frame reduces and leaves the other rows stale.
    {
        static const bool gatherOn = []() { const char* e = std::getenv("MIFX_SHARD_GATHER_SSAO_LEVEL"); return e == nullptr || std::atoi(e) != 0; }();
        constexpr int kLast = mifx_ssao::kMips - 1;
        const uint32_t W = f->frame.Width;
        const bool can = gatherOn && (chain->ssao_flags & MIFX_SSAO_FEATURE_FLAG_HALF_RESOLUTION) == 0 && !chain->ssao->depth16 &&
                         pyramid_fusable_levels(int(W), H, kLast) == kLast && !chain->band.empty();
        if (can)
        {
            auto own = [&](Rows band) { return Rows{(band.b + (1 << kLast) - 1) >> kLast, (band.e + (1 << kLast) - 1) >> kLast}; };
            chain->ssao->gather_last_level = true;
            chain->ssao->own_last_level    = own(c ? bands[rank] : chain->band);
            if (c)
            {
                std::vector<Rows> owned(world);
                for (int r = 0; r < world; ++r) owned[r] = own(bands[r]);
                chain->ssao->after_prefilter = [c, owned](const Plane& d, const Plane& z, hipStream_t s) -> mifx_status { return allgather_rows(c, {&d, &z}, owned, s); };
            }
        }
    }
    const mifx_status p1 = mifx_chain_execute_phase(chain, f, out_ldr, 1);
    chain->ssao->own_last_level    = Rows{0, 0}; // (per-frame requests: gone whatever the phase returned)
    chain->ssao->gather_last_level = false;
    chain->ssao->after_prefilter   = nullptr;
    MIFX_CHECK(p1);
    chain->sig_after_prep = nullptr;
    if (lanes3) MIFX_HIP_CHECK(hipEventRecord(chain->evSsao, A));
    if (async) MIFX_CHECK(halos_after(A, chain->evAfterP1, chain->evHaloSsao, chain->halo_ssao_pending, {{&chain->ssao->history_ao[ci], halos[2]}, {&chain->ssao->history_len[ci], halos[2]}}));
    ctx->stream = L;

    // phase 2 (behind the previous frame's phase 3, whose Bloom levels and depth-of-field output it overwrites), then the Bloom level every rank needs whole: what each
    // rank owns follows from its band
    if (lanes) MIFX_HIP_CHECK(hipStreamWaitEvent(L, chain->evBloomDone, 0)); // (never recorded = no wait)
    if (lanes3)
    {
        MIFX_HIP_CHECK(hipStreamWaitEvent(L, chain->evPrep, 0));
        chain->wait_before_composite = chain->evSsao;
        const char* hizLane = std::getenv("MIFX_SHARD_HIZ_LANE"); // (0: the hierarchy stays on L between the shade and the march -- A/B runs)
        if (Hs && (hizLane == nullptr || std::atoi(hizLane) != 0))
        {
            chain->ssr->hiz_stream = Hs; // (a per-frame request: mifx_ssr_execute takes and clears it)
            chain->ssr->hiz_done   = chain->evHiz;
        }
    }
    // Bloom's level-0 halos: between the prefilter and the first down-sampling (inside phase 2, on this lane) every rank sends the rows of level 0 it owns that another rank
    // reads but does not produce -- the rows beside the band edges -- and receives its own; who reads and who produces what follows from the plans of all ranks.
    if (c && chain->bloom->halo_level0)
    {
        bool any = false;
        for (int r = 0; r < world; ++r) any = any || (plans[r].G >= 0 && !(plans[r].compute0.b == plans[r].down[0].b && plans[r].compute0.e == plans[r].down[0].e));
        if (any)
            chain->bloom->after_level0 = [c, rank, world, &plans](const Plane& level0, hipStream_t s) -> mifx_status {
                MIFX_CHECK(c->begin());
                GroupGuard guard(c);
                c->time_start(s);
                auto meet = [](Rows a, Rows b) { return Rows{a.b > b.b ? a.b : b.b, a.e < b.e ? a.e : b.e}; };
                // the two parts of what rank q reads and does not produce: below and above the rows it prefilters itself
                auto parts = [&](int q, Rows out[2]) { out[0] = Rows{plans[q].down[0].b, plans[q].compute0.b}; out[1] = Rows{plans[q].compute0.e, plans[q].down[0].e}; };
                for (int q = 0; q < world; ++q)
                {
                    if (q == rank) continue;
                    Rows theirs[2], mine[2];
                    parts(q, theirs);
                    parts(rank, mine);
                    for (int k = 0; k < 2; ++k)
                    {
                        const Rows out = meet(theirs[k], plans[rank].own0), in = meet(mine[k], plans[q].own0);
                        if (!out.empty()) MIFX_CHECK(c->send(row_ptr(level0, out.b), row_bytes(level0, out.b, out.e), q, s));
                        if (!in.empty()) MIFX_CHECK(c->recv(row_ptr(level0, in.b), row_bytes(level0, in.b, in.e), q, s));
                    }
                }
                MIFX_CHECK(c->end(s));
                c->time_stop(s);
                return MIFX_OK;
            };
    }
    MIFX_CHECK(mifx_chain_execute_phase(chain, f, out_ldr, 2));
    chain->bloom->after_level0 = nullptr;
    chain->wait_before_composite = nullptr;
    if (c && me.gather_level >= 0)
    {
        std::vector<Rows> own(world);
        for (int r = 0; r < world; ++r) own[r] = Rows{info[r].own_begin, info[r].own_end};
        MIFX_CHECK(allgather_rows(c, *chain->bloom->down[me.gather_level], own, L));
    }
    if (async)
        MIFX_CHECK(halos_after(L, chain->evAfterP2, chain->evHaloRest, chain->halo_rest_pending,
                               {{&chain->taa->accum[ci], halos[0]}, {&chain->ssr->hist_radiance[ci], halos[1]}, {&chain->ssr->hist_variance[ci], halos[1]}}));
    ctx->stream = M;
    if (lanes)
    {
        MIFX_HIP_CHECK(hipEventRecord(chain->evPrepConsumed, L));
        MIFX_HIP_CHECK(hipStreamWaitEvent(M, chain->evPrepConsumed, 0));
        restore.joined = true;
    }
    MIFX_CHECK(mifx_chain_execute_phase(chain, f, out_ldr, 3));
    if (chain->auto_exposure) // the low-resolution luminance rows of every band, then the reduction and the tone map
    {
        if (c)
        {
            std::vector<Rows> lum(world);
            for (int r = 0; r < world; ++r) lum[r] = Rows{info[r].ae_begin, info[r].ae_end};
            MIFX_CHECK(allgather_rows(c, chain->auto_exposure->low_res, lum, M));
        }
        MIFX_CHECK(mifx_chain_execute_phase(chain, f, out_ldr, 4));
    }
    if (lanes)
    {
        MIFX_HIP_CHECK(hipEventRecord(chain->evBloomDone, M));
        chain->seen_epoch    = ctx->stream_epoch;
        chain->prep_consumed = true;
    }
    if (async) return MIFX_OK;
    return exchange_halos({{&chain->taa->accum[ci], halos[0]}, {&chain->ssr->hist_radiance[ci], halos[1]}, {&chain->ssr->hist_variance[ci], halos[1]},
                           {&chain->ssao->history_ao[ci], halos[2]}, {&chain->ssao->history_len[ci], halos[2]}}, M);
}

mifx_status mifx_chain_execute_sharded(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* out_ldr)
{
    MIFX_REQUIRE(chain != nullptr && f != nullptr && out_ldr != nullptr, "mifx_chain_execute_sharded: null argument");
    if (chain->comm == nullptr)
    {
        set_error("mifx_chain_execute_sharded: no communicator (mifx_chain_set_sharding)");
        return MIFX_ERR_INVALID_OP;
    }
    mifx_comm* c = chain->comm;
    if (c->world == 1) return mifx_chain_execute(chain, f, out_ldr);
    MIFX_REQUIRE(chain->cuts.back() == int(f->frame.Height), "mifx_chain_execute_sharded: the bands cover %d rows, the frame has %d", chain->cuts.back(), int(f->frame.Height));
    return execute_sharded_impl(chain, f, out_ldr, c);
}

// The compute side of mifx_chain_execute_sharded for ONE rank, exchanges left out (the ghost rows then hold stale values, which does not change the work): the band of
// mifx_chain_set_row_band through the same phases, lanes included.  What tools/shard_cost.py and TiledChain.time_own_band time; the frames it produces are not images.
mifx_status mifx_chain_execute_band(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* out_ldr)
{
    MIFX_REQUIRE(chain != nullptr && f != nullptr && out_ldr != nullptr, "mifx_chain_execute_band: null argument");
    MIFX_REQUIRE(!chain->band.empty() && chain->comm == nullptr, "mifx_chain_execute_band: set a row band (mifx_chain_set_row_band) on a chain without a communicator");
    return execute_sharded_impl(chain, f, out_ldr, nullptr);
}

} // extern "C"
