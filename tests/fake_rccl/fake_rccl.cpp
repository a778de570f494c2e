// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl.so that lets N PROCESSES SHARING ONE GPU run the RCCL branch of csrc/api_comm.cpp.
//
// RCCL refuses two ranks on one device, and the GPU box of this project has one GPU: until an 8-GPU node runs the driver's scaling bench, the code path
// `ncclCommInitRank -> ncclGroupStart -> ncclSend / ncclRecv ... -> ncclGroupEnd` of mifx_chain_execute_sharded would never execute with more than one rank.  This
// library implements exactly the entry points libmifx.so resolves with dlsym (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclSend, ncclRecv, ncclGroupStart,
// ncclGroupEnd, ncclGetErrorString), with RCCL's semantics as far as the caller can observe them: point-to-point operations of one (source, destination) pair match in
// order, operations inside a group are issued together at ncclGroupEnd, data is complete on the stream after the call.  Transport: hipIpcMemHandle of the sender's
// allocation through a POSIX shared-memory mailbox named by the unique id; the receiver copies device to device.  It is synchronous (the group end waits for the
// stream, the copies and the acknowledgements), which RCCL permits and a test does not mind.  Loaded through MIFX_RCCL_PATH; never linked into anything.
//
//   g++ -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.cpp -o librccl_fake.so -L/opt/rocm/lib -lamdhip64 -lrt
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace
{
constexpr int kMaxRanks = 16, kRing = 64;
struct Post
{
    hipIpcMemHandle_t handle;
    unsigned long long offset, bytes;
};
struct Pair // posts of one (source, destination) pair, in order
{
    std::atomic<unsigned> posted, taken;
    Post                  ring[kRing];
};
struct Shared
{
    std::atomic<int> ready, arrived, departed;
    int              nranks;
    Pair             pair[kMaxRanks][kMaxRanks];
};
struct Op
{
    bool        send;
    void*       ptr;
    size_t      bytes;
    int         peer;
    hipStream_t stream;
};
} // namespace

struct ncclComm
{
    int         rank = 0, nranks = 1;
    Shared*     sh   = nullptr;
    std::string name;
    std::vector<Op>                     queued;
    std::map<std::string, void*>        opened; // ipc handle bytes -> mapped base
};

namespace
{
thread_local int                    g_depth = 0;
thread_local std::vector<ncclComm*> g_touched;
// how long an operation waits for its peer: 60 s, or MIFX_FAKE_RCCL_TIMEOUT seconds (the error-path tests do not wait a minute for a rank that never posts)
double timeout_s()
{
    static const double t = [] { const char* e = std::getenv("MIFX_FAKE_RCCL_TIMEOUT"); const double v = e ? std::atof(e) : 0.0; return v > 0.0 ? v : 60.0; }();
    return t;
}
#define kTimeout timeout_s()

template <class F> bool wait_until(F&& f)
{
    const auto t0 = std::chrono::steady_clock::now();
    while (!f())
    {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeout) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return true;
}

ncclResult_t flush(ncclComm* c)
{
    std::vector<Op> ops;
    ops.swap(c->queued);
    for (const Op& op : ops)
        if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError; // the rows a send reads are complete; a receive may overwrite its target
    for (const Op& op : ops)
        if (op.send)
        {
            void*  base = nullptr;
            size_t size = 0;
            if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, op.ptr) != hipSuccess) return ncclUnhandledCudaError;
            Pair& p = c->sh->pair[c->rank][op.peer];
            if (!wait_until([&] { return p.posted.load() - p.taken.load() < unsigned(kRing); })) return ncclSystemError;
            Post& post = p.ring[p.posted.load() % kRing];
            if (hipIpcGetMemHandle(&post.handle, base) != hipSuccess) return ncclUnhandledCudaError;
            post.offset = static_cast<unsigned long long>(static_cast<char*>(op.ptr) - static_cast<char*>(base));
            post.bytes  = op.bytes;
            p.posted.fetch_add(1, std::memory_order_release);
        }
    for (const Op& op : ops)
        if (!op.send)
        {
            Pair& p = c->sh->pair[op.peer][c->rank];
            if (!wait_until([&] { return p.posted.load(std::memory_order_acquire) > p.taken.load(); }))
            {
                std::fprintf(stderr, "fake rccl: rank %d waited %g s for a send of rank %d\n", c->rank, kTimeout, op.peer);
                return ncclSystemError;
            }
            const Post post = p.ring[p.taken.load() % kRing];
            if (post.bytes != op.bytes)
            {
                std::fprintf(stderr, "fake rccl: rank %d expects %zu bytes from rank %d, which sends %llu\n", c->rank, op.bytes, op.peer, post.bytes);
                return ncclInvalidArgument;
            }
            const std::string key(reinterpret_cast<const char*>(&post.handle), sizeof(post.handle));
            auto              it = c->opened.find(key);
            if (it == c->opened.end())
            {
                void* mapped = nullptr;
                if (hipIpcOpenMemHandle(&mapped, post.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess)
                {
                    std::fprintf(stderr, "fake rccl: hipIpcOpenMemHandle failed: %s\n", hipGetErrorString(hipGetLastError()));
                    return ncclUnhandledCudaError;
                }
                it = c->opened.emplace(key, mapped).first;
            }
            // On the receive's own stream, and waited for: a device-to-device hipMemcpy on the null stream may return before the copy has landed, and the streams of the
            // library are non-blocking ones that do not wait for the null stream -- a consumer launched right behind the exchange (A3 behind the gather of the depth
            // pyramid's last level, round 6) then read the rows before they arrived.  "Data is complete on the stream after the call" is what the real library guarantees.
            if (hipMemcpyAsync(op.ptr, static_cast<char*>(it->second) + post.offset, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) return ncclUnhandledCudaError;
            if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
            p.taken.fetch_add(1, std::memory_order_release);
        }
    // a sender may reuse its rows once the receiver has copied them
    for (const Op& op : ops)
        if (op.send)
        {
            Pair& p = c->sh->pair[c->rank][op.peer];
            if (!wait_until([&] { return p.taken.load(std::memory_order_acquire) == p.posted.load(); })) return ncclSystemError;
        }
    return ncclSuccess;
}
ncclResult_t enqueue(ncclComm* c, Op op)
{
    if (c == nullptr || op.peer < 0 || op.peer >= c->nranks || op.peer == c->rank) return ncclInvalidArgument;
    c->queued.push_back(op);
    if (g_depth == 0) return flush(c);
    bool known = false;
    for (ncclComm* t : g_touched) known = known || t == c;
    if (!known) g_touched.push_back(c);
    return ncclSuccess;
}
size_t type_size(ncclDataType_t t)
{
    switch (t)
    {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}
} // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/mifx_fake_rccl_%d_%lld", int(getpid()), static_cast<long long>(std::chrono::steady_clock::now().time_since_epoch().count() % 1000000007LL));
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (comm == nullptr || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const std::string name(id.internal, strnlen(id.internal, sizeof(id.internal)));
    int  fd      = shm_open(name.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
    bool creator = fd >= 0;
    if (!creator) fd = shm_open(name.c_str(), O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (creator && ftruncate(fd, sizeof(Shared)) != 0) return ncclSystemError;
    if (!creator)
    {
        // the creator sizes the segment before anybody maps it
        struct stat_like { };
        if (!wait_until([&] { return lseek(fd, 0, SEEK_END) >= off_t(sizeof(Shared)); })) return ncclSystemError;
    }
    void* m = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    Shared* sh = static_cast<Shared*>(m);
    if (creator)
    {
        sh->nranks = nranks; // (a fresh segment is zero-filled: every counter starts at 0)
        sh->ready.store(1, std::memory_order_release);
    }
    else if (!wait_until([&] { return sh->ready.load(std::memory_order_acquire) == 1; })) return ncclSystemError;
    if (sh->nranks != nranks) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->rank = rank; c->nranks = nranks; c->sh = sh; c->name = name;
    sh->arrived.fetch_add(1);
    if (!wait_until([&] { return sh->arrived.load() >= nranks; }))
    {
        std::fprintf(stderr, "fake rccl: rank %d: only %d of %d ranks arrived\n", rank, sh->arrived.load(), nranks);
        delete c;
        return ncclSystemError;
    }
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (c == nullptr) return ncclSuccess;
    for (auto& kv : c->opened) (void)hipIpcCloseMemHandle(kv.second);
    const int left = c->sh->departed.fetch_add(1) + 1;
    if (left >= c->nranks) shm_unlink(c->name.c_str());
    munmap(c->sh, sizeof(Shared));
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int* count) // (the ranks that arrived at ncclCommInitRank: what mifx_comm_get_stats reports as ranks_in_communicator)
{
    if (c == nullptr || count == nullptr) return ncclInvalidArgument;
    *count = c->sh->arrived.load();
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t c) // (nothing runs on the device on this transport's behalf: giving up = leaving)
{
    if (c != nullptr) c->queued.clear();
    return ncclCommDestroy(c);
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t s)
{
    return enqueue(c, Op{true, const_cast<void*>(buf), count * type_size(type), peer, s});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t s) { return enqueue(c, Op{false, buf, count * type_size(type), peer, s}); }
ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<ncclComm*> comms;
    comms.swap(g_touched);
    ncclResult_t r = ncclSuccess;
    for (ncclComm* c : comms)
    {
        const ncclResult_t q = flush(c);
        if (q != ncclSuccess) r = q;
    }
    return r;
}
const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r)
    {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake rccl)";
        case ncclSystemError: return "system error / timeout (fake rccl)";
        case ncclInvalidArgument: return "invalid argument (fake rccl)";
        case ncclInvalidUsage: return "invalid usage (fake rccl)";
        default: return "error (fake rccl)";
    }
}
} // extern "C"
