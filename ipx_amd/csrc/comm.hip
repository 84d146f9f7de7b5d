// Multi-GPU exchange step: rows of AI are partitioned over the ranks of one node, each
// rank computes its partial t_g = Ws .* (A_g' y_g) and one all-reduce over xGMI sums the
// n-vector before the second pass.  Two transports:
//   * RCCL (default): ncclAllReduce / ncclAllGather on the context's stream.  RCCL is loaded lazily
//     with dlopen so that a single-GPU process never pays for (or depends on) librccl.
//   * direct (IPXK_COMM=direct): a hand-written one-shot reduce-scatter + all-gather over buffers that
//     the ranks map into each other's address space with hipIpc (DirectComm below).  xGMI is
//     point-to-point, so a ring all-reduce of S bytes is bound by ONE link, 2 (R-1)/R S / 153 GB/s
//     (183 us for the 16 MB of config 4), whereas here every rank pulls its 1/R segment from all R-1
//     peers at once over R-1 links and then the reduced segments the same way: 2 S / R per link
//     (26 us).
// The direct transport also works between processes that share ONE GPU (hipIpc within a device), which is
// how the multi-process tests run the partitioned code paths on a one-GPU box (RCCL refuses two ranks on
// one device).
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "context.hpp"

namespace ipxk {

namespace {
// the few RCCL declarations used (ABI of rccl.h, ROCm 7.2)
typedef struct { char internal[128]; } RcclUniqueId;
enum { kNcclFloat64 = 8 };   // ncclDouble
enum { kNcclSum = 0, kNcclMax = 2, kNcclMin = 3 };
typedef int (*GetUniqueIdFn)(RcclUniqueId*);
typedef int (*CommInitRankFn)(ncclComm**, int, RcclUniqueId, int);
typedef int (*CommDestroyFn)(ncclComm*);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, ncclComm*, hipStream_t);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, ncclComm*, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*CommCountFn)(const ncclComm*, int*);
typedef int (*CommUserRankFn)(const ncclComm*, int*);

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    AllGatherFn all_gather = nullptr;
    GetErrorStringFn error_string = nullptr;
    CommCountFn comm_count = nullptr;
    CommUserRankFn comm_user_rank = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    if (!r.handle) {
        r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) r.handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) throw Error(IPXK_E_HIP, std::string("cannot load librccl: ") + dlerror());
        r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
        r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
        r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
        r.all_reduce = (AllReduceFn)dlsym(r.handle, "ncclAllReduce");
        r.all_gather = (AllGatherFn)dlsym(r.handle, "ncclAllGather");
        r.error_string = (GetErrorStringFn)dlsym(r.handle, "ncclGetErrorString");
        r.comm_count = (CommCountFn)dlsym(r.handle, "ncclCommCount");
        r.comm_user_rank = (CommUserRankFn)dlsym(r.handle, "ncclCommUserRank");
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.all_gather)
            throw Error(IPXK_E_HIP, "librccl lacks an expected symbol");
    }
    return r;
}

void check(int rc, const char* what) {
    if (rc != 0) {
        Rccl& r = rccl();
        throw Error(IPXK_E_HIP, std::string(what) + " failed: " +
                                    (r.error_string ? r.error_string(rc) : "rccl error"));
    }
}
}  // namespace


// ---------------------------------------------------------------------------
// Direct exchange (IPXK_COMM=direct).
// Every rank owns two device buffers that all ranks of the node map with hipIpc:
//   S  (capacity doubles): the operand of the running collective
//   T  (capacity / R + pad): the segment this rank has reduced
// and a flag table in a POSIX shared-memory segment that every process registers with HIP
// (fine-grained host memory: a value stored there is visible to every GPU without any cache
// maintenance).  One all-reduce of `count` doubles in round e (e counts this communicator's collectives):
//   1. copy the operand into S (a producer may also write S directly);  stream-ordered store ready[r] = e
//   2. reduce kernel: waits until ready[p] >= e for all p (bounded polling of the flag table), then
//      sums segment r of every S_p IN RANK ORDER into T_r -- every rank forms every sum in the same
//      order, so the replicated result is bitwise identical everywhere;  store reduced[r] = e
//   3. gather kernel: waits for reduced[p] >= e, copies every T_p into the caller's buffer.
// S_r may be overwritten again once every peer has passed step 2 of this round, which the next
// round's flag protocol implies (a peer stores ready[e+1] only after its own step 3 of round e, and
// nobody reads S before all ready[e+1] are in).  Remote data is read with system-scope loads.  The
// ranks may live on one GPU (the tests: hipIpc between processes of one device) or on eight.
// ---------------------------------------------------------------------------
constexpr int kMaxDirectRanks = 16;
struct PeerTable { const double* S[kMaxDirectRanks]; const double* T[kMaxDirectRanks]; };

__device__ __forceinline__ double load_sys(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_SYSTEM));
}
// all lanes of the calling workgroup return once flags[p * stride] >= epoch for every rank p; false on timeout
// wait budget of the exchange kernels in 100 MHz ticks (set from IPXK_COMM_TIMEOUT_S when the communicator is built)
__device__ unsigned long long g_comm_timeout_ticks = 120ull * 100000000ull;

__device__ __forceinline__ bool wait_ranks(const unsigned* flags, int stride, int nranks, unsigned epoch, int* abort_flag) {
    __shared__ int ok_shared;
    if (threadIdx.x == 0) ok_shared = 1;
    __syncthreads();
    if ((int)threadIdx.x < nranks) {
        // bounded wait: by wall clock (s_memrealtime ticks at 100 MHz), not by a poll count whose duration depends on
        // the load of the host-memory path; the budget comes from IPXK_COMM_TIMEOUT_S (default 120 s)
        const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
        int spins = 0;
        while ((int)(__hip_atomic_load(flags + threadIdx.x * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (((++spins & 1023) == 0 && (__builtin_amdgcn_s_memrealtime() - t_begin > g_comm_timeout_ticks ||
                                            __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok_shared = 0;
                break;
            }
        }
    }
    __syncthreads();
    return ok_shared != 0;
}

// op: 0 sum, 1 max, 2 min
__global__ __launch_bounds__(kBlock) void direct_reduce_kernel(PeerTable P, int rank, int nranks, int64_t seg, int64_t count,
                                                               int op, double* __restrict__ T, const unsigned* ready,
                                                               unsigned epoch, int* abort_flag) {
    if (!wait_ranks(ready, 16, nranks, epoch, abort_flag)) return;
    const int64_t lo = rank * seg, hi = lo + seg < count ? lo + seg : count;
    for (int64_t i = lo + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < hi; i += (int64_t)gridDim.x * kBlock) {
        double acc = load_sys(P.S[0] + i);
        for (int p = 1; p < nranks; p++) {
            const double v = load_sys(P.S[p] + i);
            acc = op == 0 ? acc + v : op == 1 ? (acc != acc ? acc : (v != v ? v : (acc > v ? acc : v))) : (acc < v ? acc : v);
        }
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(T) + (i - lo), (unsigned long long)__double_as_longlong(acc),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(kBlock) void direct_gather_kernel(PeerTable P, int nranks, int64_t seg, int64_t count,
                                                               double* __restrict__ out, const unsigned* reduced,
                                                               unsigned epoch, int* abort_flag) {
    if (!wait_ranks(reduced, 16, nranks, epoch, abort_flag)) return;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (int64_t)gridDim.x * kBlock) {
        const int p = (int)(i / seg);
        out[i] = load_sys(P.T[p] + (i - p * seg));
    }
}
// all-gather: recv[p * count + i] = S_p[i]
__global__ __launch_bounds__(kBlock) void direct_allgather_kernel(PeerTable P, int nranks, int64_t count, double* __restrict__ recv,
                                                                  const unsigned* ready, unsigned epoch, int* abort_flag) {
    if (!wait_ranks(ready, 16, nranks, epoch, abort_flag)) return;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count * nranks; i += (int64_t)gridDim.x * kBlock) {
        const int p = (int)(i / count);
        recv[i] = load_sys(P.S[p] + (i - p * count));
    }
}

struct DirectComm {
    struct Header { std::atomic<int> count; std::atomic<int> generation; };
    // segment layout: Header | pad to 256 | handles: R x 2 x 64 B | pad to 4096 | flags: 2 x R x 64 B
    static constexpr size_t kHandlesAt = 256, kDevicesAt = 2400, kFlagsAt = 4096;
    std::string name;
    int rank = 0, nranks = 1;
    bool owner = false;
    char* seg = nullptr;
    size_t seg_bytes = 0;
    Header* hdr = nullptr;
    unsigned* flags_dev = nullptr;           // device view of the flag table (host memory)
    size_t capacity = 0;                     // doubles in S
    double* S = nullptr;
    double* T = nullptr;
    PeerTable peers{};
    std::vector<void*> opened;
    DevBuf<int> abort_flag;
    unsigned epoch = 0;

    unsigned* ready_dev() const { return flags_dev; }                          // ready[p] at p * 16 words
    unsigned* reduced_dev() const { return flags_dev + (size_t)nranks * 16; }
    void host_barrier(int limit_s = 120) {
        const int gen = hdr->generation.load();
        if (hdr->count.fetch_add(1) + 1 == nranks) { hdr->count.store(0); hdr->generation.fetch_add(1); return; }
        const auto t0 = std::chrono::steady_clock::now();
        while (hdr->generation.load() == gen) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(limit_s))
                throw Error(IPXK_E_HIP, "direct exchange: a rank did not reach the barrier in time");
        }
    }
    bool aborted() const {
        int f = 0;
        return abort_flag.size() && hipMemcpy(&f, abort_flag.get(), sizeof f, hipMemcpyDeviceToHost) == hipSuccess && f != 0;
    }
    void open(const std::string& nm, int r, int n, size_t cap) {
        IPXK_REQUIRE(n <= kMaxDirectRanks, "direct exchange supports at most 16 ranks");
        name = nm; rank = r; nranks = n;
        seg_bytes = kFlagsAt + (size_t)2 * n * 64 + 4096;
        int fd = -1;
        if (r == 0) {
            fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)seg_bytes) != 0) throw Error(IPXK_E_HIP, "direct exchange: cannot create the flag segment");
            owner = true;
        } else {
            for (int tries = 0; tries < 6000 && fd < 0; tries++) {
                fd = shm_open(name.c_str(), O_RDWR, 0600);
                if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
            if (fd < 0) throw Error(IPXK_E_HIP, "direct exchange: flag segment of rank 0 not found");
            for (int tries = 0; tries < 6000; tries++) {
                if (lseek(fd, 0, SEEK_END) >= (off_t)seg_bytes) break;
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        void* p = mmap(nullptr, seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) throw Error(IPXK_E_HIP, "direct exchange: mmap failed");
        seg = static_cast<char*>(p);
        hdr = reinterpret_cast<Header*>(seg);
        IPXK_HIP(hipHostRegister(seg, seg_bytes, hipHostRegisterMapped));
        void* dev = nullptr;
        IPXK_HIP(hipHostGetDevicePointer(&dev, seg + kFlagsAt, 0));
        flags_dev = static_cast<unsigned*>(dev);
        // exchange buffers: uncached device memory where the runtime offers it (nothing of a peer's data may
        // linger in an L2), plain device memory otherwise
        capacity = (cap + 15) / 16 * 16;
        const size_t tcap = capacity / n + 64;
        if (hipExtMallocWithFlags(reinterpret_cast<void**>(&S), capacity * sizeof(double), hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            IPXK_HIP(hipMalloc(reinterpret_cast<void**>(&S), capacity * sizeof(double)));
        }
        if (hipExtMallocWithFlags(reinterpret_cast<void**>(&T), tcap * sizeof(double), hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            IPXK_HIP(hipMalloc(reinterpret_cast<void**>(&T), tcap * sizeof(double)));
        }
        hipIpcMemHandle_t hs, ht;
        IPXK_HIP(hipIpcGetMemHandle(&hs, S));
        IPXK_HIP(hipIpcGetMemHandle(&ht, T));
        static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
        memcpy(seg + kHandlesAt + (size_t)r * 128, &hs, 64);
        memcpy(seg + kHandlesAt + (size_t)r * 128 + 64, &ht, 64);
        int my_device = 0;
        IPXK_HIP(hipGetDevice(&my_device));
        int* devices = reinterpret_cast<int*>(seg + kDevicesAt);
        devices[r] = my_device;
        host_barrier();                                   // every handle is in the segment
        for (int q = 0; q < n; q++) {                     // peers on other GPUs must be reachable over xGMI / PCIe
            if (devices[q] == my_device) continue;
            int can = 0;
            IPXK_HIP(hipDeviceCanAccessPeer(&can, my_device, devices[q]));
            if (!can) throw Error(IPXK_E_UNSUPPORTED, "direct exchange: no peer access between the GPUs of two ranks");
        }
        for (int q = 0; q < n; q++) {
            if (q == r) { peers.S[q] = S; peers.T[q] = T; continue; }
            hipIpcMemHandle_t a, b;
            memcpy(&a, seg + kHandlesAt + (size_t)q * 128, 64);
            memcpy(&b, seg + kHandlesAt + (size_t)q * 128 + 64, 64);
            void *ps = nullptr, *pt = nullptr;
            IPXK_HIP(hipIpcOpenMemHandle(&ps, a, hipIpcMemLazyEnablePeerAccess));
            IPXK_HIP(hipIpcOpenMemHandle(&pt, b, hipIpcMemLazyEnablePeerAccess));
            opened.push_back(ps); opened.push_back(pt);
            peers.S[q] = static_cast<const double*>(ps);
            peers.T[q] = static_cast<const double*>(pt);
        }
        abort_flag.resize(1);
        IPXK_HIP(hipMemset(abort_flag.get(), 0, sizeof(int)));
        {
            double seconds = 120.0;
            if (const char* e = getenv("IPXK_COMM_TIMEOUT_S")) if (atof(e) > 0.0) seconds = atof(e);
            const unsigned long long ticks = (unsigned long long)(seconds * 1e8);
            IPXK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_comm_timeout_ticks), &ticks, sizeof ticks));
        }
        host_barrier();                                   // everybody has mapped everything
    }
    ~DirectComm() {
        for (void* p : opened) (void)hipIpcCloseMemHandle(p);
        // nobody unmaps a buffer that a peer still has open -- unless the exchange already failed (a peer may be gone:
        // then a short wait only)
        if (hdr) { try { host_barrier(aborted() ? 2 : 120); } catch (...) {} }
        if (S) (void)hipFree(S);
        if (T) (void)hipFree(T);
        if (seg) { (void)hipHostUnregister(seg); munmap(seg, seg_bytes); }
        if (owner) shm_unlink(name.c_str());
    }
    static int grid_for(int64_t n) { return (int)std::min<int64_t>(512, std::max<int64_t>(1, (n + kBlock - 1) / kBlock)); }

    void signal(Context* c, unsigned* flag_of_mine) {
        IPXK_HIP(hipStreamWriteValue32(c->stream, flag_of_mine, epoch, 0));
    }
    // op: 0 sum, 1 max, 2 min; the operand is buf (copied into S) or, if `staged`, already in S; result in buf
    void allreduce(Context* c, double* buf, size_t count, int op, bool staged = false) {
        IPXK_REQUIRE(count <= capacity, "direct exchange: operand larger than the exchange buffer");
        hipStream_t s = c->stream;
        epoch++;
        if (!staged) IPXK_HIP(hipMemcpyAsync(S, buf, count * sizeof(double), hipMemcpyDeviceToDevice, s));
        signal(c, ready_dev() + (size_t)rank * 16);
        const int64_t seg_len = ((int64_t)count + nranks - 1) / nranks;
        hipLaunchKernelGGL(direct_reduce_kernel, dim3(grid_for(seg_len)), dim3(kBlock), 0, s, peers, rank, nranks, seg_len,
                           (int64_t)count, op, T, ready_dev(), epoch, abort_flag.get());
        signal(c, reduced_dev() + (size_t)rank * 16);
        hipLaunchKernelGGL(direct_gather_kernel, dim3(grid_for((int64_t)count)), dim3(kBlock), 0, s, peers, nranks, seg_len,
                           (int64_t)count, buf, reduced_dev(), epoch, abort_flag.get());
    }
    void allgather(Context* c, const double* send, double* recv, size_t count) {
        IPXK_REQUIRE(count <= capacity, "direct exchange: operand larger than the exchange buffer");
        hipStream_t s = c->stream;
        epoch++;
        IPXK_HIP(hipMemcpyAsync(S, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
        signal(c, ready_dev() + (size_t)rank * 16);
        hipLaunchKernelGGL(direct_allgather_kernel, dim3(grid_for((int64_t)count * nranks)), dim3(kBlock), 0, s, peers, nranks,
                           (int64_t)count, recv, ready_dev(), epoch, abort_flag.get());
        // S is reused by the next collective: its `ready` store must not overtake a peer that still reads this
        // round's S.  A second flag round closes that window (the gather kernel of an all-reduce does the same).
        signal(c, reduced_dev() + (size_t)rank * 16);
        hipLaunchKernelGGL(direct_gather_kernel, dim3(1), dim3(kBlock), 0, s, peers, nranks, (int64_t)1, (int64_t)0,
                           (double*)nullptr, reduced_dev(), epoch, abort_flag.get());
    }
    void check(Context* c) {      // host side, after a stream synchronisation
        int flag = 0;
        IPXK_HIP(hipMemcpy(&flag, abort_flag.get(), sizeof(int), hipMemcpyDeviceToHost));
        if (flag) throw Error(IPXK_E_HIP, "direct exchange: a rank did not arrive at a collective (bounded wait expired)");
    }
};
void destroy_direct(DirectComm* d) { delete d; }
constexpr char kDirectMagic[8] = {'I', 'P', 'X', 'K', 'D', 'I', 'R', 0};

// IPXK_FORCE_COMM=1 sends a single rank through the collective code path (used by the
// GPU tests: the one-GPU test box cannot host a second rank).
bool comm_active(const Context* c) {
    return (c->comm != nullptr || c->direct != nullptr) && (c->nranks > 1 || c->force_comm);
}

bool comm_rows(const Context* c) { return comm_active(c) && !c->col_partition; }
bool comm_cols(const Context* c) { return comm_active(c) && c->col_partition; }

void comm_allreduce_min(Context* c, double* buf, size_t count) {
    if (!comm_active(c) || count == 0) return;
    if (c->direct) { c->direct->allreduce(c, buf, count, 2); return; }
    check(rccl().all_reduce(buf, buf, count, kNcclFloat64, kNcclMin, c->comm, c->stream), "ncclAllReduce");
}

void comm_allreduce_sum(Context* c, double* buf, size_t count) {
    if (!comm_active(c) || count == 0) return;
    if (c->direct) { c->direct->allreduce(c, buf, count, 0); return; }
    check(rccl().all_reduce(buf, buf, count, kNcclFloat64, kNcclSum, c->comm, c->stream), "ncclAllReduce");
}

void comm_allreduce_max(Context* c, double* buf, size_t count) {
    if (!comm_active(c) || count == 0) return;
    if (c->direct) { c->direct->allreduce(c, buf, count, 1); return; }
    check(rccl().all_reduce(buf, buf, count, kNcclFloat64, kNcclMax, c->comm, c->stream), "ncclAllReduce");
}

void comm_allgather(Context* c, const double* send, double* recv, size_t count_per_rank) {
    if (!comm_active(c)) {
        if (send != recv)
            IPXK_HIP(hipMemcpyAsync(recv, send, count_per_rank * sizeof(double), hipMemcpyDeviceToDevice,
                                    c->stream));
        return;
    }
    if (c->direct) { c->direct->allgather(c, send, recv, count_per_rank); return; }
    check(rccl().all_gather(send, recv, count_per_rank, kNcclFloat64, c->comm, c->stream), "ncclAllGather");
}

void comm_check(Context* c) { if (c->direct) c->direct->check(c); }

// The direct transport reduces out of its exchange buffer: a producer that writes its operand there spares
// the copy.  Returns that buffer (or nullptr: write to the destination and call comm_allreduce_sum).
double* comm_stage(Context* c, size_t count) {
    return comm_active(c) && c->direct && count <= c->direct->capacity ? c->direct->S : nullptr;
}
// dst = sum over the ranks of the staged operands
void comm_allreduce_sum_staged(Context* c, double* dst, size_t count) {
    IPXK_REQUIRE(c->direct != nullptr, "no staged operand without the direct transport");
    c->direct->allreduce(c, dst, count, 0, true);
}

void comm_destroy(Context* c) {
    if (c->direct) { destroy_direct(c->direct); c->direct = nullptr; }
    if (c->comm) {
        (void)rccl().comm_destroy(c->comm);
        c->comm = nullptr;
    }
}

}  // namespace ipxk

using namespace ipxk;

extern "C" int ipxk_comm_unique_id(void* id128) {
    try {
        if (!id128) throw Error(IPXK_E_ARGUMENT, "id128 is NULL");
        if (const char* e = getenv("IPXK_COMM")) {
            if (std::string(e) == "direct") {       // the id names the segment that carries handles and flags
                char buf[128] = {0};
                memcpy(buf, kDirectMagic, 8);
                snprintf(buf + 8, 100, "/ipxkd_%d_%lld", (int)getpid(),
                         (long long)std::chrono::steady_clock::now().time_since_epoch().count());
                memcpy(id128, buf, 128);
                return IPXK_OK;
            }
        }
        RcclUniqueId id;
        check(rccl().get_unique_id(&id), "ncclGetUniqueId");
        memcpy(id128, &id, sizeof id);
        return IPXK_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    }
}

static int comm_init_impl(ipxk_context* c, const void* id128, int rank, int nranks, bool columns) {
    try {
        if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks)
            throw Error(IPXK_E_ARGUMENT, "ipxk_comm_init: bad argument");
        IPXK_HIP(hipSetDevice(c->device));
        comm_destroy(c);
        if (memcmp(id128, kDirectMagic, 8) == 0) {
            c->direct = new DirectComm;
            c->direct->open(std::string(static_cast<const char*>(id128) + 8), rank, nranks,
                            (size_t)std::max<int64_t>(std::max<int64_t>(c->n, c->m), 64));
        } else {
            RcclUniqueId id;
            memcpy(&id, id128, sizeof id);
            check(rccl().comm_init_rank(&c->comm, nranks, id, rank), "ncclCommInitRank");
        }
        c->rank = rank;
        c->nranks = nranks;
        c->force_comm = getenv("IPXK_FORCE_COMM") != nullptr;
        c->col_partition = columns;
        if (columns) {          // every rank holds all m rows
            c->m_global = c->m;
            return IPXK_OK;
        }
        // global row count (defines the default iteration cap m+100 identically on every rank)
        DevBuf<double> cnt(1);
        const double mine = (double)c->m;
        IPXK_HIP(hipMemcpyAsync(cnt.get(), &mine, sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (c->direct) c->direct->allreduce(c, cnt.get(), 1, 0);
        else check(rccl().all_reduce(cnt.get(), cnt.get(), 1, kNcclFloat64, kNcclSum, c->comm, c->stream), "ncclAllReduce");
        double total = 0.0;
        IPXK_HIP(hipMemcpyAsync(&total, cnt.get(), sizeof(double), hipMemcpyDeviceToHost, c->stream));
        IPXK_HIP(hipStreamSynchronize(c->stream));
        c->m_global = (int64_t)(total + 0.5);
        return IPXK_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    }
}

// what the transport itself says about the communicator (for the records of a multi-GPU run): RCCL's ncclCommCount /
// ncclCommUserRank, or the rank table of the direct exchange
extern "C" int ipxk_comm_info(const ipxk_context* c, int* transport, int* nranks, int* rank) {
    try {
        if (!c || !transport || !nranks || !rank) throw Error(IPXK_E_ARGUMENT, "ipxk_comm_info: bad argument");
        *transport = 0; *nranks = 1; *rank = 0;
        if (c->direct) { *transport = 2; *nranks = c->direct->nranks; *rank = c->direct->rank; }
        else if (c->comm) {
            *transport = 1;
            Rccl& r = rccl();
            if (!r.comm_count || !r.comm_user_rank) throw Error(IPXK_E_HIP, "librccl lacks ncclCommCount / ncclCommUserRank");
            check(r.comm_count(c->comm, nranks), "ncclCommCount");
            check(r.comm_user_rank(c->comm, rank), "ncclCommUserRank");
        }
        return IPXK_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    }
}

extern "C" int ipxk_comm_init(ipxk_context* c, const void* id128, int rank, int nranks) {
    return comm_init_impl(c, id128, rank, nranks, false);
}

extern "C" int ipxk_comm_init_columns(ipxk_context* c, const void* id128, int rank, int nranks) {
    return comm_init_impl(c, id128, rank, nranks, true);
}
