// Multi-GPU exchange step: rows of AI are partitioned over the ranks of one node, each
// rank computes its partial t_g = Ws .* (A_g' y_g) and one RCCL all-reduce over xGMI sums the
// n-vector before the second pass.  RCCL is loaded lazily with dlopen so
// that a single-GPU process never pays for (or depends on) librccl.
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

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    AllGatherFn all_gather = nullptr;
    GetErrorStringFn error_string = nullptr;
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
// Test transport (IPXK_COMM=hostshm).  RCCL refuses two ranks on one device, and the test boxes
// have one GPU: to run the partitioned code paths with REAL separate rank processes on a single
// GPU, the collectives can be carried through POSIX shared memory instead -- synchronise the
// stream, copy the operand to the rank's slot, meet at a process barrier, reduce all slots in
// rank order (bitwise identical on every rank), copy back.  Slow and blocking; never selected
// unless the environment asks for it.  Every wait is bounded.
// ---------------------------------------------------------------------------
struct ShmComm {
    static constexpr size_t kSlotBytes = size_t(40) << 20;
    struct Header { std::atomic<int> count; std::atomic<int> generation; };
    std::string name;
    int rank = 0, nranks = 1;
    Header* hdr = nullptr;
    char* slots = nullptr;
    size_t bytes = 0;
    bool owner = false;

    void open(const std::string& nm, int r, int n) {
        name = nm; rank = r; nranks = n;
        bytes = sizeof(Header) + 64 + (size_t)n * kSlotBytes;
        int fd = -1;
        if (r == 0) {
            fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) throw Error(IPXK_E_HIP, "hostshm: cannot create the segment");
            owner = true;
        } else {
            for (int tries = 0; tries < 3000 && fd < 0; tries++) {   // rank 0 may not be there yet
                fd = shm_open(name.c_str(), O_RDWR, 0600);
                if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
            if (fd < 0) throw Error(IPXK_E_HIP, "hostshm: segment of rank 0 not found");
            for (int tries = 0; tries < 3000; tries++) {              // ... or not sized yet
                if (lseek(fd, 0, SEEK_END) >= (off_t)bytes) break;
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) throw Error(IPXK_E_HIP, "hostshm: mmap failed");
        hdr = static_cast<Header*>(p);          // a fresh segment is zero-filled
        slots = static_cast<char*>(p) + sizeof(Header) + 64;
    }
    ~ShmComm() {
        if (hdr) munmap(hdr, bytes);
        if (owner) shm_unlink(name.c_str());
    }
    void barrier() {
        const int gen = hdr->generation.load();
        if (hdr->count.fetch_add(1) + 1 == nranks) {
            hdr->count.store(0);
            hdr->generation.fetch_add(1);
            return;
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (hdr->generation.load() == gen) {
            std::this_thread::sleep_for(std::chrono::microseconds(20));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
                throw Error(IPXK_E_HIP, "hostshm: a rank did not reach the collective within 60 s");
        }
    }
    double* slot(int r) const { return reinterpret_cast<double*>(slots + (size_t)r * kSlotBytes); }
    // op: 0 sum, 1 max, 2 min
    void allreduce(Context* c, double* buf, size_t count, int op) {
        IPXK_REQUIRE(count * sizeof(double) <= kSlotBytes, "hostshm: operand too large for the test transport");
        staged_d2h(slot(rank), buf, count * sizeof(double), c->stream);
        barrier();
        std::vector<double> acc(slot(0), slot(0) + count);
        for (int r = 1; r < nranks; r++) {
            const double* s = slot(r);
            for (size_t i = 0; i < count; i++)
                acc[i] = op == 0 ? acc[i] + s[i] : op == 1 ? std::max(acc[i], s[i]) : std::min(acc[i], s[i]);
        }
        barrier();                                   // everybody has read the slots
        staged_h2d(buf, acc.data(), count * sizeof(double), c->stream);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    }
    void allgather(Context* c, const double* send, double* recv, size_t count) {
        IPXK_REQUIRE(count * sizeof(double) <= kSlotBytes, "hostshm: operand too large for the test transport");
        staged_d2h(slot(rank), send, count * sizeof(double), c->stream);
        barrier();
        std::vector<double> all((size_t)nranks * count);
        for (int r = 0; r < nranks; r++) memcpy(all.data() + (size_t)r * count, slot(r), count * sizeof(double));
        barrier();
        staged_h2d(recv, all.data(), all.size() * sizeof(double), c->stream);
        IPXK_HIP(hipStreamSynchronize(c->stream));
    }
};
void destroy_shm(ShmComm* s) { delete s; }
constexpr char kShmMagic[8] = {'I', 'P', 'X', 'K', 'S', 'H', 'M', 0};

// IPXK_FORCE_COMM=1 sends a single rank through the collective code path (used by the
// GPU tests: the one-GPU test box cannot host a second rank).
bool comm_active(const Context* c) {
    return (c->comm != nullptr || c->shm != nullptr) && (c->nranks > 1 || c->force_comm);
}

bool comm_rows(const Context* c) { return comm_active(c) && !c->col_partition; }
bool comm_cols(const Context* c) { return comm_active(c) && c->col_partition; }

void comm_allreduce_min(Context* c, double* buf, size_t count) {
    if (!comm_active(c) || count == 0) return;
    if (c->shm) { c->shm->allreduce(c, buf, count, 2); return; }
    check(rccl().all_reduce(buf, buf, count, kNcclFloat64, kNcclMin, c->comm, c->stream), "ncclAllReduce");
}

void comm_allreduce_sum(Context* c, double* buf, size_t count) {
    if (!comm_active(c) || count == 0) return;
    if (c->shm) { c->shm->allreduce(c, buf, count, 0); return; }
    check(rccl().all_reduce(buf, buf, count, kNcclFloat64, kNcclSum, c->comm, c->stream), "ncclAllReduce");
}

void comm_allreduce_max(Context* c, double* buf, size_t count) {
    if (!comm_active(c) || count == 0) return;
    if (c->shm) { c->shm->allreduce(c, buf, count, 1); return; }
    check(rccl().all_reduce(buf, buf, count, kNcclFloat64, kNcclMax, c->comm, c->stream), "ncclAllReduce");
}

void comm_allgather(Context* c, const double* send, double* recv, size_t count_per_rank) {
    if (!comm_active(c)) {
        if (send != recv)
            IPXK_HIP(hipMemcpyAsync(recv, send, count_per_rank * sizeof(double), hipMemcpyDeviceToDevice,
                                    c->stream));
        return;
    }
    if (c->shm) { c->shm->allgather(c, send, recv, count_per_rank); return; }
    check(rccl().all_gather(send, recv, count_per_rank, kNcclFloat64, c->comm, c->stream), "ncclAllGather");
}

void comm_destroy(Context* c) {
    if (c->shm) { destroy_shm(c->shm); c->shm = nullptr; }
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
            if (std::string(e) == "hostshm") {      // test transport: the id names a shared-memory segment
                char buf[128] = {0};
                memcpy(buf, kShmMagic, 8);
                snprintf(buf + 8, 100, "/ipxk_%d_%lld", (int)getpid(),
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
        if (memcmp(id128, kShmMagic, 8) == 0) {
            c->shm = new ShmComm;
            c->shm->open(std::string(static_cast<const char*>(id128) + 8), rank, nranks);
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
        if (c->shm) c->shm->allreduce(c, cnt.get(), 1, 0);
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

extern "C" int ipxk_comm_init(ipxk_context* c, const void* id128, int rank, int nranks) {
    return comm_init_impl(c, id128, rank, nranks, false);
}

extern "C" int ipxk_comm_init_columns(ipxk_context* c, const void* id128, int rank, int nranks) {
    return comm_init_impl(c, id128, rank, nranks, true);
}
