// Multi-GPU exchange step: rows of AI are partitioned over the ranks of one node, each
// rank computes its partial t_g = Ws .* (A_g' y_g) and one RCCL all-reduce over xGMI sums the
// n-vector before the second pass.  RCCL is loaded lazily with dlopen so
// that a single-GPU process never pays for (or depends on) librccl.
#include <dlfcn.h>

#include <cstdlib>

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

// IPXK_FORCE_COMM=1 sends a single rank through the collective code path (used by the
// GPU tests: the one-GPU test box cannot host a second rank).
bool comm_active(const Context* c) { return c->comm != nullptr && (c->nranks > 1 || c->force_comm); }

bool comm_rows(const Context* c) { return comm_active(c) && !c->col_partition; }
bool comm_cols(const Context* c) { return comm_active(c) && c->col_partition; }

void comm_allreduce_min(Context* c, double* buf, size_t count) {
    if (!comm_active(c) || count == 0) return;
    check(rccl().all_reduce(buf, buf, count, kNcclFloat64, kNcclMin, c->comm, c->stream), "ncclAllReduce");
}

void comm_allreduce_sum(Context* c, double* buf, size_t count) {
    if (!comm_active(c) || count == 0) return;
    check(rccl().all_reduce(buf, buf, count, kNcclFloat64, kNcclSum, c->comm, c->stream), "ncclAllReduce");
}

void comm_allreduce_max(Context* c, double* buf, size_t count) {
    if (!comm_active(c) || count == 0) return;
    check(rccl().all_reduce(buf, buf, count, kNcclFloat64, kNcclMax, c->comm, c->stream), "ncclAllReduce");
}

void comm_allgather(Context* c, const double* send, double* recv, size_t count_per_rank) {
    if (!comm_active(c)) {
        if (send != recv)
            IPXK_HIP(hipMemcpyAsync(recv, send, count_per_rank * sizeof(double), hipMemcpyDeviceToDevice,
                                    c->stream));
        return;
    }
    check(rccl().all_gather(send, recv, count_per_rank, kNcclFloat64, c->comm, c->stream), "ncclAllGather");
}

void comm_destroy(Context* c) {
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
        RcclUniqueId id;
        memcpy(&id, id128, sizeof id);
        check(rccl().comm_init_rank(&c->comm, nranks, id, rank), "ncclCommInitRank");
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
        check(rccl().all_reduce(cnt.get(), cnt.get(), 1, kNcclFloat64, kNcclSum, c->comm, c->stream), "ncclAllReduce");
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
