// Basis-preconditioned operator on the device.
//   SplittedNormalMatrix::Prepare / _Apply      reference src/splitted_normal_matrix.cc:18-117
//   TriangularSolve / ForwardSolve / BackwardSolve   src/sparse_matrix.cc:224-311
//   Basis::SolveDense on fresh factors               src/basis.cc:168-170, src/forrest_tomlin.cc:67-78
//
// Triangular solves are level-scheduled gather sweeps.  For each of the four sweeps
// (U', L', L, U) the host computes the dependency level of every unknown once per
// Prepare (the factors change every IPM iteration) and stores the rows level by level.  A wide
// level is one launch with one thread (or, for long rows, 8 lanes) per unknown; a run of narrow
// levels is ONE single-workgroup launch that keeps the run's entries, unknowns and metadata in
// LDS and separates levels by workgroup barriers (tail_lds_kernel).  Every row is summed in the
// reference's order:
//   transposed sweeps ('t'):  d = sum x[i]*a (ascending storage order); x = (x - d)/diag
//   forward sweeps   ('n'):   x -= a*x_j one at a time in the reference's column order
// so a sweep reproduces the reference's arithmetic (bit-exact given identical factors).
//
// N N' is applied through the resident model matrix: N = AI[:,nonbasic] scaled by D and with
// rows in pivot order, hence N N' w = P A (M D^2) A' P' w with M the nonbasic mask -- the
// NormalMatrix kernels with weights W = M.*D^2 between two permutation kernels.  Prepare
// therefore uploads O(m + n) numbers plus the factors and never copies the matrix
// (the reference copies all of N every time, splitted_normal_matrix.cc:42-55).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>
#include <thread>

#include "context.hpp"
#include "spmv_kernels.hpp"
#include "trisolve.hpp"

namespace ipxk {

void destroy_split(SplitOperator* s) { delete s; }

static int vec_grid(int64_t len) {
    int64_t g = (len + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return (int)(g < 1024 ? g : 1024);
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// One unknown per group of GL lanes.  The group's lanes load the row's entries side by side
// (coalesced, all memory latency overlapped); the products are then combined ONE AT A TIME in
// storage order through shuffles, so the arithmetic is the reference's sequential arithmetic
// even for the long rows near the end of a forward sweep.
// acc -= / += the products held by the first `cnt` lanes of the group, one at a time in lane
// order.  All GL shuffles are issued before the first add, so their latencies overlap and only
// the adds are serial.
template <bool RUNNING, int GL>
__device__ __forceinline__ double ordered_combine(double acc, double prod, int cnt, int gbase) {
    if (GL == 1) return RUNNING ? acc - prod : acc + prod;
    double v[GL];
#pragma unroll
    for (int t = 0; t < GL; t++) v[t] = __shfl(prod, gbase + t, 64);
#pragma unroll
    for (int t = 0; t < GL; t++)
        if (t < cnt) acc = RUNNING ? acc - v[t] : acc + v[t];
    return acc;
}

template <bool RUNNING, int GL>
__device__ __forceinline__ void solve_unknown(const SweepView& S, int k, double* x) {
    const int lane = threadIdx.x & 63, gl = lane & (GL - 1), gbase = lane & ~(GL - 1);
    // one round trip for the whole record (none of these loads depends on another)
    const int r = S.order[k];
    const int p0 = S.ptr[k], p1 = S.ptr[k + 1];
    const double dg = S.diag[k];
    if (r < 0) return;   // padding slot (whole groups are padding or real together)
    const double xr = x[r];
    double acc = RUNNING ? xr : 0.0;
    for (int base = p0; base < p1; base += GL) {
        const int p = base + gl;
        double prod = 0.0;
        if (p < p1) prod = RUNNING ? S.val[p] * x[S.idx[p]] : x[S.idx[p]] * S.val[p];
        acc = ordered_combine<RUNNING, GL>(acc, prod, min(GL, p1 - base), gbase);
    }
    const double res = (RUNNING ? acc : xr - acc) / dg;
    if (gl == 0) x[r] = res;
}

// one level per launch
template <bool RUNNING, int GL>
__global__ __launch_bounds__(kBlock) void level_kernel(SweepView S, int k0, int k1, double* x,
                                                       const int* done) {
    if (done && *done) return;
    const int k = k0 + (blockIdx.x * kBlock + threadIdx.x) / GL;
    if (k < k1) solve_unknown<RUNNING, GL>(S, k, x);
}

// A run of narrow levels with EVERYTHING in LDS.  Before the first level the workgroup loads, in
// parallel, every entry of every unknown of the run: the product with an x value that is already
// final (computed by an earlier launch), or the bare coefficient plus the run-local slot of the
// dependency when that unknown belongs to the run itself; and per unknown its right-hand side,
// row extent and diagonal.  The level loop then touches only LDS -- no global load or store is
// outstanding at its barriers -- so a level costs a barrier plus a few LDS round trips instead of
// two or three L2 round trips.  The run's x values go back to global memory once, at the end.
// Sums are formed in the same storage order as in solve_unknown.  Levels of long and short rows
// mix freely in one run.
struct TailLds {                      // carve-up of the dynamic LDS block (150.0 KiB)
    static constexpr size_t pv = 0;                                   // double[kTailEntries]
    static constexpr size_t xt = pv + (size_t)kTailEntries * 8;       // double[kTailSlots]
    static constexpr size_t dg = xt + (size_t)kTailSlots * 8;         // double[kTailSlots]
    static constexpr size_t rp = dg + (size_t)kTailSlots * 8;         // int[kTailSlots + 1]
    static constexpr size_t lp = rp + (size_t)(kTailSlots + 1) * 4;   // int[kTailLevelsLds + 1]
    static constexpr size_t sl = lp + (size_t)(kTailLevelsLds + 1) * 4;   // short[kTailEntries]
    static constexpr size_t bytes = sl + (size_t)kTailEntries * 2;
};
static_assert(TailLds::bytes <= 160 * 1024, "LDS tail does not fit one CU");

template <bool RUNNING>
__global__ __launch_bounds__(kTailWidth) void tail_lds_kernel(SweepView S, const int* level_ptr,
                                                              const short* __restrict__ tslot, int l0, int l1,
                                                              int k0, int k1, int e0, int ne, double* x,
                                                              const int* done) {
    if (done && *done) return;
    extern __shared__ __align__(16) unsigned char tail_lds[];
    double* pv = reinterpret_cast<double*>(tail_lds + TailLds::pv);   // product (final dependency) or coefficient
    double* xt = reinterpret_cast<double*>(tail_lds + TailLds::xt);   // x of the run's unknowns, by run-local position
    double* dg = reinterpret_cast<double*>(tail_lds + TailLds::dg);
    int* rp = reinterpret_cast<int*>(tail_lds + TailLds::rp);         // run-local row extents
    int* lp = reinterpret_cast<int*>(tail_lds + TailLds::lp);         // run-local level extents
    short* sl = reinterpret_cast<short*>(tail_lds + TailLds::sl);
    const int nl = l1 - l0, nk = k1 - k0;
    for (int i = threadIdx.x; i <= nl; i += kTailWidth) lp[i] = level_ptr[l0 + i] - k0;
    // fill: every thread issues ALL its loads before it uses any of them (the fill is two round
    // trips -- coefficients/slots/indices, then the gathers of final x values -- not two per entry)
    {
        constexpr int EU = kTailEntries / kTailWidth, SU = kTailSlots / kTailWidth;
        static_assert(kTailEntries % kTailWidth == 0 && kTailSlots % kTailWidth == 0, "fill unroll");
        int sj[EU], jj[EU], rr[SU];
        double a[EU], xj[EU], xr[SU], dd[SU];
#pragma unroll
        for (int u = 0; u < EU; u++) {
            const int i = threadIdx.x + u * kTailWidth;
            const bool ok = i < ne;
            sj[u] = ok ? tslot[i] : 0;
            a[u] = ok ? S.val[e0 + i] : 0.0;
            jj[u] = ok ? S.idx[e0 + i] : 0;
        }
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int q = threadIdx.x + u * kTailWidth;
            rr[u] = q < nk ? S.order[k0 + q] : -1;
            dd[u] = q < nk ? S.diag[k0 + q] : 1.0;
        }
        for (int q = threadIdx.x; q <= nk; q += kTailWidth) rp[q] = S.ptr[k0 + q] - e0;
#pragma unroll
        for (int u = 0; u < EU; u++) xj[u] = (threadIdx.x + u * kTailWidth < ne && sj[u] < 0) ? x[jj[u]] : 1.0;
#pragma unroll
        for (int u = 0; u < SU; u++) xr[u] = rr[u] >= 0 ? x[rr[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < EU; u++) {
            const int i = threadIdx.x + u * kTailWidth;
            if (i < ne) {
                sl[i] = (short)sj[u];
                pv[i] = sj[u] < 0 ? (RUNNING ? a[u] * xj[u] : xj[u] * a[u]) : a[u];
            }
        }
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int q = threadIdx.x + u * kTailWidth;
            if (q < nk) {
                xt[q] = xr[u];                          // padding slots solve 0/1 and are never referenced
                dg[q] = rr[u] >= 0 ? dd[u] : 1.0;
            }
        }
    }
    __syncthreads();
    // Per level: (A) all threads turn the level's remaining coefficients into products, flat over
    // its entries; (B) one thread per unknown adds the row's products one at a time in storage
    // order straight from LDS -- those adds are the only serial work.
    for (int l = 0; l < nl; l++) {
        const int qb = lp[l], qe = lp[l + 1];
        for (int p = rp[qb] + threadIdx.x, pe = rp[qe]; p < pe; p += kTailWidth) {
            const int sj = sl[p];
            if (sj >= 0) pv[p] = RUNNING ? pv[p] * xt[sj] : xt[sj] * pv[p];
        }
        __syncthreads();
        for (int q = qb + threadIdx.x; q < qe; q += kTailWidth) {
            const int p1 = rp[q + 1];
            const double xr = xt[q];
            double acc = RUNNING ? xr : 0.0;
            int p = rp[q];
            for (; p + 4 <= p1; p += 4) {
                const double a0 = pv[p], a1 = pv[p + 1], a2 = pv[p + 2], a3 = pv[p + 3];
                if (RUNNING) { acc -= a0; acc -= a1; acc -= a2; acc -= a3; }
                else { acc += a0; acc += a1; acc += a2; acc += a3; }
            }
            for (; p < p1; p++) acc = RUNNING ? acc - pv[p] : acc + pv[p];
            xt[q] = (RUNNING ? acc : xr - acc) / dg[q];
        }
        __syncthreads();
    }
    {
        constexpr int SU = kTailSlots / kTailWidth;
        int rr[SU];
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int q = threadIdx.x + u * kTailWidth;
            rr[u] = q < nk ? S.order[k0 + q] : -1;
        }
#pragma unroll
        for (int u = 0; u < SU; u++)
            if (rr[u] >= 0) x[rr[u]] = xt[threadIdx.x + u * kTailWidth];
    }
}

// ---------------------------------------------------------------------------
// Synchronisation-free sweep: ONE launch per sweep instead of one per level.  Opt-in
// (IPXK_TRISOLVE=syncfree): on MI355X it measured SLOWER than one launch per level on the C3
// planted factors (4.2 vs 2.7 ms per basis CR iteration, round 1) -- a level hand-off through
// memory-side polling costs more than a kernel boundary here.  Kept because it is the single-
// launch form a hipGraph-free persistent variant would build on, and it is parity-tested.
// The result vector is single-assignment: it is pre-filled with a sentinel NaN and every unknown
// is written once with an agent-scope 8-byte store, so the value IS the ready flag (no separate
// flags, no fences; the hand-off form R2 of the CDNA guide).  Workgroups take chunks of the
// level-ordered positions from a ticket counter, hence every dependency of a position held by a
// running wave belongs to a lower ticket that some resident wave already owns: progress does not
// depend on dispatch order or placement.  Levels are padded to whole lane groups, so the lanes of
// one wavefront never wait for each other.  Every spin is bounded; a timeout raises `abort`.
// ---------------------------------------------------------------------------
constexpr unsigned long long kSentinel = 0x7FF8DEAD5EEDBEEFull;   // a quiet NaN nobody computes
constexpr int kSpinLimit = 1 << 20;

__global__ void fill_sentinel_kernel(int m, unsigned long long* __restrict__ x, int* ticket) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) x[i] = kSentinel;
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0;
}

// polls one dependency until it holds a value (bounded; raises `abort` on timeout)
__device__ __forceinline__ double wait_value(const unsigned long long* xo, int j, int* abort) {
    unsigned long long bits;
    int spins = 0;
    while ((bits = __hip_atomic_load(xo + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kSentinel) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > kSpinLimit || ((spins & 1023) == 0 &&
            __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    return __longlong_as_double((long long)bits);
}

// one position with GL lanes; GL == 1: the lane first issues the loads of ALL its (<= kShortRow)
// dependencies, then waits only for those still missing
template <bool RUNNING, int GL>
__device__ __forceinline__ void syncfree_unknown(const SweepView& S, int k, const double* __restrict__ xin,
                                                 double* xout, int* abort) {
    const int lane = threadIdx.x & 63, gl = lane & (GL - 1), gbase = lane & ~(GL - 1);
    const unsigned long long* xo = reinterpret_cast<const unsigned long long*>(xout);
    const int r = S.order[k];
    if (r < 0) return;                          // padding (whole wavefronts)
    const int p0 = S.ptr[k], p1 = S.ptr[k + 1];
    const double xr = xin[r];
    double acc = RUNNING ? xr : 0.0;
    if (GL == 1) {
        int j[kShortRow];
        double a[kShortRow];
        unsigned long long bits[kShortRow];
        const int len = p1 - p0;               // <= kShortRow in a "short" chunk
#pragma unroll
        for (int e = 0; e < kShortRow; e++) {
            j[e] = e < len ? S.idx[p0 + e] : 0;
            a[e] = e < len ? S.val[p0 + e] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < kShortRow; e++)
            bits[e] = e < len ? __hip_atomic_load(xo + j[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
        for (int e = 0; e < kShortRow; e++) {
            if (e < len) {
                const double xj = bits[e] == kSentinel ? wait_value(xo, j[e], abort)
                                                       : __longlong_as_double((long long)bits[e]);
                const double prod = RUNNING ? a[e] * xj : xj * a[e];
                acc = RUNNING ? acc - prod : acc + prod;
            }
        }
    } else {
        for (int base = p0; base < p1; base += GL) {
            const int p = base + gl;
            double prod = 0.0;
            if (p < p1) {
                const double xj = wait_value(xo, S.idx[p], abort);
                prod = RUNNING ? S.val[p] * xj : xj * S.val[p];
            }
            const int cnt = min(GL, p1 - base);
            for (int l = 0; l < cnt; l++) {
                const double t = __shfl(prod, gbase + l, 64);
                acc = RUNNING ? acc - t : acc + t;
            }
        }
    }
    const double res = (RUNNING ? acc : xr - acc) / S.diag[k];
    if (gl == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(xout) + r,
                           (unsigned long long)__double_as_longlong(res), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

template <bool RUNNING>
__global__ __launch_bounds__(kBlock) void syncfree_sweep_kernel(SweepView S, int npos,
                                                                const unsigned char* __restrict__ chunk_long,
                                                                const double* __restrict__ xin,
                                                                double* xout, int* ticket, int* abort,
                                                                const int* done) {
    if (done && *done) return;
    __shared__ int chunk_id;
    const int nchunks = (npos + kChunkRows - 1) / kChunkRows;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) chunk_id = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int chunk = chunk_id;
        if (chunk >= nchunks) break;
        const int kb = chunk * kChunkRows, ke = min(npos, kb + kChunkRows);
        if (chunk_long[chunk]) {
            for (int k = kb + threadIdx.x / 8; k < ke; k += kBlock / 8) syncfree_unknown<RUNNING, 8>(S, k, xin, xout, abort);
        } else {
            const int k = kb + threadIdx.x;
            if (k < ke) syncfree_unknown<RUNNING, 1>(S, k, xin, xout, abort);
        }
    }
}

// out[i] = in[perm[i]]
__global__ void gather_perm_kernel(int m, const double* __restrict__ in, const int* __restrict__ perm,
                                   double* __restrict__ out, const int* done) {
    if (done && *done) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
        out[i] = in[perm[i]];
}
// out[perm[i]] = in[i]
__global__ void scatter_perm_kernel(int m, const double* __restrict__ in, const int* __restrict__ perm,
                                    double* __restrict__ out, const int* done) {
    if (done && *done) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
        out[perm[i]] = in[i];
}

// lhs = free ? 0 : lhs + rhs;  partial dot rhs'lhs       (splitted_normal_matrix.cc:112-116)
__global__ __launch_bounds__(kBlock) void split_finish_kernel(int m, const double* __restrict__ rhs,
                                                              const unsigned char* __restrict__ free_mask,
                                                              double* __restrict__ lhs, double* partial,
                                                              const int* done) {
    if (done && *done) return;
    __shared__ double red[kBlock / 64 + 1];
    double acc = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) {
        const double r = rhs[i];
        const double l = free_mask[i] ? 0.0 : lhs[i] + r;
        lhs[i] = l;
        acc += r * l;
    }
    acc = block_reduce<SumOp>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------
// host: level analysis
// ---------------------------------------------------------------------------
// Generic builder.  Unknown i has the dependency list dep(i) = entries [rp[i], rp[i+1]) of
// (ri, rx) in the order in which they must be visited; `diag[i]` is its divisor.  Unknowns are
// processed in `ascending` or descending index order by the reference, which is a valid
// topological order of the dependencies.
// Host-side result of the level analysis of one sweep (pure CPU work: the four sweeps of a
// Prepare are analysed on four host threads, then uploaded one after the other).
constexpr int kAnalysisHelpers = 4;   // host threads per sweep for the row permutation

struct SweepHost {
    std::vector<int> order, ptr, idx, lptr;
    std::vector<short> tslot;
    std::vector<unsigned char> chunk_long;
    std::vector<double> val, valS, dg, dgS;
    bool has_scaled = false;
    // inputs of the analysis (rows in natural order) and its scratch; kept between Prepare calls so
    // that only the first one pays for allocating and faulting in a few hundred MB of host memory
    std::vector<int> rp, ri, level, next, posof;
    std::vector<double> rx, rxS, dgn, dgnS;
};

// host workspaces of split_prepare_host, owned by the context
struct PrepareHost {
    SweepHost Ut, Lt, Lf, Uf;
    std::vector<double> uscale;
    std::vector<unsigned char> fmask;
    std::vector<int> cnt;
};
void destroy_prepare_host(PrepareHost* p) { delete p; }

static void analyse_sweep(Sweep& S, SweepHost& H, int dim, bool ascending, bool running, const std::vector<int>& rp,
                        const std::vector<int>& ri, const std::vector<double>& rx,
                        const std::vector<double>& diag, const std::vector<double>* rxS,
                        const std::vector<double>* diagS) {
    S.dim = dim;
    S.running = running;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double ta0 = now();
    std::vector<int>& level = H.level;
    level.assign(dim, 0);
    int nlev = dim > 0 ? 1 : 0;
    for (int t = 0; t < dim; t++) {
        const int i = ascending ? t : dim - 1 - t;
        int lv = 0;
        for (int p = rp[i]; p < rp[i + 1]; p++) lv = std::max(lv, level[ri[p]] + 1);
        level[i] = lv;
        nlev = std::max(nlev, lv + 1);
    }
    S.nlevels = nlev;
    // counting sort of unknowns by level (stable in processing order); every level starts at a
    // multiple of `align` positions (pad slots have order -1) so that the lanes of one wavefront
    // never hold unknowns of two different levels (needed by the sync-free sweep)
    const int align = 64;
    std::vector<int> lcount(nlev, 0);
    std::vector<unsigned char> level_long(nlev, 0);
    for (int i = 0; i < dim; i++) {
        lcount[level[i]]++;
        if (rp[i + 1] - rp[i] > kShortRow) level_long[level[i]] = 1;
    }
    std::vector<int> lptr(nlev + 1, 0);
    for (int l = 0; l < nlev; l++) lptr[l + 1] = lptr[l] + (lcount[l] + align - 1) / align * align;
    const int npos = lptr[nlev];
    std::vector<int>&order = H.order, &next = H.next, &posof = H.posof;
    order.assign(std::max(npos, 1), -1);
    next.assign(lptr.begin(), lptr.end() - 1);
    posof.resize(dim);
    for (int t = 0; t < dim; t++) {
        const int i = ascending ? t : dim - 1 - t;
        posof[i] = next[level[i]];
        order[next[level[i]]++] = i;
    }
    const double ta1 = now();
    const size_t nz = ri.size();
    std::vector<int>&ptr = H.ptr, &idx = H.idx;
    std::vector<double>&val = H.val, &valS = H.valS, &dg = H.dg, &dgS = H.dgS;
    ptr.resize(npos + 1);
    idx.resize(std::max<size_t>(nz, 1));
    val.resize(std::max<size_t>(nz, 1));
    dg.assign(std::max(npos, 1), 1.0);
    if (rxS) { valS.resize(std::max<size_t>(nz, 1)); dgS.assign(std::max(npos, 1), 1.0); }
    else { valS.clear(); dgS.clear(); }
    {
        int put = 0;
        for (int k = 0; k < npos; k++) {
            const int i = order[k];
            ptr[k] = put;
            if (i >= 0) put += rp[i + 1] - rp[i];
        }
        ptr[npos] = put;
    }
    // rows into level order: independent per position (a random gather of rows, memory-latency bound
    // on the host), split over a few threads
    auto copy_rows = [&](int k0, int k1) {
        for (int k = k0; k < k1; k++) {
            const int i = order[k];
            if (i < 0) continue;
            int put = ptr[k];
            for (int p = rp[i]; p < rp[i + 1]; p++, put++) {
                idx[put] = ri[p];
                val[put] = rx[p];
                if (rxS) valS[put] = (*rxS)[p];
            }
            dg[k] = diag[i];
            if (rxS) dgS[k] = (*diagS)[i];
        }
    };
    {
        const int nt = npos >= (1 << 16) ? kAnalysisHelpers : 1;
        std::vector<std::thread> helpers;
        for (int t = 1; t < nt; t++)
            helpers.emplace_back(copy_rows, (int)((int64_t)npos * t / nt), (int)((int64_t)npos * (t + 1) / nt));
        copy_rows(0, (int)((int64_t)npos / nt));
        for (auto& h : helpers) h.join();
    }
    S.npos = npos;
    S.level_ptr = lptr;
    S.has_scaled = rxS != nullptr;
    H.has_scaled = rxS != nullptr;
    H.lptr = lptr;
    const double ta2 = now();
    // launch plan (plan_sweep) and the dependency-slot tables of its tail runs
    std::vector<int> lev_entry(nlev + 1);
    for (int lv = 0; lv <= nlev; lv++) lev_entry[lv] = H.ptr[lptr[lv]];
    const int ntslot = plan_sweep(S, lptr, level_long, lev_entry);
    H.tslot.assign((size_t)std::max(ntslot, 1), (short)-1);
    for (const Sweep::Launch& L : S.plan) {
        if (!L.tail) continue;
        const int k0 = lptr[L.l0], k1 = lptr[L.l1];
        for (int e = 0; e < L.ne; e++) {
            const int pj = posof[H.idx[L.e0 + e]];
            H.tslot[(size_t)L.tslot_off + e] = pj >= k0 && pj < k1 ? (short)(pj - k0) : (short)-1;
        }
    }
    if (getenv("IPXK_VERBOSE"))
        fprintf(stderr, "ipxk: analyse_sweep(%s,%s): levels+order %.1f ms, entry copy %.1f ms, plan %.1f ms\n",
                running ? "fwd" : "trans", ascending ? "asc" : "desc", (ta1 - ta0) * 1e3, (ta2 - ta1) * 1e3, (now() - ta2) * 1e3);
    if (getenv("IPXK_SWEEP_STATS")) {
        for (const Sweep::Launch& L : S.plan) {
            if (!L.tail && getenv("IPXK_SWEEP_STATS")[0] != '2') continue;   // "2": every launch
            int maxlen = 0; long ent = 0, unk = 0;
            for (int k = lptr[L.l0]; k < lptr[L.l1]; k++) {
                if (H.order[k] < 0) continue;
                unk++; ent += H.ptr[k + 1] - H.ptr[k]; maxlen = std::max(maxlen, H.ptr[k + 1] - H.ptr[k]);
            }
            fprintf(stderr, "sweep(%s,%s) %s levels %d..%d gl %d unknowns %ld entries %ld maxrow %d\n",
                    running ? "fwd" : "trans", ascending ? "asc" : "desc", L.tail ? "TAIL-LDS" : "level", L.l0, L.l1, L.gl, unk, ent, maxlen);
        }
    }
    H.chunk_long = sweep_chunk_flags(lptr, level_long);
}

// Launch plan.  Runs of >= kTailMinLevels narrow levels whose unknowns and entries fit the LDS of
// one CU go to ONE single-workgroup launch (tail_lds_kernel); every other level is a launch of
// its own, with one lane per unknown if all its rows have <= kShortRow entries and 8 lanes per
// unknown otherwise (the long rows towards the end of a forward sweep).
int plan_sweep(Sweep& S, const std::vector<int>& lptr, const std::vector<unsigned char>& level_long,
               const std::vector<int>& lev_entry) {
    const int nlev = (int)lptr.size() - 1;
    S.plan.clear();
    int ntslot = 0;
    auto fits = [&](int a, int b) {
        return b - a <= kTailLevelsLds && lptr[b] - lptr[a] <= kTailSlots &&
               lev_entry[b] - lev_entry[a] <= kTailEntries;
    };
    auto narrow = [&](int lv) { return lptr[lv + 1] - lptr[lv] <= kTailLevelWidth; };
    int l = 0;
    while (l < nlev) {
        if (narrow(l) && fits(l, l + 1)) {
            int b = l + 1;
            while (b < nlev && narrow(b) && fits(l, b + 1)) b++;
            if (b - l >= kTailMinLevels) {
                S.plan.push_back({l, b, true, 1, ntslot, lev_entry[l], lev_entry[b] - lev_entry[l]});
                ntslot += lev_entry[b] - lev_entry[l];
                l = b;
                continue;
            }
        }
        S.plan.push_back({l, l + 1, false, level_long[l] ? 8 : 1, 0, 0, 0});
        l++;
    }
    return ntslot;
}

// sync-free sweep: per chunk of kChunkRows positions, does it hold a long row?
std::vector<unsigned char> sweep_chunk_flags(const std::vector<int>& lptr, const std::vector<unsigned char>& level_long) {
    const int nlev = (int)lptr.size() - 1;
    const int npos = lptr[nlev];
    std::vector<unsigned char> flags((npos + kChunkRows - 1) / kChunkRows + 1, 0);
    for (int lv = 0; lv < nlev; lv++)
        if (level_long[lv] && lptr[lv + 1] > lptr[lv])
            for (int c = lptr[lv] / kChunkRows; c <= (lptr[lv + 1] - 1) / kChunkRows; c++) flags[c] = 1;
    return flags;
}

static void upload_sweep(Sweep& S, const SweepHost& H, hipStream_t s) {
    S.order.upload(H.order, s);
    S.ptr.upload(H.ptr, s);
    S.idx.upload(H.idx, s);
    S.val.upload(H.val, s);
    S.diag.upload(H.dg, s);
    if (H.has_scaled) { S.valS.upload(H.valS, s); S.diagS.upload(H.dgS, s); }
    S.level_ptr_dev.upload(H.lptr, s);
    S.chunk_long.upload(H.chunk_long, s);
    if (H.tslot.empty()) S.tslot.resize(1); else S.tslot.upload(H.tslot, s);
    IPXK_HIP(hipStreamSynchronize(s));
}

template <bool RUNNING>
static void launch_tail(Context* c, const Sweep& S, const SweepView& V, const Sweep::Launch& L, double* x,
                        const int* done) {
    static const bool configured = [] {
        IPXK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_lds_kernel<RUNNING>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)TailLds::bytes));
        return true;
    }();
    (void)configured;
    hipLaunchKernelGGL((tail_lds_kernel<RUNNING>), dim3(1), dim3(kTailWidth), TailLds::bytes, c->stream, V,
                       S.level_ptr_dev.get(), S.tslot.get() + L.tslot_off, L.l0, L.l1, S.level_ptr[L.l0],
                       S.level_ptr[L.l1], L.e0, L.ne, x, done);
}

template <bool RUNNING, int GL>
static void launch_level(Context* c, const Sweep& S, const SweepView& V, const Sweep::Launch& L, double* x,
                         const int* done) {
    const int k0 = S.level_ptr[L.l0], k1 = S.level_ptr[L.l1];
    const int g = (int)(((int64_t)(k1 - k0) * GL + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((level_kernel<RUNNING, GL>), dim3(g), dim3(kBlock), 0, c->stream, V, k0, k1, x, done);
}

static void run_sweep(Context* c, const Sweep& S, bool scaled, double* x, const int* done) {
    const SweepView V = S.view(scaled);
    for (const Sweep::Launch& L : S.plan) {
        if (S.running) {
            if (L.tail) launch_tail<true>(c, S, V, L, x, done);
            else if (L.gl == 8) launch_level<true, 8>(c, S, V, L, x, done);
            else launch_level<true, 1>(c, S, V, L, x, done);
        } else {
            if (L.tail) launch_tail<false>(c, S, V, L, x, done);
            else if (L.gl == 8) launch_level<false, 8>(c, S, V, L, x, done);
            else launch_level<false, 1>(c, S, V, L, x, done);
        }
    }
}

static void run_syncfree(Context* c, const Sweep& S, bool scaled, const double* xin, double* xout,
                         const int* done) {
    SplitOperator* sp = c->split;
    const SweepView V = S.view(scaled);
    const int m = S.dim;
    hipLaunchKernelGGL(fill_sentinel_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, c->stream, m,
                       reinterpret_cast<unsigned long long*>(xout), sp->ticket.get());
    const int nchunks = (S.npos + kChunkRows - 1) / kChunkRows;
    // few positions in flight beyond the active levels: waiting lanes poll memory
    int maxgrid = 512;
    if (const char* e = getenv("IPXK_SYNCFREE_GRID")) maxgrid = atoi(e) > 0 ? atoi(e) : maxgrid;
    const int grid = std::max(1, std::min(nchunks, maxgrid));
    if (S.running)
        hipLaunchKernelGGL(syncfree_sweep_kernel<true>, dim3(grid), dim3(kBlock), 0, c->stream, V, S.npos,
                           S.chunk_long.get(), xin, xout, sp->ticket.get(), sp->abort_flag.get(), done);
    else
        hipLaunchKernelGGL(syncfree_sweep_kernel<false>, dim3(grid), dim3(kBlock), 0, c->stream, V, S.npos,
                           S.chunk_long.get(), xin, xout, sp->ticket.get(), sp->abort_flag.get(), done);
}

// ForwardSolve: L then U (sparse_matrix.cc:303-306)
void forward_solve_dev(Context* c, double* x, bool scaled, const int* done) {
    SplitOperator* S = c->split;
    if (S->syncfree) {
        run_syncfree(c, S->Lf, scaled, x, S->wsf.get(), done);
        run_syncfree(c, S->Uf, scaled, S->wsf.get(), x, done);
        return;
    }
    run_sweep(c, S->Lf, scaled, x, done);
    run_sweep(c, S->Uf, scaled, x, done);
}
// BackwardSolve: U' then L' (sparse_matrix.cc:308-311)
void backward_solve_dev(Context* c, double* x, bool scaled, const int* done) {
    SplitOperator* S = c->split;
    if (S->syncfree) {
        run_syncfree(c, S->Ut, scaled, x, S->wsf.get(), done);
        run_syncfree(c, S->Lt, scaled, S->wsf.get(), x, done);
        return;
    }
    run_sweep(c, S->Ut, scaled, x, done);
    run_sweep(c, S->Lt, scaled, x, done);
}

// raises if a sync-free sweep timed out (host side, after the stream has been synchronized)
void check_sweep_abort(Context* c) {
    SplitOperator* S = c->split;
    if (!S || !S->syncfree) return;
    int flag = 0;
    S->abort_flag.download(&flag, 1, c->stream);
    IPXK_HIP(hipStreamSynchronize(c->stream));
    if (flag) {
        IPXK_HIP(hipMemsetAsync(S->abort_flag.get(), 0, sizeof(int), c->stream));
        throw Error(IPXK_E_HIP, "sync-free triangular sweep timed out waiting for a dependency");
    }
}

void split_levels(const Context* c, ipxint levels[4]) {
    levels[0] = c->split->Ut.nlevels;
    levels[1] = c->split->Lt.nlevels;
    levels[2] = c->split->Lf.nlevels;
    levels[3] = c->split->Uf.nlevels;
}

// ---------------------------------------------------------------------------
// Prepare
// ---------------------------------------------------------------------------
void split_prepare_host(Context* c, const ipxint* Lp, const ipxint* Li, const double* Lx,
                        const ipxint* Up, const ipxint* Ui, const double* Ux, const ipxint* rowperm,
                        const ipxint* colperm, const ipxint* basis, const ipxint* status,
                        const double* colscale) {
    const int m = (int)c->m, n = (int)c->n;
    hipStream_t s = c->stream;
    IPXK_REQUIRE(c->nranks == 1, "the basis path does not shard: run it as independent replicas");
    IPXK_REQUIRE(Lp[m] < (int64_t(1) << 31) && Up[m] < (int64_t(1) << 31), "factor nnz exceeds 32 bits");
    IPXK_REQUIRE(Lp[0] == 0 && Up[0] == 0, "column pointers must start at 0");
    for (int k = 0; k < m; k++) {
        IPXK_REQUIRE(Lp[k + 1] >= Lp[k] && Up[k + 1] > Up[k] && Up[k + 1] <= Up[m] && Lp[k + 1] <= Lp[m],
                     "factor column pointers not monotone");
        IPXK_REQUIRE(Ui[Up[k + 1] - 1] == k, "U must hold its diagonal last in each column");
        IPXK_REQUIRE(basis[k] >= 0 && basis[k] < n + m, "basis entry out of range");
        IPXK_REQUIRE(rowperm[k] >= 0 && rowperm[k] < m && colperm[k] >= 0 && colperm[k] < m, "permutation entry out of range");
    }
    {   // rowperm, colperm are permutations; basis entries distinct is the caller's contract
        std::vector<unsigned char> seen_r(m, 0), seen_c(m, 0);
        for (int k = 0; k < m; k++) {
            IPXK_REQUIRE(!seen_r[rowperm[k]] && !seen_c[colperm[k]], "rowperm / colperm is not a permutation");
            seen_r[rowperm[k]] = seen_c[colperm[k]] = 1;
        }
    }
    if (c->split) { destroy_split(c->split); c->split = nullptr; }
    std::unique_ptr<SplitOperator> S(new SplitOperator);
    S->m = m;

    if (!c->prepare_host) c->prepare_host = new PrepareHost;
    PrepareHost& P = *c->prepare_host;
    // column scaling of U (splitted_normal_matrix.cc:30-39): nothing for BASIC_FREE
    std::vector<double>& uscale = P.uscale;
    std::vector<unsigned char>& fmask = P.fmask;
    uscale.assign(m, 1.0);
    fmask.assign(m, 0);
    S->num_free = 0;
    for (int k = 0; k < m; k++) {
        const ipxint j = basis[colperm[k]];
        if (status[j] == IPXK_BASIC) uscale[k] = colscale[j];
        else if (status[j] == IPXK_BASIC_FREE) { fmask[k] = 1; S->num_free++; }   // :58-64
    }

    const bool verbose = getenv("IPXK_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = now();
    SweepHost &hUt = P.Ut, &hLt = P.Lt, &hLf = P.Lf, &hUf = P.Uf;
    // --- U' sweep: unknown k gathers the rows above the diagonal of column k, ascending
    auto job_Ut = [&] {
        SweepHost& H = hUt;
        H.rp.resize(m + 1); H.ri.resize(Up[m] - m);
        H.rx.resize(H.ri.size()); H.rxS.resize(H.ri.size()); H.dgn.resize(m); H.dgnS.resize(m);
        int put = 0;
        for (int k = 0; k < m; k++) {
            H.rp[k] = put;
            for (ipxint p = Up[k]; p < Up[k + 1] - 1; p++, put++) {
                H.ri[put] = (int)Ui[p];
                H.rx[put] = Ux[p];
                H.rxS[put] = Ux[p] * uscale[k];
            }
            H.dgn[k] = Ux[Up[k + 1] - 1];
            H.dgnS[k] = H.dgn[k] * uscale[k];
        }
        H.rp[m] = put;
        analyse_sweep(S->Ut, H, m, true, false, H.rp, H.ri, H.rx, H.dgn, &H.rxS, &H.dgnS);
    };
    // --- L' sweep: unknown k gathers column k of L (rows > k), descending, unit diagonal
    auto job_Lt = [&] {
        SweepHost& H = hLt;
        H.rp.resize(m + 1); H.ri.resize(Lp[m]); H.rx.resize(Lp[m]); H.dgn.assign(m, 1.0);
        for (int k = 0; k <= m; k++) H.rp[k] = (int)Lp[k];
        for (ipxint p = 0; p < Lp[m]; p++) { H.ri[p] = (int)Li[p]; H.rx[p] = Lx[p]; }
        analyse_sweep(S->Lt, H, m, false, false, H.rp, H.ri, H.rx, H.dgn, nullptr, nullptr);
    };
    // --- L sweep: unknown i subtracts L[i,j]*x_j for the columns j < i of row i, ascending j
    //     (the order in which the reference's column loop updates x[i], sparse_matrix.cc:283-297)
    auto job_Lf = [&] {
        SweepHost& H = hLf;
        H.rp.assign(m + 1, 0); H.ri.resize(Lp[m]); H.rx.resize(Lp[m]); H.dgn.assign(m, 1.0);
        for (ipxint p = 0; p < Lp[m]; p++) H.rp[Li[p] + 1]++;
        for (int i = 0; i < m; i++) H.rp[i + 1] += H.rp[i];
        H.next.assign(H.rp.begin(), H.rp.end() - 1);
        for (int j = 0; j < m; j++)
            for (ipxint p = Lp[j]; p < Lp[j + 1]; p++) {
                const int put = H.next[Li[p]]++;
                H.ri[put] = j;
                H.rx[put] = Lx[p];
            }
        analyse_sweep(S->Lf, H, m, true, true, H.rp, H.ri, H.rx, H.dgn, nullptr, nullptr);
    };
    // --- U sweep: unknown i subtracts U[i,j]*x_j for the columns j > i of row i, DESCENDING j
    //     (sparse_matrix.cc:267-281), then divides by U[i,i]
    auto job_Uf = [&] {
        SweepHost& H = hUf;
        H.rp.assign(m + 1, 0);
        for (int k = 0; k < m; k++)
            for (ipxint p = Up[k]; p < Up[k + 1] - 1; p++) H.rp[Ui[p] + 1]++;
        for (int i = 0; i < m; i++) H.rp[i + 1] += H.rp[i];
        H.ri.resize(H.rp[m]); H.rx.resize(H.rp[m]); H.rxS.resize(H.rp[m]); H.dgn.resize(m); H.dgnS.resize(m);
        H.next.assign(H.rp.begin(), H.rp.end() - 1);
        for (int k = m - 1; k >= 0; k--) {   // descending column order within each row
            for (ipxint p = Up[k]; p < Up[k + 1] - 1; p++) {
                const int put = H.next[Ui[p]]++;
                H.ri[put] = k;
                H.rx[put] = Ux[p];
                H.rxS[put] = Ux[p] * uscale[k];
            }
            H.dgn[k] = Ux[Up[k + 1] - 1];
            H.dgnS[k] = H.dgn[k] * uscale[k];
        }
        analyse_sweep(S->Uf, H, m, false, true, H.rp, H.ri, H.rx, H.dgn, &H.rxS, &H.dgnS);
    };
    // IPXK_PREPARE=host keeps the analysis on host threads (the form the device version is checked
    // against); default: on the device (prepare_device.hip)
    const bool on_device = !(getenv("IPXK_PREPARE") && std::string(getenv("IPXK_PREPARE")) == "host");
    double tp1 = tp0;
    if (on_device) {
        analyse_sweeps_device(c, S.get(), Lp, Li, Lx, Up, Ui, Ux, uscale);
        tp1 = now();
    } else {
        for (int k = 0; k < m; k++) {       // the device path checks the indices in a kernel
            for (ipxint p = Lp[k]; p < Lp[k + 1]; p++)
                IPXK_REQUIRE(Li[p] > k && Li[p] < m, "L must be strictly lower triangular with indices in range");
            for (ipxint p = Up[k]; p < Up[k + 1] - 1; p++)
                IPXK_REQUIRE(Ui[p] >= 0 && Ui[p] < k, "U must be upper triangular with indices in range");
        }
        // the analyses are independent and sequential each: one host thread per sweep
        std::exception_ptr err[3];
        auto guard = [&](int i, auto job) { return std::thread([&, i, job] { try { job(); } catch (...) { err[i] = std::current_exception(); } }); };
        std::thread t1 = guard(0, job_Lt), t2 = guard(1, job_Lf), t3 = guard(2, job_Uf);
        job_Ut();
        t1.join(); t2.join(); t3.join();
        for (auto& e : err) if (e) std::rethrow_exception(e);
        tp1 = now();
        upload_sweep(S->Ut, hUt, s);
        upload_sweep(S->Lt, hLt, s);
        upload_sweep(S->Lf, hLf, s);
        upload_sweep(S->Uf, hUf, s);
    }
    const double tp2 = now();
    // --- N N' weights: colscale^2 on NONBASIC columns (splitted_normal_matrix.cc:42-55)
    {
        std::vector<double> W((size_t)n + m, 0.0);
        for (int j = 0; j < n + m; j++)
            if (status[j] == IPXK_NONBASIC) W[j] = colscale[j] * colscale[j];
        S->Wsplit.upload(W, s);
    }
    // --- permutations (InversePerm, utils.cc:73-80) and bookkeeping for KKTSolverBasis::_Solve
    {
        std::vector<int> rpm(m), rpi(m), cpm(m), bs(m), stt((size_t)n + m);
        for (int i = 0; i < m; i++) { rpm[i] = (int)rowperm[i]; cpm[i] = (int)colperm[i]; bs[i] = (int)basis[i]; }
        for (int i = 0; i < m; i++) rpi[rpm[i]] = i;
        for (int j = 0; j < n + m; j++) stt[j] = (int)status[j];
        S->rowperm.upload(rpm, s);
        S->rowperm_inv.upload(rpi, s);
        S->colperm.upload(cpm, s);
        S->basis.upload(bs, s);
        S->status.upload(stt, s);
        S->colscale.upload(colscale, (size_t)n + m, s);
        S->free_mask.upload(fmask, s);
    }
    const size_t mm = (size_t)std::max(m, 1);
    S->w0.resize(mm); S->w1.resize(mm); S->w2.resize(mm); S->w3.resize(mm); S->tI.resize(mm);
    S->wsf.resize(mm);
    S->ticket.resize(1);
    S->abort_flag.resize(1);
    IPXK_HIP(hipMemsetAsync(S->abort_flag.get(), 0, sizeof(int), s));
    if (const char* e = getenv("IPXK_TRISOLVE")) S->syncfree = std::string(e) == "syncfree";
    if (c->partials.size() == 0) c->partials.resize((size_t)kNumPartialSlots * kPartialStride);
    IPXK_HIP(hipStreamSynchronize(s));
    if (verbose)
        fprintf(stderr, "ipxk: split_prepare [%s]: level analysis %.1f ms, upload of analysed factors %.1f ms, "
                        "weights/permutations %.1f ms\n", on_device ? "device" : "host",
                (tp1 - tp0) * 1e3, (tp2 - tp1) * 1e3, (now() - tp2) * 1e3);
    c->split = S.release();
}

// ---------------------------------------------------------------------------
// _Apply                                    (splitted_normal_matrix.cc:90-117)
// ---------------------------------------------------------------------------
int split_apply_dev(Context* c, const double* rhs, double* lhs, const int* done) {
    SplitOperator* S = c->split;
    const int m = S->m, n = (int)c->n;
    hipStream_t s = c->stream;
    const int g = vec_grid(m);
    double* work = S->w0.get();
    double* u = S->w1.get();
    // work = inverse(B') * rhs
    IPXK_HIP(hipMemcpyAsync(work, rhs, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    time_mark(c, kTimeBt, true);
    backward_solve_dev(c, work, true, done);
    time_mark(c, kTimeBt, false);
    time_mark(c, kTimeOp, true);
    // lhs = N N' work : un-permute, A (M D^2) A', permute
    hipLaunchKernelGGL(gather_perm_kernel, dim3(g), dim3(kBlock), 0, s, m, work, S->rowperm_inv.get(), u, done);
    EpiScale e1{{}, S->Wsplit.get(), c->tcols.get()};
    launch_spmv(c->Acols, u, e1, nullptr, done, s);
    EpiNormalRows e2{{}, S->Wsplit.get() + n, u, work};
    launch_spmv(c->Arows, c->tcols.get(), e2, nullptr, done, s);
    hipLaunchKernelGGL(gather_perm_kernel, dim3(g), dim3(kBlock), 0, s, m, work, S->rowperm.get(), lhs, done);
    time_mark(c, kTimeOp, false);
    // lhs = inverse(B) * lhs
    time_mark(c, kTimeB, true);
    forward_solve_dev(c, lhs, true, done);
    time_mark(c, kTimeB, false);
    // lhs += rhs; zero free positions; dot
    hipLaunchKernelGGL(split_finish_kernel, dim3(g), dim3(kBlock), 0, s, m, rhs, S->free_mask.get(), lhs,
                       c->part(kPartCdot), done);
    return g;
}

// Basis::SolveDense on the fresh, unscaled factors (forrest_tomlin.cc:67-78)
void solve_dense_dev(Context* c, const double* rhs, double* lhs, char trans) {
    SplitOperator* S = c->split;
    const int m = S->m;
    hipStream_t s = c->stream;
    const int g = vec_grid(m);
    double* work = S->w3.get();
    if (trans == 't' || trans == 'T') {
        hipLaunchKernelGGL(gather_perm_kernel, dim3(g), dim3(kBlock), 0, s, m, rhs, S->colperm.get(), work,
                           (const int*)nullptr);
        backward_solve_dev(c, work, false, nullptr);
        hipLaunchKernelGGL(scatter_perm_kernel, dim3(g), dim3(kBlock), 0, s, m, work, S->rowperm.get(), lhs,
                           (const int*)nullptr);
    } else {
        hipLaunchKernelGGL(gather_perm_kernel, dim3(g), dim3(kBlock), 0, s, m, rhs, S->rowperm.get(), work,
                           (const int*)nullptr);
        forward_solve_dev(c, work, false, nullptr);
        hipLaunchKernelGGL(scatter_perm_kernel, dim3(g), dim3(kBlock), 0, s, m, work, S->colperm.get(), lhs,
                           (const int*)nullptr);
    }
}

// ---------------------------------------------------------------------------
// KKTSolverBasis::_Solve                        (kkt_solver_basis.cc:75-194)
// ---------------------------------------------------------------------------
// work[p] = a[basis[p]] for BASIC_FREE positions, 0 otherwise            (:87-97)
__global__ void basis_free_rhs_kernel(int m, const int* __restrict__ basis, const int* __restrict__ status,
                                      const double* __restrict__ a, double* __restrict__ work) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < m; p += gridDim.x * blockDim.x) {
        const int j = basis[p];
        work[p] = status[j] == IPXK_BASIC_FREE ? a[j] : 0.0;
    }
}
// slack columns: tI[i] = W[n+i]*(a[n+i] - work[i])  (work == nullptr: W*a)   (:102-120, :178-188)
__global__ void basis_slack_kernel(int m, const double* __restrict__ WI, const double* __restrict__ aI,
                                   const double* __restrict__ work, double* __restrict__ tI) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const double s = WI[i];
        tI[i] = s != 0.0 ? (aI[i] - (work ? work[i] : 0.0)) * s : 0.0;
    }
}
// rhs[p] = (rhs[p]-work[p])/d + a[j]*d for BASIC, 0 for BASIC_FREE          (:128-138)
__global__ void basis_reduce_rhs_kernel(int m, const int* __restrict__ basis, const int* __restrict__ status,
                                        const double* __restrict__ colscale, const double* __restrict__ a,
                                        const double* __restrict__ work, double* __restrict__ rhs) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < m; p += gridDim.x * blockDim.x) {
        const int j = basis[p];
        if (status[j] == IPXK_BASIC) {
            const double d = colscale[j];
            rhs[p] = (rhs[p] - work[p]) / d + a[j] * d;
        } else {
            rhs[p] = 0.0;
        }
    }
}
// y[p] = y[p]/d for BASIC, a[j] for BASIC_FREE                               (:164-174)
__global__ void basis_unscale_y_kernel(int m, const int* __restrict__ basis, const int* __restrict__ status,
                                       const double* __restrict__ colscale, const double* __restrict__ a,
                                       double* __restrict__ y) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < m; p += gridDim.x * blockDim.x) {
        const int j = basis[p];
        y[p] = status[j] == IPXK_BASIC ? y[p] / colscale[j] : a[j];
    }
}
// x[basis[p]] = work[p]                                                      (:192-193)
__global__ void basis_scatter_x_kernel(int m, const int* __restrict__ basis, const double* __restrict__ work,
                                       double* __restrict__ x) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < m; p += gridDim.x * blockDim.x)
        x[basis[p]] = work[p];
}

// out[i] = acc + tI[i]  (acc = sum_j a_ij t_j starting from 0)
struct EpiBasisRhs : ProdMul {
    const double* tI; double* out;
    static constexpr bool kNeg = false;
    __device__ __forceinline__ double init(int) const { return 0.0; }
    __device__ __forceinline__ void finish(int i, double acc, double&) const { out[i] = acc + tI[i]; }
};
// out[i] = (b[i] - sum_j a_ij x_j) - tI[i]
struct EpiBasisResidual : ProdMul {
    const double* b; const double* tI; double* out;
    static constexpr bool kNeg = true;
    __device__ __forceinline__ double init(int i) const { return b[i]; }
    __device__ __forceinline__ void finish(int i, double acc, double&) const { out[i] = acc - tI[i]; }
};

CrResult kkt_basis_solve_dev(Context* c, const double* a, const double* b, double tol, ipxint maxiter,
                             double* x, double* y, ipxk_interrupt_fn interrupt, void* user,
                             ipxk_times* times) {
    SplitOperator* S = c->split;
    const int m = S->m, n = (int)c->n;
    hipStream_t s = c->stream;
    const int g = vec_grid(m);
    const double* W = S->Wsplit.get();
    double* rhs = S->w2.get();
    double* work = S->w1.get();     // note: split_apply_dev uses w0/w1 only inside the CR loop
    if (c->v_lhs.size() < (size_t)std::max(m, 1)) c->v_lhs.resize(std::max(m, 1));
    if (c->v_rhs.size() < (size_t)std::max(m, 1)) c->v_rhs.resize(std::max(m, 1));
    double* lhs = c->v_lhs.get();
    double* crrhs = c->v_rhs.get();

    // :87-99
    if (S->num_free > 0) {
        hipLaunchKernelGGL(basis_free_rhs_kernel, dim3(g), dim3(kBlock), 0, s, m, S->basis.get(),
                           S->status.get(), a, S->tI.get());
        solve_dense_dev(c, S->tI.get(), work, 'T');
    }
    const double* wk = S->num_free > 0 ? work : nullptr;
    // :101-121  rhs = sum over nonbasic j of AI[:,j] * d2_j*(a_j - AI[:,j]'work)
    if (wk) {
        EpiBasisColumns ec{{}, W, a, c->tcols.get()};
        launch_spmv(c->Acols, wk, ec, nullptr, nullptr, s);
    } else {
        // no free variables: alpha_j = d2_j * a_j
        hipLaunchKernelGGL(basis_slack_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, s, n, W, a,
                           (const double*)nullptr, c->tcols.get());
    }
    hipLaunchKernelGGL(basis_slack_kernel, dim3(g), dim3(kBlock), 0, s, m, W + n, a + n, wk, S->tI.get());
    {
        EpiBasisRhs er{{}, S->tI.get(), rhs};
        launch_spmv(c->Arows, c->tcols.get(), er, nullptr, nullptr, s);
    }
    solve_dense_dev(c, rhs, rhs, 'N');
    // :124
    solve_dense_dev(c, b, work, 'N');
    // :128-138
    hipLaunchKernelGGL(basis_reduce_rhs_kernel, dim3(g), dim3(kBlock), 0, s, m, S->basis.get(),
                       S->status.get(), S->colscale.get(), a, work, rhs);
    // :141-143
    hipLaunchKernelGGL(gather_perm_kernel, dim3(g), dim3(kBlock), 0, s, m, rhs, S->colperm.get(), crrhs,
                       (const int*)nullptr);
    // :146-157
    IPXK_HIP(hipMemsetAsync(lhs, 0, sizeof(double) * m, s));
    CrResult res = cr_solve_dev(c, crrhs, tol, nullptr, maxiter, lhs, true, interrupt, user, nullptr, 0, times);
    // :160-161
    hipLaunchKernelGGL(scatter_perm_kernel, dim3(g), dim3(kBlock), 0, s, m, lhs, S->colperm.get(), y,
                       (const int*)nullptr);
    // :164-175
    hipLaunchKernelGGL(basis_unscale_y_kernel, dim3(g), dim3(kBlock), 0, s, m, S->basis.get(),
                       S->status.get(), S->colscale.get(), a, y);
    solve_dense_dev(c, y, y, 'T');
    // :178-188  x[nonbasic] and work = b - N*x[nonbasic]
    {
        EpiBasisColumns ec{{}, W, a, x};
        launch_spmv(c->Acols, y, ec, nullptr, nullptr, s);
        hipLaunchKernelGGL(basis_slack_kernel, dim3(g), dim3(kBlock), 0, s, m, W + n, a + n, (const double*)y,
                           x + n);
        EpiBasisResidual er{{}, b, x + n, work};
        launch_spmv(c->Arows, x, er, nullptr, nullptr, s);
    }
    // :191-193
    solve_dense_dev(c, work, work, 'N');
    hipLaunchKernelGGL(basis_scatter_x_kernel, dim3(g), dim3(kBlock), 0, s, m, S->basis.get(), work, x);
    IPXK_HIP(hipGetLastError());
    return res;
}

}  // namespace ipxk
