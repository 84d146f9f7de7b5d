// Basis-preconditioned operator on the device.
//   SplittedNormalMatrix::Prepare / _Apply      reference src/splitted_normal_matrix.cc:18-117
//   TriangularSolve / ForwardSolve / BackwardSolve   src/sparse_matrix.cc:224-311
//   Basis::SolveDense on fresh factors               src/basis.cc:168-170, src/forrest_tomlin.cc:67-78
//
// Triangular solves are level-scheduled gather sweeps WITHOUT a launch or a barrier per level.  For
// each of the four sweeps (U', L', L, U) Prepare computes the dependency level of every unknown and
// packs the rows level by level into chunks of one wavefront's work (prepare_device.hip).  A sweep
// is out of place and single-assignment: the result vector is pre-filled with a sentinel, every
// unknown is stored exactly once with one 8-byte store, and a consumer polls the values it needs
// with L1-bypassing loads until they differ from the sentinel -- the value IS the ready flag, no
// flags, no fences, no atomics.  Wavefront w of W resident wavefronts owns chunks w, w+W, w+2W, ...
// of the level-ordered chunk sequence; a chunk depends only on chunks before it, so the lowest
// unfinished chunk can always proceed: no deadlock whatever the dispatch order or placement.  While
// a wavefront waits for the dependencies of its chunk, the records of its next chunk are already in
// flight, so a level costs about one store-to-load hand-off (~1 us chip-wide) instead of a kernel
// boundary plus three dependent round trips (~6 us).  Runs of narrow levels are confined to the
// workgroups of ONE XCD, whose L2 then carries the hand-off (~0.6 us per level); see
// sweep_run_kernel for how that stays independent of the actual placement.
// Every row is summed in the reference's order:
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
#include <string>

#include "context.hpp"
#include "spmv_kernels.hpp"
#include "trisolve.hpp"

namespace ipxk {

void destroy_split(SplitOperator* s) { delete s; }

static void bump_between(Context* c, bool trans, double* y, const int* done);   // dense bump of an LU from the device (below)

static int vec_grid(int64_t len) {
    int64_t g = (len + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return (int)(g < 1024 ? g : 1024);
}

// ---------------------------------------------------------------------------
// sweep kernel
// ---------------------------------------------------------------------------
using gu64 = unsigned long long;
constexpr gu64 kSentinel = 0x7FF8DEAD5EEDBEEFull;   // a quiet NaN nobody computes
constexpr gu64 kPlainNan = 0x7FF8000000000000ull;
constexpr int kSpinLimit = 1 << 22;                 // polls (>= 0.2 us each) before a wave gives up

constexpr int kSweepGrid = 256;      // workgroups of an all-XCD run (one per CU: all resident)
constexpr int kSweepXcdWgs = 32;     // participating workgroups of a one-XCD run (one per CU of an XCD)
constexpr int kNarrowLevel = 96;     // levels of up to this many chunks may join a one-XCD run
constexpr int kMinXcdLevels = 10;    // shorter runs of narrow levels are not worth a launch of their own (4 until round 3: with the
                                     // inverted blocks below, the 6-10 level runs left next to them cost 5-10 us more than they saved)

__device__ __forceinline__ gu64 load_sc1(const gu64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // bypasses L1, served by L2 / fabric
}

// What one lane needs for its part of a chunk; loaded one chunk ahead of use.
struct LaneRec {
    int src;        // index of the right-hand side in the input vector (-1: padding)
    int len;        // entries of the row
    int sub;        // merged chunks: which of the chunk's levels the row belongs to
    int dst2;       // second destination of the result (SweepView::dst2), -1: none
    double dg, xr;
    int j[8];       // dependency positions
    double a[8];
};

// long chunks: step of the row at which the FIRST round of 8 steps starts (0, or the last 8 steps when the
// rounds are taken from the end of the row)
__device__ __forceinline__ int first_round_step(const SweepView& S, const ChunkDesc& d) {
    return (d.width < -8 && S.newest_first && d.sub <= 1) ? -d.width - 8 : 0;
}

__device__ __forceinline__ void load_rec(LaneRec& R, const SweepView& S, const ChunkDesc& d, int lane,
                                         const double* __restrict__ xin) {
    // every address depends on the (scalar) descriptor only: one round trip, fully coalesced
    const int pos = d.width >= 0 ? d.pos0 + lane : d.pos0 + (lane >> 3);
    R.src = S.src[pos];
    R.dst2 = S.dst2 ? S.dst2[pos] : -1;
    R.dg = S.diag[pos];
    const int lenword = S.len[pos];
    R.len = lenword & ((1 << kLenBits) - 1);
    R.sub = lenword >> kLenBits;
    const int steps = d.width >= 0 ? d.width : min(-d.width, 8);     // wave-uniform
    const int ent = d.ent0 + first_round_step(S, d) * 64;            // scalar
#pragma unroll
    for (int e = 0; e < 8; e++) {
        R.j[e] = 0; R.a[e] = 0.0;
        if (e < steps) {
            R.j[e] = S.idx[ent + e * 64 + lane];
            R.a[e] = S.val[ent + e * 64 + lane];
        }
    }
    R.xr = R.src >= 0 ? xin[R.src] : 0.0;
}

// ---- how results travel from the wavefront that computes them to the wavefronts that need them ----
// (the value is the flag in every case: a slot holds the sentinel until its one and only store)
// Through memory: every look at a dependency bypasses L1; results are stored write-through, or with plain
// stores that stay in the XCD's L2 when all workgroups of the launch are known to share one XCD.
struct HandGlobal {
    const gu64* xo; double* xout; bool plain_store; double* out2;
    __device__ __forceinline__ gu64 look(int pj) const { return load_sc1(xo + pj); }
    __device__ __forceinline__ void store(int pos, gu64 out) const {
        if (plain_store) __hip_atomic_store(reinterpret_cast<gu64*>(xout) + pos, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(reinterpret_cast<gu64*>(xout) + pos, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
// first look at the dependencies of the lane's (up to 8) entries starting at entry `first`; dependencies at
// positions [lo, hi) belong to the (merged) chunk itself and travel through lane shuffles instead
template <class Hand>
__device__ __forceinline__ void issue_polls(const LaneRec& R, bool ell, int gl, int first, const Hand& H, gu64 (&bits)[8],
                                            int lo = 0, int hi = 0) {
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int e = ell ? t : first + t * 8 + gl;
        bits[t] = 0ull;
        if (R.src >= 0 && e >= 0 && e < R.len && !(R.j[t] >= lo && R.j[t] < hi)) bits[t] = H.look(R.j[t]);
    }
}

// polls until every dependency holds a value; false on timeout (abort raised)
template <class Hand>
__device__ __forceinline__ bool wait_polls(const LaneRec& R, const Hand& H, gu64 (&bits)[8], int* abort_flag) {
    int spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int t = 0; t < 8; t++) ok &= bits[t] != kSentinel;
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int t = 0; t < 8; t++)
            if (bits[t] == kSentinel) bits[t] = H.look(R.j[t]);
        if (++spins > kSpinLimit ||
            ((spins & 255) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
}

template <class Hand>
__device__ __forceinline__ void store_result(const Hand& H, int pos, double res, int dst2 = -1) {
    gu64 out = (gu64)__double_as_longlong(res);
    if (out == kSentinel) out = kPlainNan;     // a result must never look unfinished
    H.store(pos, out);
    if (dst2 >= 0) H.out2[dst2] = res;         // second copy for the kernel AFTER this launch: a plain store
}

// value of lane (this lane + N) of the same 16-lane row (DPP row_shl:N); lanes whose source lies outside
// the row keep their own value
template <int N>
__device__ __forceinline__ double row_shift_left(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x100 + N, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x100 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// acc -= / += the products held by the first `cnt` lanes of the 8-lane group, one at a time in lane
// order.  Only the group's FIRST lane ends up with the sum: it fetches its neighbours' products with DPP
// row shifts (plain VALU moves, no LDS crossbar as a shuffle would need; an aligned group of 8 never
// leaves its 16-lane row), then adds them in order.
template <bool RUNNING>
__device__ __forceinline__ double ordered_combine(double acc, double prod, int cnt) {
    double v[kLongLanes];
    v[0] = prod;
    v[1] = row_shift_left<1>(prod); v[2] = row_shift_left<2>(prod); v[3] = row_shift_left<3>(prod);
    v[4] = row_shift_left<4>(prod); v[5] = row_shift_left<5>(prod); v[6] = row_shift_left<6>(prod);
    v[7] = row_shift_left<7>(prod);
#pragma unroll
    for (int t = 0; t < kLongLanes; t++)
        if (t < cnt) acc = RUNNING ? acc - v[t] : acc + v[t];
    return acc;
}

// A MERGED chunk (trisolve.hpp): `nsub` consecutive tiny levels in one chunk.  The dependencies outside the
// chunk have been polled (bits); the wavefront runs the levels in order, every lane recomputes its row in every
// round and keeps the value of the round that is its own level -- by then all its dependencies inside the chunk
// (rows of earlier levels: other lanes of this wavefront) hold their results, which travel by lane shuffles.
// Each row is still summed in its own order: bit-identical to the unmerged form.
template <bool RUNNING, class Hand>
__device__ __forceinline__ void solve_merged(const LaneRec& R, const ChunkDesc& d, int lane, const Hand& H, const gu64 (&bits)[8]) {
    const bool ell = d.width >= 0;
    const int gl = lane & 7;
    const int npos = ell ? 64 : kLongLanes;
    const int len = R.src >= 0 ? R.len : 0;
    double result = 0.0;
    const int steps = ell ? d.width : min(-d.width, 8);            // wave-uniform: entries (ELL) / steps of 8 (long rows)
    // where each dependency comes from: a lane of this wavefront (rows of earlier levels of the chunk) or memory
    int from[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int e = ell ? t : t * 8 + gl;
        const int off = R.j[t] - d.pos0;
        from[t] = (t < steps && e < len && off >= 0 && off < npos) ? (ell ? off : off << 3) : -1;
    }
    for (int s = 0; s < d.sub; s++) {
        // all shuffles of the round are issued before the first use, so their latencies overlap
        double xin[8];
#pragma unroll
        for (int t = 0; t < 8; t++)
            if (t < steps) xin[t] = __shfl(result, from[t] >= 0 ? from[t] : lane, 64);
        double acc = RUNNING ? R.xr : 0.0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (t >= steps) break;                                  // wave-uniform
            const int e = ell ? t : t * 8 + gl;
            double prod = 0.0;
            if (e < len) {
                const double xj = from[t] >= 0 ? xin[t] : __longlong_as_double((long long)bits[t]);
                prod = RUNNING ? R.a[t] * xj : xj * R.a[t];
            }
            if (ell) { if (e < len) acc = RUNNING ? acc - prod : acc + prod; }
            else acc = ordered_combine<RUNNING>(acc, prod, min(kLongLanes, len - t * 8));
        }
        if (R.sub == s && (ell || gl == 0)) result = R.src >= 0 ? (RUNNING ? acc : R.xr - acc) / R.dg : 0.0;
    }
    if (ell) store_result(H, d.pos0 + lane, result, R.dst2);
    else if (gl == 0) store_result(H, d.pos0 + (lane >> 3), result, R.dst2);
}

// solves the chunk whose records are in R (first look at the dependencies already issued into bits);
// false on timeout
template <bool RUNNING, bool MERGED, class Hand>
__device__ __forceinline__ bool solve_chunk(LaneRec& R, const ChunkDesc& d, int lane, const SweepView& S, const Hand& H,
                                            gu64 (&bits)[8], int* abort_flag) {
    const bool ell = d.width >= 0;
    if (!wait_polls(R, H, bits, abort_flag)) return false;
    if (MERGED && d.sub > 1) { solve_merged<RUNNING>(R, d, lane, H, bits); return true; }
    if (ell) {
        double acc = RUNNING ? R.xr : 0.0;
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (e < R.len) {
                const double xj = __longlong_as_double((long long)bits[e]);
                const double prod = RUNNING ? R.a[e] * xj : xj * R.a[e];
                acc = RUNNING ? acc - prod : acc + prod;
            }
        // padding positions get a value too (1 wavefront = 1 contiguous store; nobody depends on them)
        store_result(H, d.pos0 + lane, R.src >= 0 ? (RUNNING ? acc : R.xr - acc) / R.dg : 0.0, R.dst2);
        return true;
    }
    const int gl = lane & 7;
    const int len = R.src >= 0 ? R.len : 0;
    double acc = RUNNING ? R.xr : 0.0;
    if (first_round_step(S, d) == 0) {
        for (int first = 0;;) {          // 64 entries of the row per round (one round unless the row is longer)
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int e0 = first + t * 8;                   // first entry of this step of the group
                if (!__any(e0 < len)) break;                    // wave-uniform
                double prod = 0.0;
                if (e0 + gl < len) {
                    const double xj = __longlong_as_double((long long)bits[t]);
                    prod = RUNNING ? R.a[t] * xj : xj * R.a[t];
                }
                acc = ordered_combine<RUNNING>(acc, prod, min(kLongLanes, len - e0));
            }
            first += 64;
            if (!__any(first < len)) break;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int step = first / 8 + t;                 // wave-uniform
                R.j[t] = 0; R.a[t] = 0.0;
                if (step < -d.width) { R.j[t] = S.idx[d.ent0 + step * 64 + lane]; R.a[t] = S.val[d.ent0 + step * 64 + lane]; }
            }
            issue_polls(R, false, gl, first, H, bits);
            if (!wait_polls(R, H, bits, abort_flag)) return false;
        }
    } else {
        // rounds from the END of the row (SweepView::newest_first, rows of more than 64 entries): all rounds but
        // the last one wait for unknowns that were solved long ago
        const int nsteps = -d.width;
        for (int base = nsteps - 8;;) {
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int e0 = (base + t) * 8;
                if (e0 < 0 || !__any(e0 < len)) continue;       // wave-uniform
                double prod = 0.0;
                if (e0 + gl < len) {
                    const double xj = __longlong_as_double((long long)bits[t]);
                    prod = RUNNING ? R.a[t] * xj : xj * R.a[t];
                }
                acc = ordered_combine<RUNNING>(acc, prod, min(kLongLanes, len - e0));
            }
            base -= 8;
            if (base <= -8) break;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int step = base + t;                      // wave-uniform
                R.j[t] = 0; R.a[t] = 0.0;
                if (step >= 0) { R.j[t] = S.idx[d.ent0 + step * 64 + lane]; R.a[t] = S.val[d.ent0 + step * 64 + lane]; }
            }
            issue_polls(R, false, gl, base * 8, H, bits);
            if (!wait_polls(R, H, bits, abort_flag)) return false;
        }
    }
    if (gl == 0) store_result(H, d.pos0 + (lane >> 3), R.src >= 0 ? (RUNNING ? acc : R.xr - acc) / R.dg : 0.0, R.dst2);
    return true;
}

__device__ __forceinline__ ChunkDesc scalar_desc(const ChunkDesc& v) {   // wave-uniform (scalar) values
    ChunkDesc d;
    d.pos0 = __builtin_amdgcn_readfirstlane(v.pos0);
    d.ent0 = __builtin_amdgcn_readfirstlane(v.ent0);
    d.width = __builtin_amdgcn_readfirstlane(v.width);
    d.npos = v.npos;
    d.sub = __builtin_amdgcn_readfirstlane(v.sub);
    return d;
}

// the wavefront's chunks c, c + NW, ... < c1; A holds the records of chunk c (descriptor d), dn is the
// descriptor of chunk c + NW
// (Round 3, measured and dropped: THREE chunks in flight per wavefront -- the first look at the dependencies of chunk
// c + NW and its right-hand side issued before chunk c waits, the records of chunk c + 2 NW behind them.  The wide
// levels move 11 G unknowns/s = 6.5 us per chunk and wavefront, which looks like a lack of overlapped round trips;
// but the deeper pipeline made both pairs slower, backward 272 -> 288 us, forward 300 -> 314 us.)
template <bool RUNNING, bool MERGED, class Hand>
__device__ __forceinline__ void chunk_loop(const SweepView& S, int c, int c1, int NW, int lane, const double* __restrict__ xin,
                                           const Hand& H, LaneRec& A, ChunkDesc d, ChunkDesc dn, int* abort_flag) {
    LaneRec B;
    for (;;) {
        gu64 bits[8];
        const int own = MERGED && d.sub > 1 ? (d.width >= 0 ? 64 : kLongLanes) : 0;     // merged: the chunk's own positions
        issue_polls(A, d.width >= 0, lane & 7, d.width >= 0 ? 0 : first_round_step(S, d) * 8, H, bits, d.pos0, d.pos0 + own);
        // the next chunk's records (and the descriptor after that) travel while this chunk waits
        const int cn = c + NW;
        ChunkDesc dnn = dn;
        if (cn < c1) {
            load_rec(B, S, dn, lane, xin);
            if (cn + NW < c1) dnn = scalar_desc(S.chunks[cn + NW]);
        }
        if (!solve_chunk<RUNNING, MERGED>(A, d, lane, S, H, bits, abort_flag)) return;
        if (cn >= c1) return;
        A = B; d = dn; dn = dnn; c = cn;
    }
}

// One run of consecutive levels = chunks [c0, c1) of a sweep.
// xcd_mode == 0: every workgroup takes part; results are stored write-through.
// xcd_mode == 1: only the workgroups with blockIdx % 8 == 0 take part -- under the round-robin
//   dispatch of gfx950 they share one XCD, whose L2 then serves the polls, and results are stored
//   with plain stores that stay in that L2.  That placement is an observation, not a contract, so
//   it is CHECKED: every participant publishes the id of the XCD it runs on (HW_REG_XCC_ID) and
//   reads everybody else's; only if all agree are plain stores used, otherwise every participant
//   falls back to write-through stores (all participants see the same ids and decide alike).
//   Correctness therefore never depends on where workgroups land, only the speed does.
// MERGED: the run contains merged chunks (the lean instantiation without that path serves all other runs)
template <bool RUNNING, bool MERGED>
__global__ __launch_bounds__(kBlock) void sweep_run_kernel(SweepView S, int c0, int c1, const double* __restrict__ xin,
                                                           double* xout, int xcd_mode, unsigned epoch, gu64* xcc_slots,
                                                           int* abort_flag, const int* done) {
    if (xcd_mode && (blockIdx.x & 7)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = xcd_mode ? blockIdx.x >> 3 : blockIdx.x;
    const int nparts = xcd_mode ? (gridDim.x + 7) >> 3 : gridDim.x;
    const int gw = part * (kBlock / 64) + wave, NW = nparts * (kBlock / 64);
    const gu64* xo = reinterpret_cast<const gu64*>(xout);
    int c = c0 + gw;
    const bool active = c < c1;
    // the first two descriptors travel while the `done` flag is read
    ChunkDesc raw = S.chunks[active ? c : c0], rawn = S.chunks[c + NW < c1 ? c + NW : c0];
    if (done && *done) return;
    unsigned xcc = 0;
    if (xcd_mode && wave == 0) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xff;
        if (lane == 0) __hip_atomic_store(xcc_slots + part, ((gu64)epoch << 32) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ChunkDesc d = scalar_desc(raw), dn = scalar_desc(rawn);
    LaneRec A;
    if (active) load_rec(A, S, d, lane, xin);            // in flight during the placement check
    bool plain = false;
    if (xcd_mode) {
        __shared__ int same_xcd;
        if (wave == 0) {
            bool same = true;
            for (int i = lane; i < nparts; i += 64) {
                gu64 v;
                int spins = 0;
                while (((v = load_sc1(xcc_slots + i)) >> 32) != epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kSpinLimit) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                same &= (unsigned)(v & 0xff) == xcc && (v >> 32) == epoch;
            }
            same = __all(same);
            if (lane == 0) same_xcd = same ? 1 : 0;
        }
        __syncthreads();
        plain = same_xcd != 0;
    }
    if (!active) return;
    const HandGlobal H{xo, xout, plain, S.out2};
    chunk_loop<RUNNING, MERGED>(S, c, c1, NW, lane, xin, H, A, d, dn, abort_flag);
}

// pre-fills the result vectors of up to four sweeps with the sentinel
struct FillList { gu64* p[4]; int n[4]; };
__global__ void fill_sentinel_kernel(FillList L, const int* done) {
    if (done && *done) return;
#pragma unroll
    for (int k = 0; k < 4; k++)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L.n[k]; i += gridDim.x * blockDim.x) L.p[k][i] = kSentinel;
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
// a sweep's result (by position) into index order: out[perm ? perm[k] : k] = y[posof[k]]
__global__ void unpack_result_kernel(int m, const double* __restrict__ y, const int* __restrict__ posof,
                                     const int* __restrict__ perm, double* __restrict__ out) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x)
        out[perm ? perm[k] : k] = y[posof[k]];
}
// out[p] = order[p] >= 0 ? map[order[p]] : -1   (map == nullptr: identity)
__global__ void compose_kernel(int n, const int* __restrict__ order, const int* __restrict__ map, int* __restrict__ out) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const int i = order[p];
        out[p] = i >= 0 ? (map ? map[i] : i) : -1;
    }
}

// out[map[i]] = i
__global__ void invert_map_kernel(int n, const int* __restrict__ map, int* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[map[i]] = i;
}

// lhs = free ? 0 : lhs + rhs;  partial dot rhs'lhs       (splitted_normal_matrix.cc:112-116)
__global__ __launch_bounds__(kBlock) void split_finish_kernel(int m, const double* __restrict__ rhs,
                                                              const unsigned char* __restrict__ free_mask,
                                                              const double* __restrict__ y, const int* __restrict__ posof,
                                                              double* __restrict__ lhs, double* partial,
                                                              const int* done) {
    if (done && *done) return;
    __shared__ double red[kBlock / 64 + 1];
    double acc = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) {
        const double r = rhs[i];
        const double l = free_mask[i] ? 0.0 : y[posof[i]] + r;
        lhs[i] = l;
        acc += r * l;
    }
    acc = block_reduce<SumOp>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------
// inverted head / tail of a sweep (Sweep::Block, trisolve.hpp)
// ---------------------------------------------------------------------------
constexpr int kBlockInvThreads = 1024;
// slot of entry e of a row whose first entry sits at `base` (base < 0: a row of a long chunk, -(slot + 1))
__device__ __forceinline__ int row_slot(int base, int e) {
    return base >= 0 ? base + e * 64 : (-base - 1) + (e >> 3) * 64 + (e & 7);
}
// per block row: how many of its entries look at positions in front of the block (< p0) / inside it
__global__ void block_count_kernel(int K, int p0, const int* __restrict__ tpos, const int* __restrict__ base, const int* __restrict__ len,
                                   const int* __restrict__ idx, int* __restrict__ hcnt, int* __restrict__ tcnt) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < K; t += gridDim.x * blockDim.x) {
        const int L = len[tpos[t]] & ((1 << kLenBits) - 1), b = base[t];
        int h = 0;
        for (int e = 0; e < L; e++) h += idx[row_slot(b, e)] < p0 ? 1 : 0;
        hcnt[t] = h;
        tcnt[t] = L - h;
    }
}
// outside entries -> their slots (row order kept); inside entries -> (block rank of the dependency, unscaled value)
__global__ void block_fill_kernel(int K, int p0, const int* __restrict__ tpos, const int* __restrict__ base, const int* __restrict__ len,
                                  const int* __restrict__ idx, const double* __restrict__ val, const int* __restrict__ rank_of_pos,
                                  const int* __restrict__ hptr, const int* __restrict__ tptr, int* __restrict__ hslot,
                                  int* __restrict__ hidx, int* __restrict__ tcol, double* __restrict__ tval) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < K; t += gridDim.x * blockDim.x) {
        const int L = len[tpos[t]] & ((1 << kLenBits) - 1), b = base[t];
        int h = hptr[t], q = tptr[t];
        for (int e = 0; e < L; e++) {
            const int slot = row_slot(b, e), j = idx[slot];
            if (j < p0) { hslot[h] = slot; hidx[h] = j; h++; }
            else { tcol[q] = rank_of_pos[j - p0]; tval[q] = val[slot]; q++; }
        }
    }
}
// M = inverse(T22), T22 = diag + the inside entries, lower triangular in block order.  A workgroup owns 64 columns of M
// (lane = column) and walks the block's levels; the rows of a level are independent and shared among the wavefronts.
// Row i of the columns j0.. needs the rows k < i of the SAME columns: written by this workgroup in earlier levels.
__global__ __launch_bounds__(kBlockInvThreads) void block_inverse_kernel(int K, int nlev, const int* __restrict__ lev, const int* __restrict__ tptr,
                                                                         const int* __restrict__ tcol, const double* __restrict__ tval,
                                                                         const double* __restrict__ dg, const int* __restrict__ tpos,
                                                                         double* M) {
    const int j0 = blockIdx.x * 64, j = j0 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
    for (int l = 0; l < nlev; l++) {
        const int r1 = lev[l + 1];
        for (int i = lev[l] + wave; i < r1; i += kBlockInvThreads / 64) {
            if (i < j0) continue;                                  // rows above the block's first column: zero (M is pre-filled)
            double s2 = i == j ? 1.0 : 0.0;
            for (int e = tptr[i]; e < tptr[i + 1]; e++) {
                const int k = tcol[e];
                if (k >= j0 && j < K) s2 -= tval[e] * M[(size_t)k * K + j];
            }
            if (j < K) M[(size_t)i * K + j] = s2 / dg[tpos[i]];
        }
        __syncthreads();                                           // (workgroup-scope release / acquire of the rows just written)
    }
}
// right-hand side of block unknown l as the block's kernels read it: xin[zsrc[l]]
__global__ void block_zsrc_kernel(int K, const int* __restrict__ tpos, const int* __restrict__ src, int* __restrict__ zsrc) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < K; t += gridDim.x * blockDim.x) zsrc[t] = src[tpos[t]];
}
// z[t] = (rhs of block row t minus its outside entries, all final: the launches of the earlier levels are over)
// [/ the column scale of unknown t: scaled U' sweep], 32 lanes per row
__global__ __launch_bounds__(kBlock) void block_gather_kernel(SweepView S, int K, const int* __restrict__ zsrc, const int* __restrict__ hptr,
                                                              const int* __restrict__ hslot, const int* __restrict__ hidx,
                                                              const int* __restrict__ unk, const double* __restrict__ pre_scale,
                                                              const double* __restrict__ xin, const double* __restrict__ y,
                                                              double* __restrict__ z, const int* done) {
    if (done && *done) return;
    const int g = threadIdx.x & 31;
    for (int t = (blockIdx.x * kBlock + threadIdx.x) >> 5; t < K; t += (gridDim.x * kBlock) >> 5) {
        const int e1 = hptr[t + 1];
        const double b = xin[zsrc[t]];
        double s2 = 0.0;
        for (int e = hptr[t] + g; e < e1; e += 32) s2 += S.val[hslot[e]] * y[hidx[e]];
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) s2 += __shfl_xor(s2, d, 64);
        if (g == 0) {
            const double r = b - s2;
            z[t] = pre_scale ? r / pre_scale[unk[t]] : r;
        }
    }
}
// y[pos[t]] = (row t of M) z  [/ the column scale of unknown t: scaled U sweep]; M is lower triangular: a workgroup
// takes kGemvPairs pairs of rows (t, K-1-t) -- every pair K+1 entries together --, thread q the columns q, q + 256, ...
// of all its rows: z[l] is fetched (INLINE_Z: formed, two gathers and a division) once per workgroup and column, the
// 2 kGemvPairs loads of a column are independent, fixed reduction tree.
// INLINE_Z (a head: rows without outside entries): z[l] = xin[zsrc[l]] [/ pre_scale] is formed on the fly, no gather
// launch in front.  Second copy of the result as SweepView::dst2 asks.
// (Until round 5 one pair per workgroup: a head of 1800 unknowns formed its z 900 times over, 15.6 us for 12.5 MB.)
constexpr int kGemvPairs = 2;
template <bool INLINE_Z>
__global__ __launch_bounds__(kBlock) void block_gemv_kernel(int K, const double* __restrict__ M, const double* __restrict__ z,
                                                            const int* __restrict__ zsrc, const double* __restrict__ xin,
                                                            const double* __restrict__ pre_scale,
                                                            const int* __restrict__ tpos, const int* __restrict__ unk,
                                                            const double* __restrict__ post_scale, double* __restrict__ y,
                                                            const int* __restrict__ dst2, double* __restrict__ out2, const int* done) {
    if (done && *done) return;
    constexpr int R = 2 * kGemvPairs;
    __shared__ double red[R][kBlock / 64];
    // rows of the workgroup: pair r = (t0 + r, K - 1 - t0 - r); a pair past the middle is left out (row = -1)
    const int t0 = blockIdx.x * kGemvPairs;
    int row[R];
#pragma unroll
    for (int r = 0; r < kGemvPairs; r++) {
        const int ta = t0 + r, tb = K - 1 - ta;
        row[r] = ta <= tb ? ta : -1;
        row[kGemvPairs + r] = ta < tb ? tb : -1;
    }
    const int ncol = K - t0;                                       // the longest row of the workgroup: K - 1 - t0
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0;
    // four columns per round: their z and all the loads of M issued before the first product
    constexpr int UN = 4;
    for (int l0 = threadIdx.x; l0 < ncol; l0 += UN * kBlock) {
        double zl[UN], mv[UN][R];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int l = l0 + u * kBlock;
            zl[u] = 0.0;
            if (l < ncol) {
                if (INLINE_Z) { zl[u] = xin[zsrc[l]]; if (pre_scale) zl[u] /= pre_scale[unk[l]]; }
                else zl[u] = z[l];
            }
#pragma unroll
            for (int r = 0; r < R; r++) mv[u][r] = l <= row[r] ? M[(size_t)row[r] * K + l] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int r = 0; r < R; r++)
                if (l0 + u * kBlock <= row[r]) acc[r] += mv[u][r] * zl[u];      // (a column beyond the row contributes nothing, whatever its z)
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc[r] += __shfl_xor(acc[r], d, 64);
        if (lane == 0) red[r][wave] = acc[r];
    }
    __syncthreads();
    if (threadIdx.x < R) {
        int t = -1;
#pragma unroll
        for (int r = 0; r < R; r++) if (threadIdx.x == r) t = row[r];
        if (t >= 0) {
            double s2 = 0.0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; w++) s2 += red[threadIdx.x][w];
            const double v = post_scale ? s2 / post_scale[unk[t]] : s2;
            const int pos = tpos[t];
            y[pos] = v;
            if (dst2 && dst2[pos] >= 0) out2[dst2[pos]] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// guard of every explicit inverse (round 4)
// ---------------------------------------------------------------------------
// IPX's late bases are ill conditioned by construction (that is why src/basis.cc:130-152 has a stability loop and
// src/lu_factorization.cc:87-127 a residual test): substitution with a triangular factor is backward stable whatever
// its condition, a product with its computed INVERSE is not (error ~ cond * eps).  So every inverse computed at
// Prepare is probed with two fixed vectors z:  || T (M z) - z ||_inf / || z ||_inf  must not exceed the tolerance
// (IPXK_INVERSE_TOL; default 1e-10 for the inverted levels of a sweep, 1e-8 for a dense block); a block that fails keeps its
// level-scheduled / blocked solve.
// Dense blocks of the factors (hundreds to thousands of rows of a dense LU) get 1e-8: || D X - I || of a computed inverse is
// ~ cond(D) * eps whoever computes it -- measured on well conditioned 1024 / 2048 / 4096 / 8000-row blocks: 9e-12 / 2e-10 /
// 8e-10 / 2e-9 by recursive doubling on the matrix cores, 3e-12 / 8e-11 / 1e-10 by one substitution per column (what a
// dtrsm does) -- and an inverse that good perturbs the solves far below every tolerance the IPM asks of a KKT solve
// (0.3 * sqrt(mu), src/ipm.cc:572); the catastrophes the guard is there for are orders of magnitude above it.
static double inverse_tol(bool dense_block = false) {           // (read per Prepare: the tests switch it)
    const char* e = getenv("IPXK_INVERSE_TOL");
    return e ? atof(e) : dense_block ? 1e-8 : 1e-10;
}
__device__ __forceinline__ double probe_z(int q, int l) {       // entries in [0.5, 1.5], two unrelated sign patterns
    const unsigned h = (unsigned)l * 2654435761u + (unsigned)q * 40503u;
    const double mag = 0.5 + (double)((h >> 9) & 1023u) / 1024.0;
    return ((h >> 20) ^ (unsigned)(q * l)) & 1u ? -mag : mag;
}
__device__ __forceinline__ void probe_max(double* slot, double v) {   // maximum of non-negative doubles through their bit patterns
    if (!(v == v)) v = __builtin_huge_val();                           // a NaN residual fails the test
    atomicMax(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double_as_longlong(v));
}
// w_q = M z_q for the lower-triangular M of an inverted head / tail (row major, K x K): one wavefront per row
__global__ __launch_bounds__(kBlock) void block_probe_mz_kernel(int K, const double* __restrict__ M, double* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); i < K; i += gridDim.x * (kBlock / 64)) {
        double s0 = 0.0, s1 = 0.0;
        for (int l = lane; l <= i; l += 64) { const double a = M[(size_t)i * K + l]; s0 += a * probe_z(0, l); s1 += a * probe_z(1, l); }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64); }
        if (lane == 0) { w[i] = s0; w[K + i] = s1; }
    }
}
// res[q] = max_i | (T22 w_q)_i - z_q(i) |,  T22 = diagonal + the block's inside entries (unscaled)
__global__ void block_probe_res_kernel(int K, const int* __restrict__ tptr, const int* __restrict__ tcol, const double* __restrict__ tval,
                                       const double* __restrict__ dg, const int* __restrict__ tpos, const double* __restrict__ w, double* res) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) {
        const double d = dg[tpos[i]];
        double r0 = d * w[i], r1 = d * w[K + i];
        for (int e = tptr[i]; e < tptr[i + 1]; e++) { const int k = tcol[e]; r0 += tval[e] * w[k]; r1 += tval[e] * w[K + k]; }
        probe_max(res + 0, fabs(r0 - probe_z(0, i)));
        probe_max(res + 1, fabs(r1 - probe_z(1, i)));
    }
}
// dense block D22 = (L22 + I) U22 (column major in D: L22 below, U22 on and above the diagonal), inv row major:
// w_q = inv z_q (one wavefront per row), t_q = U22 w_q, r_q = (L22 + I) t_q - z_q (one thread per row: lanes read a column's
// consecutive rows)
__global__ __launch_bounds__(kBlock) void bump_probe_mz_kernel(int kb, const double* __restrict__ inv, double* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); i < kb; i += gridDim.x * (kBlock / 64)) {
        double s0 = 0.0, s1 = 0.0;
        for (int l = lane; l < kb; l += 64) { const double a = inv[(size_t)i * kb + l]; s0 += a * probe_z(0, l); s1 += a * probe_z(1, l); }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64); }
        if (lane == 0) { w[i] = s0; w[kb + i] = s1; }
    }
}
// t = U22 w (upper part of D with the diagonal) and r = (L22 + I) t - z, D column major: a workgroup takes 64 rows and a
// chunk of 256 columns (lanes along the rows: every load is a 512-byte segment of a column of D, 4 column groups per
// workgroup), the chunks' partial sums are added in chunk order by the second kernel of each stage
constexpr int kProbeChunk = 256;
__global__ __launch_bounds__(kBlock) void bump_probe_partial_kernel(int kb, const double* __restrict__ D, const double* __restrict__ w, int upper,
                                                                    double* __restrict__ part) {
    __shared__ double red[2][4][64];
    const int r = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int l0 = blockIdx.y * kProbeChunk, l1 = min(kb, l0 + kProbeChunk);
    double s0 = 0.0, s1 = 0.0;
    if (r < kb)
        for (int l = l0 + g; l < l1; l += 4) {
            const bool in = upper ? l >= r : l < r;            // U22: columns from the diagonal on; L22: strictly below it
            if (in) { const double a = D[(size_t)l * kb + r]; s0 += a * w[l]; s1 += a * w[kb + l]; }
        }
    red[0][g][threadIdx.x & 63] = s0; red[1][g][threadIdx.x & 63] = s1;
    __syncthreads();
    if (g == 0 && r < kb) {
        const int x = threadIdx.x;
        part[((size_t)blockIdx.y * 2 + 0) * kb + r] = ((red[0][0][x] + red[0][1][x]) + red[0][2][x]) + red[0][3][x];
        part[((size_t)blockIdx.y * 2 + 1) * kb + r] = ((red[1][0][x] + red[1][1][x]) + red[1][2][x]) + red[1][3][x];
    }
}
// stage 1 (res == nullptr): t = sum of the chunks;  stage 2: r = t + sum of the chunks - z, res[q] = max |r_q|
__global__ void bump_probe_finish_kernel(int kb, int nchunks, const double* __restrict__ part, const double* __restrict__ tin,
                                         double* __restrict__ tout, double* res) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kb; i += gridDim.x * blockDim.x) {
        double s0 = tin ? tin[i] : 0.0, s1 = tin ? tin[kb + i] : 0.0;
        for (int c = 0; c < nchunks; c++) { s0 += part[((size_t)c * 2 + 0) * kb + i]; s1 += part[((size_t)c * 2 + 1) * kb + i]; }
        if (tout) { tout[i] = s0; tout[kb + i] = s1; }
        if (res) {
            probe_max(res + 0, fabs(s0 - probe_z(0, i)));
            probe_max(res + 1, fabs(s1 - probe_z(1, i)));
        }
    }
}
// (z has entries of magnitude in [0.5, 1.5]: || z ||_inf is between 1 and 1.5 for any block of a few rows, the residuals are taken as they are)

// the levels [la, lb) of S as an inverted block
static void build_block(Context* c, Sweep& S, Sweep::Block& T, int la, int lb, const char* what) {
    hipStream_t s = c->stream;
    const int c0 = S.level_chunk[la], c1 = S.level_chunk[lb], nc = c1 - c0;
    int64_t K = 0;
    for (int l = la; l < lb; l++) K += S.level_width[l];
    std::vector<ChunkDesc> ch((size_t)nc);
    IPXK_HIP(hipMemcpyAsync(ch.data(), S.chunks.get() + c0, (size_t)nc * sizeof(ChunkDesc), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    const int p0 = ch[0].pos0, p1 = ch.back().pos0 + (ch.back().width >= 0 ? 64 : kLongLanes), np = p1 - p0;
    std::vector<int> order((size_t)np);
    IPXK_HIP(hipMemcpyAsync(order.data(), S.order.get() + p0, (size_t)np * sizeof(int), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    std::vector<int> tpos, unk, base, rank((size_t)np, -1), lev;
    tpos.reserve((size_t)K); unk.reserve((size_t)K); base.reserve((size_t)K);
    // block order = position order; level boundaries from the level widths (a merged chunk holds its levels in order)
    for (const ChunkDesc& d : ch) {
        const int npos_c = d.width >= 0 ? 64 : kLongLanes;
        for (int q = 0; q < npos_c; q++) {
            const int pos = d.pos0 + q;
            if (order[pos - p0] < 0) continue;
            rank[pos - p0] = (int)tpos.size();
            tpos.push_back(pos);
            unk.push_back(order[pos - p0]);
            base.push_back(d.width >= 0 ? d.ent0 + q : -(d.ent0 + 8 * q + 1));
        }
    }
    IPXK_REQUIRE((int64_t)tpos.size() == K, "inverted block of a sweep: positions and level widths disagree");
    lev.push_back(0);
    for (int l = la; l < lb; l++) lev.push_back(lev.back() + S.level_width[l]);
    const int Ki = (int)K;
    DevBuf<int> &dbase = T.w_base, &drank = T.w_rank, &hcnt = T.w_hcnt, &tcnt = T.w_tcnt, &dlev = T.w_lev, &tptr = T.w_tptr, &tcol = T.w_tcol;
    DevBuf<double>& tval = T.w_tval;
    hcnt.ensure((size_t)Ki); tcnt.ensure((size_t)Ki);
    T.pos.upload(tpos, s); T.unk.upload(unk, s);
    dbase.upload(base, s); drank.upload(rank, s); dlev.upload(lev, s);
    hipLaunchKernelGGL(block_count_kernel, dim3(vec_grid(Ki)), dim3(kBlock), 0, s, Ki, p0, T.pos.get(), dbase.get(), S.len.get(),
                       S.idx.get(), hcnt.get(), tcnt.get());
    std::vector<int> hc((size_t)Ki), tc((size_t)Ki), hp((size_t)Ki + 1, 0), tp((size_t)Ki + 1, 0);
    hcnt.download(hc.data(), hc.size(), s); tcnt.download(tc.data(), tc.size(), s);
    IPXK_HIP(hipStreamSynchronize(s));
    for (int t = 0; t < Ki; t++) { hp[t + 1] = hp[t] + hc[t]; tp[t + 1] = tp[t] + tc[t]; }
    T.hptr.upload(hp, s); tptr.upload(tp, s);
    T.nh = hp[Ki];
    T.hslot.ensure((size_t)std::max(T.nh, 1)); T.hidx.ensure((size_t)std::max(T.nh, 1)); T.zsrc.ensure((size_t)Ki); tcol.ensure((size_t)std::max(tp[Ki], 1)); tval.ensure((size_t)std::max(tp[Ki], 1));
    hipLaunchKernelGGL(block_fill_kernel, dim3(vec_grid(Ki)), dim3(kBlock), 0, s, Ki, p0, T.pos.get(), dbase.get(), S.len.get(),
                       S.idx.get(), S.val.get(), drank.get(), T.hptr.get(), tptr.get(), T.hslot.get(), T.hidx.get(), tcol.get(), tval.get());
    T.M.ensure((size_t)Ki * Ki); T.z.ensure((size_t)Ki);
    IPXK_HIP(hipMemsetAsync(T.M.get(), 0, (size_t)Ki * Ki * sizeof(double), s));
    hipLaunchKernelGGL(block_inverse_kernel, dim3((Ki + 63) / 64), dim3(kBlockInvThreads), 0, s, Ki, lb - la, dlev.get(), tptr.get(),
                       tcol.get(), tval.get(), S.diag.get(), T.pos.get(), T.M.get());
    // the guard: T22 (M z) against z for two fixed vectors
    T.w_probe.ensure((size_t)2 * Ki + 2);
    DevBuf<double>& pw = T.w_probe;
    IPXK_HIP(hipMemsetAsync(pw.get() + 2 * (size_t)Ki, 0, 2 * sizeof(double), s));
    hipLaunchKernelGGL(block_probe_mz_kernel, dim3((Ki + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, s, Ki, T.M.get(), pw.get());
    hipLaunchKernelGGL(block_probe_res_kernel, dim3(vec_grid(Ki)), dim3(kBlock), 0, s, Ki, tptr.get(), tcol.get(), tval.get(), S.diag.get(),
                       T.pos.get(), pw.get(), pw.get() + 2 * (size_t)Ki);
    double h[2] = {0.0, 0.0};
    IPXK_HIP(hipMemcpyAsync(h, pw.get() + 2 * (size_t)Ki, sizeof h, hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));                             // also: the host vectors uploaded above go out of scope
    IPXK_HIP(hipGetLastError());
    const double resid = std::max(h[0], h[1]);
    const bool good = resid <= inverse_tol();
    c->split_stats.inverse_probes++;
    c->split_stats.worst_probe = std::max(c->split_stats.worst_probe, resid);
    if (!good) c->split_stats.inverse_rejected++;
    if (getenv("IPXK_VERBOSE") || getenv("IPXK_SWEEP_STATS"))
        fprintf(stderr, "ipxk: sweep %s: levels %d..%d (%d unknowns, %d outside + %d inside entries) inverted; probe |T M z - z| = %.2e%s\n", what, la,
                lb - 1, Ki, T.nh, tp[Ki], resid, good ? "" : " -> REJECTED, these levels stay in the level-scheduled sweep");
    if (!good) { T.K = 0; return; }
    T.K = Ki; T.la = la; T.lb = lb; T.p0 = p0; T.p1 = p1;
}

void build_sweep_blocks(Context* c, Sweep& S, bool level_launches) {
    S.head.K = S.tail.K = 0;
    // IPXK_TAIL_INVERSE / IPXK_HEAD_INVERSE: unknowns at most (default 2048); 0: never.  Large factors only: below
    // IPXK_TAIL_MIN_DIM rows (default 200 000) the whole sweep is a few launches anyway, and the small cases of the
    // test-suite stay bit-identical to the reference's arithmetic.
    auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    const int tail_max = env_int("IPXK_TAIL_INVERSE", 2048), head_max = env_int("IPXK_HEAD_INVERSE", 2048);
    const int min_dim = env_int("IPXK_TAIL_MIN_DIM", 200000);
    constexpr int kMinLevels = 8, kMinUnknowns = 256;
    const int nlev = S.nlevels;
    if (level_launches || nlev < kMinLevels + 1 || S.dim < min_dim) return;
    // the longest run of final levels with at most tail_max unknowns and at most 1/64 of the sweep
    int la = nlev;
    int64_t K = 0;
    for (const int64_t cap = std::min<int64_t>(tail_max, S.dim / 64); la > 1 && K + S.level_width[la - 1] <= cap;) K += S.level_width[--la];
    // (a block starts and ends with a chunk: consecutive tiny levels may share a MERGED chunk, which stays whole)
    while (la < nlev && la > 0 && S.level_chunk[la] == S.level_chunk[la - 1]) K -= S.level_width[la++];
    const bool tail = nlev - la >= kMinLevels && K >= kMinUnknowns;
    if (!tail) la = nlev;
    // ... and of first levels (their rows have no entries outside the block)
    int lb = 0;
    K = 0;
    for (const int64_t cap = std::min<int64_t>(head_max, S.dim / 64); lb < la - 1 && K + S.level_width[lb] <= cap;) K += S.level_width[lb++];
    while (lb > 0 && lb < nlev && S.level_chunk[lb] == S.level_chunk[lb - 1]) K -= S.level_width[--lb];
    const bool head = lb >= kMinLevels && K >= kMinUnknowns;
    if (getenv("IPXK_SWEEP_STATS"))
        fprintf(stderr, "ipxk: sweep blocks: %d levels, tail candidate %d.. (%s), head candidate ..%d (%lld unknowns, %s)\n", nlev, la,
                tail ? "taken" : "not taken", lb - 1, (long long)K, head ? "taken" : "not taken");
    if (head) build_block(c, S, S.head, 0, lb, "head");
    if (tail) build_block(c, S, S.tail, la, nlev, "tail");
}

// ---------------------------------------------------------------------------
// launch plan and sweeps
// ---------------------------------------------------------------------------
// Levels of at most kNarrowLevel chunks are "narrow"; a run of at least kMinXcdLevels narrow levels becomes
// a one-XCD launch, everything between two such runs one all-XCD launch.
// (Measured and dropped: runs of very narrow levels on ONE workgroup with the hand-off through LDS -- the
// hand-off itself is 5x cheaper, but one CU streams the runs' records at 25-50 GB/s and the C3 iteration
// got 70-200 us slower.)
void plan_sweep(Sweep& S, bool level_launches) {
    S.plan.clear();
    const int nlev = S.tail.K > 0 ? S.tail.la : S.nlevels;        // an inverted tail takes the levels from la on,
    const int lfirst = S.head.K > 0 ? S.head.lb : 0;              // an inverted head the levels below lb
    if (nlev == 0) return;
    auto push = [&](int l0, int l1, int kind) {
        const int c0 = S.level_chunk[l0], c1 = S.level_chunk[l1];
        if (c1 > c0) S.plan.push_back({c0, c1, kind, S.merged_prefix[c1] > S.merged_prefix[c0]});
    };
    if (level_launches) {
        for (int l = 0; l < nlev; l++) push(l, l + 1, Sweep::kAllXcds);
        return;
    }
    int narrow_max = kNarrowLevel, min_levels = kMinXcdLevels;
    if (const char* e = getenv("IPXK_SWEEP_NARROW")) narrow_max = atoi(e);
    if (const char* e = getenv("IPXK_SWEEP_MINLEVELS")) min_levels = std::max(1, atoi(e));
    auto nchunks = [&](int lv) { return S.level_chunk[lv + 1] - S.level_chunk[lv]; };
    std::vector<unsigned char> kind(nlev, Sweep::kAllXcds);
    for (int l = lfirst; l < nlev;) {
        if (nchunks(l) > narrow_max) { l++; continue; }
        int b = l;
        while (b < nlev && nchunks(b) <= narrow_max) b++;
        if (b - l >= min_levels) for (int t = l; t < b; t++) kind[t] = Sweep::kOneXcd;
        l = b;
    }
    for (int l = lfirst; l < nlev;) {
        int b = l + 1;
        while (b < nlev && kind[b] == kind[l]) b++;
        push(l, b, kind[l]);
        if (getenv("IPXK_SWEEP_STATS")) {
            int64_t unknowns = 0;
            for (int t = l; t < b; t++) unknowns += S.level_width[t];
            fprintf(stderr, "ipxk: sweep plan: levels %d..%d %s, %d chunks, %lld unknowns; widths", l, b - 1, kind[l] == Sweep::kOneXcd ? "one XCD" : "all XCDs",
                    S.level_chunk[b] - S.level_chunk[l], (long long)unknowns);
            for (int t = l; t < b; t++) fprintf(stderr, " %d", S.level_width[t]);
            fprintf(stderr, "\n");
        }
        l = b;
    }
}

// runs the sweep on the input vector xin (addressed through S.src); the result goes to S.y, which must
// hold the sentinel in every position (fill_results)
static void run_sweep(Context* c, const Sweep& S, bool scaled, const double* xin, const int* done,
                      const int* dst2 = nullptr, double* out2 = nullptr) {
    SplitOperator* sp = c->split;
    SweepView V = S.view(scaled);
    V.dst2 = dst2; V.out2 = out2;
    double* xout = S.y.get();
    // Every workgroup of a run must be resident (a wavefront may wait for a chunk that another workgroup of
    // the same launch owns): never launch more workgroups than the device holds at once.  One block per CU
    // is held back from what the occupancy query reports (it over-reports by one for some kernels).
    // (per operator, i.e. per context and device; the smallest occupancy of the four instantiations counts.  Two
    // contexts must not run basis sweeps on ONE device at the same time: their workgroups would compete for the
    // residency each of them assumes -- a violation ends in the bounded spin's time-out error, not in a hang.)
    if (sp->sweep_grid_all == 0) {
        const char* e = getenv("IPXK_SWEEP_GRID");
        int want = e && atoi(e) > 0 ? atoi(e) : kSweepGrid;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            int per_cu = 1 << 30, got = 0;
            auto ask = [&](auto kernel) {
                int v = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kernel, kBlock, 0) == hipSuccess) { per_cu = std::min(per_cu, v); got++; }
            };
            ask(sweep_run_kernel<true, true>); ask(sweep_run_kernel<true, false>);
            ask(sweep_run_kernel<false, true>); ask(sweep_run_kernel<false, false>);
            if (got == 4) want = std::max(1, std::min(want, prop.multiProcessorCount * std::max(1, per_cu - 1)));
        }
        sp->sweep_grid_all = want;
    }
    // x2 = inverse(T22) (b2 - T21 x1) for the levels the plan leaves out (Sweep::Block)
    auto run_block = [&](const Sweep::Block& T) {
        const double* us = scaled ? sp->uscale.get() : nullptr;
        const double *pre = S.scale_mode == 1 ? us : nullptr, *post = S.scale_mode == 2 ? us : nullptr;
        const int wgs = ((T.K + 1) / 2 + kGemvPairs - 1) / kGemvPairs;
        if (T.nh == 0) {
            hipLaunchKernelGGL(block_gemv_kernel<true>, dim3(wgs), dim3(kBlock), 0, c->stream, T.K, T.M.get(), T.z.get(), T.zsrc.get(), xin, pre,
                               T.pos.get(), T.unk.get(), post, xout, dst2, out2, done);
            return;
        }
        hipLaunchKernelGGL(block_gather_kernel, dim3((T.K * 32 + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, V, T.K, T.zsrc.get(),
                           T.hptr.get(), T.hslot.get(), T.hidx.get(), T.unk.get(), pre, xin, xout, T.z.get(), done);
        hipLaunchKernelGGL(block_gemv_kernel<false>, dim3(wgs), dim3(kBlock), 0, c->stream, T.K, T.M.get(), T.z.get(), T.zsrc.get(), xin, pre,
                           T.pos.get(), T.unk.get(), post, xout, dst2, out2, done);
    };
    if (S.head.K > 0) run_block(S.head);
    const int grid_all = sp->sweep_grid_all;
    static const int wgs_xcd = [] { const char* e = getenv("IPXK_SWEEP_XCD_WGS"); return e && atoi(e) > 0 ? std::min(atoi(e), 64) : kSweepXcdWgs; }();
    for (const Sweep::Launch& L : S.plan) {
        const bool one_xcd = L.kind == Sweep::kOneXcd;
        const int need = (L.c1 - L.c0 + kBlock / 64 - 1) / (kBlock / 64);     // workgroups with a chunk per wave
        int grid = std::max(1, std::min(need, one_xcd ? wgs_xcd : grid_all));
        unsigned epoch = 0;
        if (one_xcd) { grid *= 8; epoch = ++sp->epoch; if (epoch == 0) epoch = ++sp->epoch; }
        auto kernel = S.running ? (L.merged ? sweep_run_kernel<true, true> : sweep_run_kernel<true, false>)
                                : (L.merged ? sweep_run_kernel<false, true> : sweep_run_kernel<false, false>);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, c->stream, V, L.c0, L.c1, xin, xout, one_xcd ? 1 : 0, epoch,
                           sp->xcc_slots.get(), sp->abort_flag.get(), done);
    }
    if (S.tail.K > 0) run_block(S.tail);
}

// sentinel into the result vectors of the given sweeps (one launch)
static void fill_results(Context* c, std::initializer_list<const Sweep*> sweeps, const int* done) {
    FillList L{};
    int k = 0, most = 1;
    for (const Sweep* S : sweeps) { L.p[k] = reinterpret_cast<gu64*>(S->y.get()); L.n[k] = S->npos; most = std::max(most, S->npos); k++; }
    hipLaunchKernelGGL(fill_sentinel_kernel, dim3(vec_grid(most)), dim3(kBlock), 0, c->stream, L, done);
}

static void unpack_result(Context* c, const Sweep& S, const int* perm, double* out) {
    const int m = c->split->m;
    hipLaunchKernelGGL(unpack_result_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, c->stream, m, S.y.get(), S.posof.get(),
                       perm, out);
}

// ForwardSolve: L then U (sparse_matrix.cc:303-306) on a vector in index order; in may be out.
// (The L sweep reads its right-hand side through rowperm, see split_prepare_host: undo that here.)
void forward_solve_dev(Context* c, const double* in, double* out, bool scaled, const int* done) {
    SplitOperator* S = c->split;
    const int m = S->m;
    hipLaunchKernelGGL(scatter_perm_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, c->stream, m, in, S->rowperm.get(),
                       S->w3.get(), done);
    fill_results(c, {&S->Lf, &S->Uf}, done);
    run_sweep(c, S->Lf, scaled, S->w3.get(), done);
    bump_between(c, false, S->Lf.y.get(), done);
    run_sweep(c, S->Uf, scaled, S->Lf.y.get(), done);
    unpack_result(c, S->Uf, nullptr, out);
}
// BackwardSolve: U' then L' (sparse_matrix.cc:308-311); in may be out
void backward_solve_dev(Context* c, const double* in, double* out, bool scaled, const int* done) {
    SplitOperator* S = c->split;
    fill_results(c, {&S->Ut, &S->Lt}, done);
    run_sweep(c, S->Ut, scaled, in, done);
    bump_between(c, true, S->Ut.y.get(), done);
    run_sweep(c, S->Lt, scaled, S->Ut.y.get(), done);
    unpack_result(c, S->Lt, nullptr, out);
}

// raises if a sweep gave up waiting for a dependency (host side, after the stream has been synchronized)
void check_sweep_abort(Context* c) {
    SplitOperator* S = c->split;
    if (!S) return;
    int flag = 0;
    S->abort_flag.download(&flag, 1, c->stream);
    IPXK_HIP(hipStreamSynchronize(c->stream));
    if (flag) {
        IPXK_HIP(hipMemsetAsync(S->abort_flag.get(), 0, sizeof(int), c->stream));
        throw Error(IPXK_E_HIP, "triangular sweep timed out waiting for a dependency");
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
// scaling-dependent part of Prepare, from the raw status / colscale arrays on the device:
// N N' weights colscale^2 on NONBASIC columns (splitted_normal_matrix.cc:42-55)
__global__ void scaling_columns_kernel(int64_t N, const ipxint* __restrict__ status, const double* __restrict__ colscale,
                                       double* __restrict__ W, int* __restrict__ status32, int* bad) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (int64_t)gridDim.x * blockDim.x) {
        const ipxint st = status[j];
        if (st < IPXK_NONBASIC_FIXED || st > IPXK_BASIC_FREE) *bad = 1;
        status32[j] = (int)st;
        W[j] = st == IPXK_NONBASIC ? colscale[j] * colscale[j] : 0.0;
    }
}
// column scaling of U in pivot order (:30-39; nothing for BASIC_FREE) and the free positions (:58-64)
__global__ void scaling_pivots_kernel(int m, const int* __restrict__ colperm, const int* __restrict__ basis,
                                      const int* __restrict__ status32, const double* __restrict__ colscale,
                                      double* __restrict__ uscale, unsigned char* __restrict__ fmask, int* num_free) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) {
        const int j = basis[colperm[k]];
        const int st = status32[j];
        uscale[k] = st == IPXK_BASIC ? colscale[j] : 1.0;
        fmask[k] = st == IPXK_BASIC_FREE ? 1 : 0;
        if (st == IPXK_BASIC_FREE) atomicAdd(num_free, 1);
    }
}

static void upload_scaling(Context* c, SplitOperator* S, const ipxint* status, const double* colscale) {
    const int m = S->m, n = (int)c->n;
    const size_t N = (size_t)n + m;
    hipStream_t s = c->stream;
    S->status_raw.upload(status, N, s);
    S->colscale.upload(colscale, N, s);
    S->Wsplit.ensure(N); S->status.ensure(N);
    S->uscale.ensure(std::max(m, 1)); S->free_mask.ensure(std::max(m, 1));
    S->counters.ensure(2);
    IPXK_HIP(hipMemsetAsync(S->counters.get(), 0, 2 * sizeof(int), s));
    hipLaunchKernelGGL(scaling_columns_kernel, dim3(vec_grid((int64_t)N)), dim3(kBlock), 0, s, (int64_t)N,
                       S->status_raw.get(), S->colscale.get(), S->Wsplit.get(), S->status.get(), S->counters.get());
    hipLaunchKernelGGL(scaling_pivots_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, s, m, S->colperm.get(), S->basis.get(),
                       S->status.get(), S->colscale.get(), S->uscale.get(), S->free_mask.get(), S->counters.get() + 1);
    int h[2] = {0, 0};
    S->counters.download(h, 2, s);
    if (h[0]) throw Error(IPXK_E_ARGUMENT, "status entry out of range");
    S->num_free = h[1];
    rescale_sweeps_device(c, S);
    // N N' runs on the model matrix with weights that are zero on the BASIC and fixed columns: value arrays in
    // which those columns' entries are zero let both passes skip the gathers of those entries (spmv.hip)
    S->masked_values = !(getenv("IPXK_MASKED_VALUES") && getenv("IPXK_MASKED_VALUES")[0] == '0');
    // IPXK_COMPACT_N=0: keep streaming the whole model matrix with masked values (round-2 form)
    const bool compact = !(getenv("IPXK_COMPACT_N") && getenv("IPXK_COMPACT_N")[0] == '0');
    c->Acols.compact.valid = c->Arows.compact.valid = false;
    // large models: N as a pair of gather matrices of its own, built on the device (nmatrix.hip); otherwise ...
    S->real_N = S->masked_values && compact && nmatrix_prepare(c, S->Wsplit.get());
    if (S->masked_values && !S->real_N) {
        // N as a matrix of its own (splitted_normal_matrix.cc:42-55): the tiles of the two gather matrices without
        // the entries of zero-weight columns; layouts without tiles (phased) and long rows keep the masked values
        if (compact) {
            c->Acols.compact_tiles(S->Wsplit.get(), true, s);    // a row of the gather matrix = a structural column
            c->Arows.compact_tiles(S->Wsplit.get(), false, s);   // the gathered index = a structural column
        }
        if (!c->Acols.compact.valid) c->Acols.mask_values(S->Wsplit.get(), true, s);
        if (!c->Arows.compact.valid) c->Arows.mask_values(S->Wsplit.get(), false, s);
    }
}

// what follows the analysis of the factors and the upload of the permutations in a Prepare: scaling, work
// vectors, the composed gather maps of the four sweeps
static void finish_prepare(Context* c, SplitOperator* S, const ipxint* status, const double* colscale) {
    const int m = S->m;
    hipStream_t s = c->stream;
    upload_scaling(c, S, status, colscale);
    const size_t mm = (size_t)std::max(m, 1);
    S->w0.resize(mm); S->w1.resize(mm); S->w2.resize(mm); S->w3.resize(mm); S->tI.resize(mm);
    // where every sweep finds its right-hand side: U' in the input vector itself, L' in the result of U'
    // (by position), L in the input vector THROUGH rowperm (the operator's N N' product is formed in the row
    // order of A: the permutation into pivot order is folded into the gather), U in the result of L;
    // and where the row order of A finds the result of the backward pair
    {
        auto compose = [&](int n, const int* order, const int* map, int* out) {
            if (n > 0) hipLaunchKernelGGL(compose_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, s, n, order, map, out);
        };
        compose(S->Ut.npos, S->Ut.order.get(), nullptr, S->Ut.src.get());
        compose(S->Lt.npos, S->Lt.order.get(), S->Ut.posof.get(), S->Lt.src.get());
        compose(S->Lf.npos, S->Lf.order.get(), S->rowperm.get(), S->Lf.src.get());
        compose(S->Uf.npos, S->Uf.order.get(), S->Lf.posof.get(), S->Uf.src.get());
        for (Sweep* W : {&S->Ut, &S->Lt, &S->Lf, &S->Uf})          // inverted blocks: where their right-hand sides sit
            for (Sweep::Block* T : {&W->head, &W->tail})
                if (T->K > 0)
                    hipLaunchKernelGGL(block_zsrc_kernel, dim3(vec_grid(T->K)), dim3(kBlock), 0, s, T->K, T->pos.get(), W->src.get(), T->zsrc.get());
        S->perm_after_backward.ensure(mm);
        compose(m, S->rowperm_inv.get(), S->Lt.posof.get(), S->perm_after_backward.get());
        // ... and its inverse, by position of the L' sweep (padding positions: -1)
        S->row_after_backward.ensure((size_t)std::max(S->Lt.npos, 1));
        IPXK_HIP(hipMemsetAsync(S->row_after_backward.get(), 0xff, (size_t)std::max(S->Lt.npos, 1) * sizeof(int), s));
        if (m > 0) hipLaunchKernelGGL(invert_map_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, s, m, S->perm_after_backward.get(),
                                      S->row_after_backward.get());
    }
    S->xcc_slots.resize(64);
    IPXK_HIP(hipMemsetAsync(S->xcc_slots.get(), 0, 64 * sizeof(gu64), s));
    S->abort_flag.resize(1);
    IPXK_HIP(hipMemsetAsync(S->abort_flag.get(), 0, sizeof(int), s));
    if (c->partials.size() == 0) c->partials.resize((size_t)kNumPartialSlots * kPartialStride);
    IPXK_HIP(hipStreamSynchronize(s));
}

// ---------------------------------------------------------------------------
// The dense bump of an LU from the device (SplitOperator::bump_*, trisolve.hpp)
// ---------------------------------------------------------------------------
constexpr int kBumpMin = 32;          // smaller bumps stay in the level-scheduled structure
constexpr int kBumpThreads = 1024;

// D22: bump column t = pivot stage s0 + t; U22 on and above the diagonal, L22 (multipliers) below
__global__ __launch_bounds__(kBlock) void bump_extract_kernel(int s0, int kb, const ipxint* __restrict__ Lp, const ipxint* __restrict__ Li,
                                                              const double* __restrict__ Lx, const ipxint* __restrict__ Up, const ipxint* __restrict__ Ui,
                                                              const double* __restrict__ Ux, double* __restrict__ D) {
    // a wavefront per column (a column of U holds up to s0 + kb entries, of which the last <= kb belong to the block)
    const int lane = threadIdx.x & 63;
    for (int t = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); t < kb; t += gridDim.x * (kBlock / 64)) {
        const int j = s0 + t;
        const ipxint u1 = Up[j + 1], u0 = max(Up[j], u1 - kb);     // indices ascend: entries with Ui >= s0 are among the last kb
        for (ipxint p = u0 + lane; p < u1; p += 64)
            if (Ui[p] >= s0) D[(size_t)t * kb + (Ui[p] - s0)] = Ux[p];
        for (ipxint p = Lp[j] + lane; p < Lp[j + 1]; p += 64) D[(size_t)t * kb + (Li[p] - s0)] = Lx[p];
    }
}
// U~: columns < s0 as they are; a bump column keeps its entries above the bump (a prefix: indices ascend) and gets
// the diagonal 1.  L~: columns >= s0 are empty (their entries all lie inside the bump).
__global__ void bump_ucount_kernel(int m, int s0, const ipxint* __restrict__ Up, const ipxint* __restrict__ Ui, int* __restrict__ cnt) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
        int c = (int)(Up[j + 1] - Up[j]);
        if (j >= s0) {
            c = 1;
            for (ipxint p = Up[j]; p < Up[j + 1] && Ui[p] < s0; p++) c++;
        }
        cnt[j] = c;
    }
}
// first entry of every column of U~: unchanged in front of the bump, then the bump columns' counts accumulated
__global__ void bump_ustart_kernel(int m, int s0, const ipxint* __restrict__ Up, const int* __restrict__ cnt, int* __restrict__ start) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x)
        if (j <= s0) start[j] = (int)Up[j];
}
__global__ void bump_ustart_tail_kernel(int m, int s0, const ipxint* __restrict__ Up, const int* __restrict__ cnt, int* __restrict__ start) {
    for (int j = s0 + 1; j < m; j++) start[j] = start[j - 1] + cnt[j - 1];     // <= 4096 columns, once per Prepare
}
__global__ void bump_ufill_kernel(int m, int s0, const ipxint* __restrict__ Up, const ipxint* __restrict__ Ui, const double* __restrict__ Ux,
                                  const int* __restrict__ start, const int* __restrict__ cnt, ipxint* __restrict__ Tp,
                                  ipxint* __restrict__ Ti, double* __restrict__ Tx, const ipxint* __restrict__ Lp, ipxint* __restrict__ TLp) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j <= m; j += gridDim.x * blockDim.x) {
        TLp[j] = Lp[j < s0 ? j : s0];
        if (j == m) { Tp[m] = start[m - 1] + cnt[m - 1]; continue; }
        const int q0 = start[j], c = cnt[j];
        Tp[j] = q0;
        for (int e = 0; e < c; e++) { Ti[q0 + e] = Ui[Up[j] + e]; Tx[q0 + e] = Ux[Up[j] + e]; }
        if (j >= s0) { Ti[q0 + c - 1] = j; Tx[q0 + c - 1] = 1.0; }
    }
}
// inverses of the 64 x 64 diagonal blocks of L22+I (unit lower, from below the diagonal of D) or of U22 (upper)
__global__ __launch_bounds__(64) void bump_invert_blocks_kernel(int kb, const double* __restrict__ D, double* __restrict__ inv, int upper) {
    __shared__ double T[64][65];
    const int b0 = blockIdx.x * 64, nb = min(64, kb - b0), c = threadIdx.x;
    for (int l = 0; l < 64; l++) {
        double v = c == l ? 1.0 : 0.0;
        if (c < nb && l < nb) {
            const double d = D[(size_t)(b0 + l) * kb + (b0 + c)];            // element (row c, column l) of the block
            if (upper) v = c <= l ? d : 0.0;
            else v = c > l ? d : (c == l ? 1.0 : 0.0);
        }
        T[c][l] = v;
    }
    __syncthreads();
    double x[64];
    if (!upper) {
#pragma unroll 1
        for (int i = 0; i < 64; i++) {                         // column c of the inverse: T x = e_c, forward
            double s2 = i == c ? 1.0 : 0.0;
            for (int l = c; l < i; l++) s2 -= T[i][l] * x[l];
            x[i] = i < c ? 0.0 : s2 / T[i][i];
        }
    } else {
#pragma unroll 1
        for (int i = 63; i >= 0; i--) {                        // backward
            double s2 = i == c ? 1.0 : 0.0;
            for (int l = i + 1; l <= c; l++) s2 -= T[i][l] * x[l];
            x[i] = i > c ? 0.0 : s2 / T[i][i];
        }
    }
    double* out = inv + (size_t)blockIdx.x * 64 * 64;
    for (int i = 0; i < 64; i++) out[i + 64 * c] = x[i];       // column major
}
// x_bump <- inverse(D22) x_bump (TRANS: inverse(D22')) in place in a sweep's result vector, between the two sweeps
// of a pair.  One workgroup, x in LDS; per 64-block one product with the inverted diagonal block and one update
// of the part of x still to be solved.
template <bool TRANS>
__device__ __forceinline__ void bump_solve_lds(int kb, const double* __restrict__ D, const double* __restrict__ invL,
                                               const double* __restrict__ invU, double* x, double* xb) {
    const int nblk = (kb + 63) / 64, tid = threadIdx.x;
    // two triangular solves; `first` is the lower-triangular-type one (blocks ascending)
    for (int phase = 0; phase < 2; phase++) {
        const bool lower = phase == 0;                       // !TRANS: L22+I then U22;  TRANS: U22' then (L22+I)'
        const double* inv = TRANS ? (lower ? invU : invL) : (lower ? invL : invU);
        for (int q = 0; q < nblk; q++) {
            const int bq = lower ? q : nblk - 1 - q;
            const int b0 = bq * 64, nb = min(64, kb - b0);
            const double* Ib = inv + (size_t)bq * 64 * 64;
            {   // x_b <- inverse(block) x_b (TRANS: its transpose); 16 threads per row, fixed combination order
                const int r = tid >> 4, g = tid & 15;
                double s2 = 0.0;
                if (r < nb)
                    for (int l = g; l < nb; l += 16) s2 += (TRANS ? Ib[l + 64 * r] : Ib[r + 64 * l]) * x[b0 + l];
#pragma unroll
                for (int d = 8; d >= 1; d >>= 1) s2 += __shfl_xor(s2, d, 64);
                if (g == 0 && r < 64) xb[r] = s2;
            }
            __syncthreads();
            if (tid < nb) x[b0 + tid] = xb[tid];
            // the unknowns still to come lose this block's contribution
            const int i0 = lower ? b0 + nb : 0, i1 = lower ? kb : b0;
            for (int i = i0 + tid; i < i1; i += kBumpThreads) {
                double s2 = x[i];
                // element (row i, column b0 + l) of the triangular matrix of this phase
                //   !TRANS: D[(b0+l)*kb + i]   (L22 below / U22 above the diagonal, column major)
                //    TRANS: D[i*kb + b0 + l]   (the transposed factor)
                int l = 0;
                for (; l + 8 <= nb; l += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = TRANS ? D[(size_t)i * kb + b0 + l + u] : D[(size_t)(b0 + l + u) * kb + i];
#pragma unroll
                    for (int u = 0; u < 8; u++) s2 -= v[u] * xb[l + u];
                }
                for (; l < nb; l++) s2 -= (TRANS ? D[(size_t)i * kb + b0 + l] : D[(size_t)(b0 + l) * kb + i]) * xb[l];
                x[i] = s2;
            }
            __syncthreads();
        }
    }
}
template <bool TRANS>
__global__ __launch_bounds__(kBumpThreads) void bump_solve_kernel(int kb, const double* __restrict__ D, const double* __restrict__ invL,
                                                                  const double* __restrict__ invU, const int* __restrict__ pos,
                                                                  double* y, const int* done, double* gx = nullptr) {
    if (done && *done) return;
    extern __shared__ double xs[];       // kb + 64; a block too large for LDS (more than kBumpLdsRows rows) keeps x in the global scratch gx
    double* x = gx ? gx : xs;
    double* xb = gx ? xs : xs + kb;
    const int tid = threadIdx.x;
    for (int t = tid; t < kb; t += kBumpThreads) x[t] = y[pos[t]];
    __syncthreads();
    bump_solve_lds<TRANS>(kb, D, invL, invU, x, xb);
    for (int t = tid; t < kb; t += kBumpThreads) y[pos[t]] = x[t];
}
// The blocked solve applied to the guard's two vectors: w[q kb + t] = (inverse(D22) z_q)[t] as the one-workgroup solve computes it
// (workgroup q).  Its residual is what an explicit inverse of the same block can be held to.
__global__ __launch_bounds__(kBumpThreads) void bump_probe_solve_kernel(int kb, const double* __restrict__ D, const double* __restrict__ invL,
                                                                        const double* __restrict__ invU, double* __restrict__ w, double* gx = nullptr) {
    extern __shared__ double xs[];       // kb + 64 (or 64 with x in the global scratch: one stretch of kb per workgroup)
    const int q = blockIdx.x;
    double* x = gx ? gx + (size_t)q * kb : xs;
    double* xb = gx ? xs : xs + kb;
    for (int t = threadIdx.x; t < kb; t += kBumpThreads) x[t] = probe_z(q, t);
    __syncthreads();
    bump_solve_lds<false>(kb, D, invL, invU, x, xb);
    for (int t = threadIdx.x; t < kb; t += kBumpThreads) w[(size_t)q * kb + t] = x[t];
}
// Explicit inverse of a large block: workgroup j solves D22 x = e_j with the blocked solve above; x = column j of
// inverse(D22) = row j of its transpose.  Both orientations are stored row major, so that either product below reads
// contiguous rows.
__global__ __launch_bounds__(kBumpThreads) void bump_inverse_kernel(int kb, const double* __restrict__ D, const double* __restrict__ invL,
                                                                    const double* __restrict__ invU, double* __restrict__ inv,
                                                                    double* __restrict__ invT) {
    extern __shared__ double xs[];       // kb + 64
    double* x = xs;
    double* xb = xs + kb;
    const int tid = threadIdx.x, j = blockIdx.x;
    for (int t = tid; t < kb; t += kBumpThreads) x[t] = t == j ? 1.0 : 0.0;
    __syncthreads();
    bump_solve_lds<false>(kb, D, invL, invU, x, xb);
    for (int t = tid; t < kb; t += kBumpThreads) {
        invT[(size_t)j * kb + t] = x[t];
        inv[(size_t)t * kb + j] = x[t];
    }
}
__global__ void bump_gather_kernel(int kb, const int* __restrict__ pos, const double* __restrict__ y, double* __restrict__ x, const int* done) {
    if (done && *done) return;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < kb; t += gridDim.x * blockDim.x) x[t] = y[pos[t]];
}
// y[pos[i]] = row i of M times x: one wavefront per row, lanes stride the row, fixed shuffle tree
__global__ __launch_bounds__(kBlock) void bump_gemv_kernel(int kb, const double* __restrict__ M, const double* __restrict__ x,
                                                           const int* __restrict__ pos, double* __restrict__ y, const int* done) {
    if (done && *done) return;
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); i < kb; i += gridDim.x * (kBlock / 64)) {
        const double* row = M + (size_t)i * kb;
        double s2 = 0.0;
        for (int l = lane; l < kb; l += 64) s2 += row[l] * x[l];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s2 += __shfl_xor(s2, d, 64);
        if (lane == 0) y[pos[i]] = s2;
    }
}
__global__ void bump_positions_kernel(int s0, int kb, const int* __restrict__ posof_fwd, const int* __restrict__ posof_bwd,
                                      int* __restrict__ pf, int* __restrict__ pb) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < kb; t += gridDim.x * blockDim.x) {
        pf[t] = posof_fwd[s0 + t];
        pb[t] = posof_bwd[s0 + t];
    }
}
// between the two sweeps of a pair: `y` is the result of the first one
// the blocked solve keeps the kb unknowns of the block in LDS: beyond 64 KB of dynamic LDS the kernels have to be allowed; beyond the
// 160 KB of a compute unit (blocks of more than kBumpLdsRows rows -- what the LU leaves of an 80 000-row IPM basis) the unknowns live
// in a global scratch vector instead (one workgroup: its own stores are visible to it after a barrier)
constexpr int kBumpLdsRows = 160 * 1024 / 8 - 64;
static void allow_bump_lds(size_t bytes) {
    static size_t allowed = 64 * 1024;
    if (bytes <= allowed) return;
    IPXK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bump_solve_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    IPXK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bump_solve_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    IPXK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bump_inverse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    IPXK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bump_probe_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    allowed = bytes;
}
static void bump_between(Context* c, bool trans, double* y, const int* done) {
    SplitOperator* S = c->split;
    if (S->bump_size == 0) return;
    const int kb = S->bump_size;
    if (S->bump_explicit) {
        // large block: x_bump <- inverse(D22) x_bump (or its transpose) as one product over the chip
        const int* pos = trans ? S->bump_pos_bwd.get() : S->bump_pos_fwd.get();
        hipLaunchKernelGGL(bump_gather_kernel, dim3(vec_grid(kb)), dim3(kBlock), 0, c->stream, kb, pos, y, S->bump_x.get(), done);
        hipLaunchKernelGGL(bump_gemv_kernel, dim3((kb + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, c->stream, kb,
                           trans ? S->bump_invT.get() : S->bump_inv.get(), S->bump_x.get(), pos, y, done);
        return;
    }
    const bool in_lds = kb <= kBumpLdsRows;
    const size_t lds = (size_t)((in_lds ? kb : 0) + 64) * sizeof(double);
    double* gx = nullptr;
    if (in_lds) allow_bump_lds(lds);
    else { S->bump_gx.ensure((size_t)2 * kb); gx = S->bump_gx.get(); }
    if (trans) hipLaunchKernelGGL(bump_solve_kernel<true>, dim3(1), dim3(kBumpThreads), lds, c->stream, kb, S->bumpD.get(), S->bump_invL.get(),
                                  S->bump_invU.get(), S->bump_pos_bwd.get(), y, done, gx);
    else hipLaunchKernelGGL(bump_solve_kernel<false>, dim3(1), dim3(kBumpThreads), lds, c->stream, kb, S->bumpD.get(), S->bump_invL.get(),
                            S->bump_invU.get(), S->bump_pos_fwd.get(), y, done, gx);
}

// Cuts the trailing block [s0, s0 + kb) = [s0, m) out of the factors: D22 = (L22 + I) U22 goes to S->bumpD (dense, with
// the inverted 64 x 64 diagonal blocks), the returned factors are L without L22 and U with U22 replaced by I (trisolve.hpp).
// Exact for ANY trailing block; it pays when the block is (nearly) dense.
struct CutBuffers {                                    // (the operator's own, kept from one Prepare to the next)
    DevBuf<ipxint> &TLp, &TUp, &TUi; DevBuf<double>& TUx;
    explicit CutBuffers(SplitOperator* S) : TLp(S->cut_Lp), TUp(S->cut_Up), TUi(S->cut_Ui), TUx(S->cut_Ux) {}
};
static DeviceFactors cut_dense_block(Context* c, SplitOperator* S, const DeviceFactors& in, int s0, int kb, CutBuffers& B) {
    hipStream_t s = c->stream;
    const int m = S->m, nblk = (kb + 63) / 64;
    S->bumpD.ensure((size_t)kb * kb);
    IPXK_HIP(hipMemsetAsync(S->bumpD.get(), 0, (size_t)kb * kb * sizeof(double), s));
    hipLaunchKernelGGL(bump_extract_kernel, dim3((kb + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, s, s0, kb, in.Lp, in.Li, in.Lx, in.Up, in.Ui,
                       in.Ux, S->bumpD.get());
    DevBuf<int> &cnt = S->cut_cnt, &start = S->cut_start;
    cnt.ensure((size_t)m); start.ensure((size_t)m);
    hipLaunchKernelGGL(bump_ucount_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, s, m, s0, in.Up, in.Ui, cnt.get());
    hipLaunchKernelGGL(bump_ustart_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, s, m, s0, in.Up, cnt.get(), start.get());
    hipLaunchKernelGGL(bump_ustart_tail_kernel, dim3(1), dim3(1), 0, s, m, s0, in.Up, cnt.get(), start.get());
    B.TLp.ensure((size_t)m + 1); B.TUp.ensure((size_t)m + 1);
    B.TUi.ensure((size_t)std::max<int64_t>(in.nzU, 1)); B.TUx.ensure((size_t)std::max<int64_t>(in.nzU, 1));
    hipLaunchKernelGGL(bump_ufill_kernel, dim3(vec_grid(m + 1)), dim3(kBlock), 0, s, m, s0, in.Up, in.Ui, in.Ux, start.get(), cnt.get(),
                       B.TUp.get(), B.TUi.get(), B.TUx.get(), in.Lp, B.TLp.get());
    ipxint ends[2] = {0, 0};
    IPXK_HIP(hipMemcpyAsync(&ends[0], B.TLp.get() + m, sizeof(ipxint), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipMemcpyAsync(&ends[1], B.TUp.get() + m, sizeof(ipxint), hipMemcpyDeviceToHost, s));
    S->bump_invL.ensure((size_t)nblk * 64 * 64); S->bump_invU.ensure((size_t)nblk * 64 * 64);
    hipLaunchKernelGGL(bump_invert_blocks_kernel, dim3(nblk), dim3(64), 0, s, kb, S->bumpD.get(), S->bump_invL.get(), 0);
    hipLaunchKernelGGL(bump_invert_blocks_kernel, dim3(nblk), dim3(64), 0, s, kb, S->bumpD.get(), S->bump_invU.get(), 1);
    // large blocks: the inverse itself (IPXK_BUMP_INVERSE_MIN rows and more, default 512; 0 = never), so that the solve
    // between two sweeps is one matrix-vector product over the chip instead of a blocked solve by one workgroup
    // (measured with a 1316-row block: 4.3 ms -> 0.06 ms per CR iteration of the drop-in solver; the reference's CPU solver: 0.54 ms)
    // ... up to IPXK_BUMP_INVERSE_MAX rows (default: every block the LU can produce).  The inverse costs kb workgroups a whole
    // blocked solve each -- about 1 s at 8000 rows -- but the one-workgroup solve it replaces takes 18 ms per application there:
    // measured on a 12 000 x 30 000 LP through the drop-in solver (24 Prepares, 1100 CR iterations): 23 + 0.7 s against 43 s
    static const int inverse_min = [] { const char* e = getenv("IPXK_BUMP_INVERSE_MIN"); return e ? atoi(e) : 512; }();
    static const int inverse_max = [] { const char* e = getenv("IPXK_BUMP_INVERSE_MAX"); return e ? atoi(e) : 32768; }();
    S->bump_explicit = inverse_min > 0 && kb >= inverse_min && kb <= inverse_max;
    if (S->bump_explicit) {
        S->bump_inv.ensure((size_t)kb * kb); S->bump_invT.ensure((size_t)kb * kb); S->bump_x.ensure((size_t)kb);
        // blocks of IPXK_DENSE_INVERSE_MIN rows and more (default: all of them) on the matrix cores (dense_inverse.hip: triangular
        // inverses by recursive doubling + one product, v_mfma_f64_16x16x4_f64); below, or with IPXK_DENSE_INVERSE_MIN=0, the
        // older kernel: one blocked solve per column of the identity
        const char* di_env = getenv("IPXK_DENSE_INVERSE_MIN");          // (read per Prepare: the tests switch it)
        const int di_min = di_env ? atoi(di_env) : 1;
        const bool by_blas = di_min > 0 && kb >= di_min;
        const bool in_lds = kb <= kBumpLdsRows;
        if (in_lds) allow_bump_lds((size_t)(kb + 64) * sizeof(double));
        IPXK_REQUIRE(by_blas || in_lds, "a dense block of this size is inverted on the matrix cores only (IPXK_DENSE_INVERSE_MIN)");
        if (!by_blas) hipLaunchKernelGGL(bump_inverse_kernel, dim3(kb), dim3(kBumpThreads), (size_t)(kb + 64) * sizeof(double), s, kb, S->bumpD.get(),
                           S->bump_invL.get(), S->bump_invU.get(), S->bump_inv.get(), S->bump_invT.get());
        // the guard (whoever computed the inverse): D22 (inverse z) against z; a block that fails keeps the blocked solve.  The
        // inverse from the matrix cores gets up to two refinement steps first (X += X (I - D22 X)) when the probe says they can
        // converge: the IPM's late bases are ill conditioned, and the product of two triangular inverses then misses the
        // tolerance by two or three digits (dense_inverse.hip) -- without the steps every block of a 12 000 x 30 000 LP's main
        // phase fell back to the one-workgroup solve, 12 ms per CR iteration instead of 0.5.
        const int nchunks = (kb + kProbeChunk - 1) / kProbeChunk;
        S->bump_probe.ensure((size_t)4 * kb + 2 + (size_t)2 * nchunks * kb);
        double* pw = S->bump_probe.get();
        double* part = pw + 4 * (size_t)kb + 2;
        static const int max_refine = [] { const char* e = getenv("IPXK_DENSE_INVERSE_REFINE"); return e ? std::max(0, atoi(e)) : 2; }();
        double resid = 0.0, first_resid = 0.0;
        int refine = 0;
        for (;;) {
            if (by_blas) dense_lu_inverse(c, kb, S->bumpD.get(), S->bump_invL.get(), S->bump_invU.get(), S->bump_invT.get(), S->bump_inv.get(), refine);
            IPXK_HIP(hipMemsetAsync(pw + 4 * (size_t)kb, 0, 2 * sizeof(double), s));
            hipLaunchKernelGGL(bump_probe_mz_kernel, dim3((kb + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, s, kb, S->bump_inv.get(), pw);
            const dim3 pgrid((kb + 63) / 64, nchunks);
            hipLaunchKernelGGL(bump_probe_partial_kernel, pgrid, dim3(kBlock), 0, s, kb, S->bumpD.get(), pw, 1, part);
            hipLaunchKernelGGL(bump_probe_finish_kernel, dim3(vec_grid(kb)), dim3(kBlock), 0, s, kb, nchunks, part, (const double*)nullptr, pw + 2 * (size_t)kb,
                               (double*)nullptr);
            hipLaunchKernelGGL(bump_probe_partial_kernel, pgrid, dim3(kBlock), 0, s, kb, S->bumpD.get(), pw + 2 * (size_t)kb, 0, part);
            hipLaunchKernelGGL(bump_probe_finish_kernel, dim3(vec_grid(kb)), dim3(kBlock), 0, s, kb, nchunks, part, pw + 2 * (size_t)kb, (double*)nullptr,
                               pw + 4 * (size_t)kb);
            double h[2] = {0.0, 0.0};
            IPXK_HIP(hipMemcpyAsync(h, pw + 4 * (size_t)kb, sizeof h, hipMemcpyDeviceToHost, s));
            IPXK_HIP(hipStreamSynchronize(s));
            resid = std::max(h[0], h[1]);
            if (refine == 0) first_resid = resid;
            c->split_stats.inverse_probes++;
            if (resid <= inverse_tol(true) || !by_blas || refine >= max_refine || !(resid < 0.25)) break;
            refine++;
            c->split_stats.inverse_refined++;
        }
        // an inverse that misses the tolerance narrowly is held against what it replaces: the blocked solve's own residual on the
        // same two vectors (an ill-conditioned block leaves neither at 1e-8); within four times that, it stays
        double resid_solve = -1.0;
        if (!(resid <= inverse_tol(true)) && resid < 1e-5 && inverse_tol(true) > 0.0) {       // (tolerance 0: "reject everything", tests)
            IPXK_HIP(hipMemsetAsync(pw + 4 * (size_t)kb, 0, 2 * sizeof(double), s));
            double* gx = nullptr;
            if (!in_lds) { S->bump_gx.ensure((size_t)2 * kb); gx = S->bump_gx.get(); }
            hipLaunchKernelGGL(bump_probe_solve_kernel, dim3(2), dim3(kBumpThreads), (size_t)((in_lds ? kb : 0) + 64) * sizeof(double), s, kb, S->bumpD.get(),
                               S->bump_invL.get(), S->bump_invU.get(), pw, gx);
            const dim3 pgrid((kb + 63) / 64, nchunks);
            hipLaunchKernelGGL(bump_probe_partial_kernel, pgrid, dim3(kBlock), 0, s, kb, S->bumpD.get(), pw, 1, part);
            hipLaunchKernelGGL(bump_probe_finish_kernel, dim3(vec_grid(kb)), dim3(kBlock), 0, s, kb, nchunks, part, (const double*)nullptr, pw + 2 * (size_t)kb,
                               (double*)nullptr);
            hipLaunchKernelGGL(bump_probe_partial_kernel, pgrid, dim3(kBlock), 0, s, kb, S->bumpD.get(), pw + 2 * (size_t)kb, 0, part);
            hipLaunchKernelGGL(bump_probe_finish_kernel, dim3(vec_grid(kb)), dim3(kBlock), 0, s, kb, nchunks, part, pw + 2 * (size_t)kb, (double*)nullptr,
                               pw + 4 * (size_t)kb);
            double h[2] = {0.0, 0.0};
            IPXK_HIP(hipMemcpyAsync(h, pw + 4 * (size_t)kb, sizeof h, hipMemcpyDeviceToHost, s));
            IPXK_HIP(hipStreamSynchronize(s));
            resid_solve = std::max(h[0], h[1]);
        }
        const bool accepted = resid <= inverse_tol(true) || (resid_solve >= 0.0 && resid <= 4.0 * resid_solve);
        c->split_stats.worst_probe = std::max(c->split_stats.worst_probe, resid);
        if (!accepted) { S->bump_explicit = false; c->split_stats.inverse_rejected++; }
        if (resid_solve >= 0.0 && (getenv("IPXK_VERBOSE") || getenv("IPXK_SWEEP_STATS")))
            fprintf(stderr, "ipxk:   (the blocked solve's own probe on this block: %.2e)\n", resid_solve);
        if (getenv("IPXK_VERBOSE") || getenv("IPXK_SWEEP_STATS"))
            fprintf(stderr, "ipxk: dense block of %d rows inverted (%s); probe |D (inverse z) - z| = %.2e%s%s\n", kb,
                    by_blas ? "recursive doubling on the matrix cores" : "one blocked solve per column", resid,
                    refine > 0 ? (refine == 1 ? " after one refinement step" : " after two refinement steps") : "",
                    S->bump_explicit ? "" : " -> REJECTED, the blocked solve stays");
        if (refine > 0 && (getenv("IPXK_VERBOSE") || getenv("IPXK_SWEEP_STATS"))) fprintf(stderr, "ipxk:   (probe before the refinement %.2e)\n", first_resid);
    }
    IPXK_HIP(hipStreamSynchronize(s));               // cnt / start go out of scope; ends
    S->bump_start = s0;
    S->bump_size = kb;
    return DeviceFactors{B.TLp.get(), in.Li, B.TUp.get(), B.TUi.get(), in.Lx, B.TUx.get(), ends[0], ends[1]};
}

// A dense trailing block in factors that come from the host (the dense bump of an LU kernel -- lu.hip's or any other
// -- is pivoted last).  Where
// the device computed the factors it knows the block (LuView); factors handed over by ipx::Basis have gone through
// the host, and without this a 2000-row bump is a chain of 2000 dependency levels (12 ms per operator application
// against 1 ms with the block cut out: the drop-in class on the IPM's random LPs).
static int trailing_dense_block(int m, const ipxint* Lp) {
    // the largest trailing block (up to 32768 columns: the largest dense block the LU produces) whose part of L is at least 30 % full: a dense LU of a sparse bump
    // starts with sparse columns and fills up, so single columns say little; the block as a whole does
    int s0 = m;
    const int lo = std::max(0, m - 32768);
    for (int j = m - 2; j >= lo; j--) {
        const double kb = (double)(m - j), have = (double)(Lp[m] - Lp[j]);
        if (have >= 0.3 * (kb * (kb - 1.0) / 2.0)) s0 = j;
    }
    return s0;
}

void split_prepare_host(Context* c, const ipxint* Lp, const ipxint* Li, const double* Lx,
                        const ipxint* Up, const ipxint* Ui, const double* Ux, const ipxint* rowperm,
                        const ipxint* colperm, const ipxint* basis, const ipxint* status,
                        const double* colscale) {
    maxvol_drop_etas(c);
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
    // the operator object (and its device buffers) is reused from one Prepare to the next; while it is
    // being rebuilt the context has no operator, and a failure leaves it that way
    std::unique_ptr<SplitOperator> S(c->split ? c->split : c->split_spare ? c->split_spare : new SplitOperator);
    if (!c->split) c->split_spare = nullptr;
    c->split = nullptr;
    S->m = m;
    if (const char* e = getenv("IPXK_TRISOLVE")) S->level_launches = std::string(e) == "levels";

    const bool verbose = getenv("IPXK_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = now();
    S->bump_start = S->bump_size = 0;
    {
        // factors as given, on the device; a dense trailing block (the bump of the LU) is cut out of the sweeps
        DevBuf<ipxint> &dLp = S->in_Lp, &dLi = S->in_Li, &dUp = S->in_Up, &dUi = S->in_Ui;
        DevBuf<double> &dLx = S->in_Lx, &dUx = S->in_Ux;
        const int64_t nzL = Lp[m], nzU = Up[m];
        dLp.upload(Lp, (size_t)m + 1, s); dUp.upload(Up, (size_t)m + 1, s);
        dLi.upload(Li, (size_t)nzL, s);   dLx.upload(Lx, (size_t)nzL, s);
        dUi.upload(Ui, (size_t)nzU, s);   dUx.upload(Ux, (size_t)nzU, s);
        dLi.ensure(1); dLx.ensure(1);
        const DeviceFactors F0{dLp.get(), dLi.get(), dUp.get(), dUi.get(), dLx.get(), dUx.get(), nzL, nzU};
        const char* dense_env = getenv("IPXK_BUMP_DENSE");
        const int bump_min = getenv("IPXK_BUMP_MIN") ? atoi(getenv("IPXK_BUMP_MIN")) : kBumpMin;
        const int s0 = m > 0 ? trailing_dense_block(m, Lp) : m;
        CutBuffers cut(S.get());
        // the cut takes the entries of a U column above the block as a PREFIX of the column and the block's own as its last
        // entries: that needs ascending row indices inside the columns from s0 on.  The contract of ipxk_split_prepare asks
        // for the diagonal last only (the reference's GetLuFactors does return sorted columns); unsorted ones keep the whole
        // factors in the sweeps.
        bool sorted_U = true;
        for (int j = s0; j < m && sorted_U; j++)
            for (ipxint p = Up[j] + 1; p < Up[j + 1]; p++)
                if (Ui[p] <= Ui[p - 1]) { sorted_U = false; break; }
        if (sorted_U && m - s0 >= bump_min && m - s0 <= 32768 && !(dense_env && dense_env[0] == '0')) {
            const DeviceFactors F = cut_dense_block(c, S.get(), F0, s0, m - s0, cut);
            analyse_sweeps_resident(c, S.get(), F, nullptr, nullptr, nullptr, nullptr);
        } else {
            analyse_sweeps_resident(c, S.get(), F0, Lp, Li, Up, Ui);
        }
        if (S->bump_size > 0) {
            S->bump_pos_fwd.ensure((size_t)S->bump_size); S->bump_pos_bwd.ensure((size_t)S->bump_size);
            hipLaunchKernelGGL(bump_positions_kernel, dim3(vec_grid(S->bump_size)), dim3(kBlock), 0, s, S->bump_start, S->bump_size,
                               S->Lf.posof.get(), S->Ut.posof.get(), S->bump_pos_fwd.get(), S->bump_pos_bwd.get());
        }
        IPXK_HIP(hipStreamSynchronize(s));           // the uploaded factors go out of scope
    }
    const double tp1 = now();
    // permutations (InversePerm, utils.cc:73-80) and bookkeeping for KKTSolverBasis::_Solve
    {
        std::vector<int> rpm(m), rpi(m), cpm(m), bs(m);
        for (int i = 0; i < m; i++) { rpm[i] = (int)rowperm[i]; cpm[i] = (int)colperm[i]; bs[i] = (int)basis[i]; }
        for (int i = 0; i < m; i++) rpi[rpm[i]] = i;
        S->rowperm.upload(rpm, s);
        S->rowperm_inv.upload(rpi, s);
        S->colperm.upload(cpm, s);
        S->basis.upload(bs, s);
    }
    finish_prepare(c, S.get(), status, colscale);
    if (verbose)
        fprintf(stderr, "ipxk: split_prepare: analysis and packing %.1f ms, permutations/scaling %.1f ms\n",
                (tp1 - tp0) * 1e3, (now() - tp1) * 1e3);
    c->split = S.release();
}

// 64-bit permutations / basis list on the device -> the operator's 32-bit copies (+ InversePerm, utils.cc:73-80)
__global__ void perms_from_lu_kernel(int m, const ipxint* __restrict__ rowperm, const ipxint* __restrict__ colperm,
                                     const ipxint* __restrict__ basis, int* __restrict__ rpm, int* __restrict__ rpi,
                                     int* __restrict__ cpm, int* __restrict__ bs) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) {
        const int r = (int)rowperm[k];
        rpm[k] = r;
        rpi[r] = k;
        cpm[k] = (int)colperm[k];
        bs[k] = (int)basis[k];
    }
}

// Prepare from the factors of the last ipxk_lu_factorize_basis: Basis::GetLuFactors (basis.cc:162-166) +
// SplittedNormalMatrix::Prepare (splitted_normal_matrix.cc:18-66) with L, U and the permutations never leaving the
// device.
void split_prepare_lu(Context* c, const ipxint* status, const double* colscale) {
    maxvol_drop_etas(c);
    LuView V;
    IPXK_REQUIRE(lu_view(c, &V) && V.from_basis, "no LU factorization of a basis of this context's matrix (ipxk_lu_factorize_basis)");
    IPXK_REQUIRE(V.ndep == 0, "the factorization replaced dependent columns: repair the basis and factorize again "
                              "(Basis::AdaptToSingularFactorization, src/basis.cc)");
    const int m = (int)c->m;
    IPXK_REQUIRE(V.dim == m, "dimension mismatch");
    IPXK_REQUIRE(c->nranks == 1, "the basis path does not shard: run it as independent replicas");
    hipStream_t s = c->stream;
    std::unique_ptr<SplitOperator> S(c->split ? c->split : c->split_spare ? c->split_spare : new SplitOperator);
    if (!c->split) c->split_spare = nullptr;
    c->split = nullptr;
    S->m = m;
    if (const char* e = getenv("IPXK_TRISOLVE")) S->level_launches = std::string(e) == "levels";
    // the dense bump leaves the level-scheduled structure (SplitOperator::bump_*)
    DeviceFactors F = V.F;
    CutBuffers cut(S.get());
    S->bump_start = S->bump_size = 0;
    const char* dense_env = getenv("IPXK_BUMP_DENSE");
    const int bump_min = getenv("IPXK_BUMP_MIN") ? atoi(getenv("IPXK_BUMP_MIN")) : kBumpMin;      // (tests)
    if (V.bump_size >= bump_min && V.bump_size > 0 && V.bump_start + V.bump_size == m && !(dense_env && dense_env[0] == '0'))
        F = cut_dense_block(c, S.get(), V.F, V.bump_start, V.bump_size, cut);
    analyse_sweeps_resident(c, S.get(), F, nullptr, nullptr, nullptr, nullptr);
    const size_t mm = (size_t)std::max(m, 1);
    if (S->bump_size > 0) {
        S->bump_pos_fwd.ensure((size_t)S->bump_size); S->bump_pos_bwd.ensure((size_t)S->bump_size);
        hipLaunchKernelGGL(bump_positions_kernel, dim3(vec_grid(S->bump_size)), dim3(kBlock), 0, s, S->bump_start, S->bump_size,
                           S->Lf.posof.get(), S->Ut.posof.get(), S->bump_pos_fwd.get(), S->bump_pos_bwd.get());
    }
    S->rowperm.ensure(mm); S->rowperm_inv.ensure(mm); S->colperm.ensure(mm); S->basis.ensure(mm);
    if (m > 0)
        hipLaunchKernelGGL(perms_from_lu_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, s, m, V.rowperm, V.colperm, V.basis,
                           S->rowperm.get(), S->rowperm_inv.get(), S->colperm.get(), S->basis.get());
    finish_prepare(c, S.get(), status, colscale);
    c->split = S.release();
}

// Same basis, new scaling factors: KKTSolverBasis::_Factorize without basis changes
// (kkt_solver_basis.cc:59-64 keeps the factorization; only the scaling of U and N changes,
// splitted_normal_matrix.cc:30-55).  The level schedule and the packed factors are reused.
void split_rescale_host(Context* c, const ipxint* status, const double* colscale) {
    SplitOperator* S = c->split;
    upload_scaling(c, S, status, colscale);
    IPXK_HIP(hipStreamSynchronize(c->stream));
}

// ---- the operator behind an eta file (Context::etas_live, maxvolume.hip) ----
// B_new = B_old E (E the product of Maxvolume's last exchanges, acting on vectors by basis position), hence with the column scaling S of
// the NEW basis   inverse(B~) = inverse(S) inverse(E) inverse(U) inverse(L),   inverse(B~') = inverse(L') inverse(U') inverse(E') inverse(S):
// the UNSCALED sweeps of the resident factors, the eta file between them and the scaling, and the scaling as a vector operation.
// t[colperm[k]] = rhs[k] / scale[k]          (pivot order -> basis position)
__global__ void eta_scatter_scale_kernel(int m, const double* __restrict__ rhs, const int* __restrict__ colperm, const double* __restrict__ scale,
                                         double* __restrict__ t, const int* done) {
    if (done && *done) return;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) t[colperm[k]] = rhs[k] / scale[k];
}
// lhs = free ? 0 : t[colperm[k]] / scale[k] + rhs;  partial dot rhs'lhs       (splitted_normal_matrix.cc:112-116)
__global__ __launch_bounds__(kBlock) void split_finish_etas_kernel(int m, const double* __restrict__ rhs, const unsigned char* __restrict__ free_mask,
                                                                   const double* __restrict__ t, const int* __restrict__ colperm,
                                                                   const double* __restrict__ scale, double* __restrict__ lhs, double* partial,
                                                                   const int* done) {
    if (done && *done) return;
    __shared__ double red[kBlock / 64 + 1];
    double acc = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) {
        const double r = rhs[i];
        const double l = free_mask[i] ? 0.0 : t[colperm[i]] / scale[i] + r;
        lhs[i] = l;
        acc += r * l;
    }
    acc = block_reduce<SumOp>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ void narrow_basis_kernel(int m, const ipxint* __restrict__ in, int* __restrict__ out) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < m; p += gridDim.x * blockDim.x) out[p] = (int)in[p];
}
void split_follow_basis(Context* c, const ipxint* basis_dev, const ipxint* status, const double* colscale) {
    SplitOperator* S = c->split;
    IPXK_REQUIRE(S != nullptr, "SplittedNormalMatrix not prepared");
    const int m = S->m;
    if (m > 0) hipLaunchKernelGGL(narrow_basis_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, c->stream, m, basis_dev, S->basis.get());
    S->eta_t.ensure((size_t)std::max(m, 1)); S->eta_in.ensure((size_t)std::max(m, 1));
    upload_scaling(c, S, status, colscale);       // scaling of the new basis in pivot order, free positions, weights / matrix of the new N
    IPXK_HIP(hipStreamSynchronize(c->stream));
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
    const bool etas = c->etas_live;
    fill_results(c, {&S->Ut, &S->Lt, &S->Lf, &S->Uf}, done);
    // inverse(B') * rhs
    time_mark(c, kTimeBt, true);
    const double* bt_in = rhs;
    if (etas) {
        hipLaunchKernelGGL(eta_scatter_scale_kernel, dim3(g), dim3(kBlock), 0, s, m, rhs, S->colperm.get(), S->uscale.get(), S->eta_t.get(), done);
        maxvol_apply_etas(c, true, S->eta_t.get());
        hipLaunchKernelGGL(gather_perm_kernel, dim3(g), dim3(kBlock), 0, s, m, S->eta_t.get(), S->colperm.get(), S->eta_in.get(), done);
        bt_in = S->eta_in.get();
    }
    run_sweep(c, S->Ut, !etas, bt_in, done);
    bump_between(c, true, S->Ut.y.get(), done);
    // (the L' sweep also leaves its result in u in the row order of A, for the N N' product)
    run_sweep(c, S->Lt, !etas, S->Ut.y.get(), done, S->row_after_backward.get(), u);
    time_mark(c, kTimeBt, false);
    time_mark(c, kTimeOp, true);
    // N N' of it: A (M D^2) A'
    EpiScale e1{{}, S->Wsplit.get(), c->tcols.get()};
    EpiNormalRows e2{{}, S->Wsplit.get() + n, u, work};
    if (S->real_N) {
        nmatrix_apply(c, S->Wsplit.get() + n, u, work, done);
    } else if (S->masked_values) {
        // the entries of BASIC / fixed columns have weight zero in both passes: masked value arrays, no gathers for them
        launch_spmv<EpiScale, true>(c->Acols, u, e1, nullptr, done, s);
        launch_spmv<EpiNormalRows, true>(c->Arows, c->tcols.get(), e2, nullptr, done, s);
    } else {
        launch_spmv(c->Acols, u, e1, nullptr, done, s);
        launch_spmv(c->Arows, c->tcols.get(), e2, nullptr, done, s);
    }
    time_mark(c, kTimeOp, false);
    // inverse(B) * that (the L sweep reads `work` through rowperm)
    time_mark(c, kTimeB, true);
    run_sweep(c, S->Lf, !etas, work, done);
    bump_between(c, false, S->Lf.y.get(), done);
    run_sweep(c, S->Uf, !etas, S->Lf.y.get(), done);
    if (etas) {
        unpack_result(c, S->Uf, S->colperm.get(), S->eta_t.get());          // by basis position
        maxvol_apply_etas(c, false, S->eta_t.get());
    }
    time_mark(c, kTimeB, false);
    // lhs = result + rhs; zero free positions; dot
    if (etas)
        hipLaunchKernelGGL(split_finish_etas_kernel, dim3(g), dim3(kBlock), 0, s, m, rhs, S->free_mask.get(), S->eta_t.get(), S->colperm.get(),
                           S->uscale.get(), lhs, c->part(kPartCdot), done);
    else
        hipLaunchKernelGGL(split_finish_kernel, dim3(g), dim3(kBlock), 0, s, m, rhs, S->free_mask.get(), S->Uf.y.get(),
                           S->Uf.posof.get(), lhs, c->part(kPartCdot), done);
    return g;
}

// Basis::SolveDense on the fresh, unscaled factors (forrest_tomlin.cc:67-78); rhs may be lhs
void solve_dense_dev(Context* c, const double* rhs, double* lhs, char trans) {
    SplitOperator* S = c->split;
    const int m = S->m;
    hipStream_t s = c->stream;
    const int g = vec_grid(m);
    if (trans == 't' || trans == 'T') {
        double* work = S->w3.get();
        if (c->etas_live) {
            // inverse(B_new') = inverse(B_old') inverse(E'): the eta file first, on a copy (rhs may be lhs, and is not to be changed otherwise)
            IPXK_HIP(hipMemcpyAsync(S->eta_t.get(), rhs, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, s));
            maxvol_apply_etas(c, true, S->eta_t.get());
            rhs = S->eta_t.get();
        }
        hipLaunchKernelGGL(gather_perm_kernel, dim3(g), dim3(kBlock), 0, s, m, rhs, S->colperm.get(), work,
                           (const int*)nullptr);
        fill_results(c, {&S->Ut, &S->Lt}, nullptr);
        run_sweep(c, S->Ut, false, work, nullptr);
        bump_between(c, true, S->Ut.y.get(), nullptr);
        run_sweep(c, S->Lt, false, S->Ut.y.get(), nullptr);
        unpack_result(c, S->Lt, S->rowperm.get(), lhs);            // lhs[rowperm[k]] = solution[k]
    } else {
        fill_results(c, {&S->Lf, &S->Uf}, nullptr);
        run_sweep(c, S->Lf, false, rhs, nullptr);                   // reads rhs[rowperm[.]]
        bump_between(c, false, S->Lf.y.get(), nullptr);
        run_sweep(c, S->Uf, false, S->Lf.y.get(), nullptr);
        unpack_result(c, S->Uf, S->colperm.get(), lhs);            // lhs[colperm[k]] = solution[k]
        if (c->etas_live) maxvol_apply_etas(c, false, lhs);        // inverse(B_new) = inverse(E) inverse(B_old)
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
