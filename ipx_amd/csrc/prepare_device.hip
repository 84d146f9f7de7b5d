// Device-side analysis of the four triangular sweeps of SplittedNormalMatrix::Prepare
// (reference src/splitted_normal_matrix.cc:18-66 hands over L, U; the sweeps are those of
// TriangularSolve, src/sparse_matrix.cc:224-301).
//
// The factors change whenever the basis changes, so the level schedule has to be rebuilt then.  On
// the host that is ~100 ms of pointer chasing at 1M rows; here L and U are uploaded as given and
// everything else runs on the GPU:
//   1. row lists in natural order: U' and L' read the columns directly; the forward sweeps need
//      the row-wise forms -- a stable radix sort of the entries by row index (rocPRIM; entries
//      enumerated in ascending column order for L and in DESCENDING column order for U, the order
//      in which the reference's column loops update a row);
//   2. dependency levels by relaxation level[i] = max(level[dep] + 1) until nothing changes;
//   3. a stable sort of the unknowns (in processing order) by (level, row class): inside a level the
//      rows come grouped by length -- long rows (> 8 entries, 8 lanes each), then rows of 5..8,
//      3..4, 1..2 and 0 entries (one lane each, ELL blocks of width 8 / 4 / 2 / 0) -- so that every
//      wavefront-sized chunk is homogeneous;
//   4. the rows packed chunk by chunk (trisolve.hpp: ChunkDesc).
// The host only sees O(#levels) numbers (group sizes) from which it lays out the chunks and derives
// the launch plan (plan_sweep).  The column-scaled value sets of the U sweeps are derived from the
// packed unscaled ones (rescale_sweeps_device), which is also all that a change of the scaling
// factors alone needs (ipxk_split_rescale).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "context.hpp"
#include "trisolve.hpp"

namespace ipxk {

namespace {

int grid_for(int64_t n) { return (int)std::min<int64_t>(4096, std::max<int64_t>(1, (n + kBlock - 1) / kBlock)); }

#define IPXK_GRID_STRIDE(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ---- step 0: the factors must be what the contract says (lu_update.h:43-60) before any kernel trusts
//      their indices: an out-of-range row index would be an out-of-bounds access on the device
__global__ void validate_factors_kernel(int m, const ipxint* __restrict__ Lp, const ipxint* __restrict__ Li,
                                        const ipxint* __restrict__ Up, const ipxint* __restrict__ Ui, int* bad) {
    IPXK_GRID_STRIDE(k, m) {
        const ipxint l0 = Lp[k], l1 = Lp[k + 1], u0 = Up[k], u1 = Up[k + 1];
        bool ok = l0 <= l1 && u0 < u1;
        if (ok) {
            for (ipxint p = l0; p < l1; p++) ok &= Li[p] > k && Li[p] < m;           // strictly lower
            for (ipxint p = u0; p < u1 - 1; p++) ok &= Ui[p] >= 0 && Ui[p] < k;       // strictly upper
            ok &= Ui[u1 - 1] == k;                                                   // diagonal last
        }
        if (!ok) *bad = 1;
    }
}

// ---- step 1: row lists ---------------------------------------------------------------
__global__ void ut_rows_kernel(int m, const ipxint* __restrict__ Up, const ipxint* __restrict__ Ui,
                               const double* __restrict__ Ux, int* __restrict__ rp, int* __restrict__ ri,
                               double* __restrict__ rx, double* __restrict__ dgn) {
    IPXK_GRID_STRIDE(k, m) {
        const ipxint p0 = Up[k], p1 = Up[k + 1] - 1;      // diagonal last
        const int base = (int)(p0 - k);
        rp[k] = base;
        for (ipxint p = p0; p < p1; p++) {
            const int q = base + (int)(p - p0);
            ri[q] = (int)Ui[p];
            rx[q] = Ux[p];
        }
        dgn[k] = Ux[p1];
        if (k == m - 1) rp[m] = (int)(Up[m] - m);
    }
}

__global__ void lt_rows_kernel(int m, int64_t nz, const ipxint* __restrict__ Lp, const ipxint* __restrict__ Li,
                               const double* __restrict__ Lx, int* __restrict__ rp, int* __restrict__ ri,
                               double* __restrict__ rx) {
    IPXK_GRID_STRIDE(p, nz) { ri[p] = (int)Li[p]; rx[p] = Lx[p]; }
    IPXK_GRID_STRIDE(k, (int64_t)m + 1) rp[k] = (int)Lp[k];
}

__global__ void fill_double_kernel(int64_t n, double v, double* __restrict__ out) {
    IPXK_GRID_STRIDE(i, n) out[i] = v;
}

// sort input for the row-wise form of L: key = row, value = entry, entries in storage order
__global__ void lf_keys_kernel(int m, const ipxint* __restrict__ Lp, const ipxint* __restrict__ Li,
                               int* __restrict__ keys, int* __restrict__ vals, int* __restrict__ colof) {
    IPXK_GRID_STRIDE(j, m) {
        for (ipxint p = Lp[j]; p < Lp[j + 1]; p++) {
            keys[p] = (int)Li[p];
            vals[p] = (int)p;
            colof[p] = (int)j;
        }
    }
}

// ... of U without its diagonal, columns enumerated in DESCENDING order
__global__ void uf_keys_kernel(int m, const ipxint* __restrict__ Up, const ipxint* __restrict__ Ui,
                               int* __restrict__ keys, int* __restrict__ vals, int* __restrict__ colof) {
    const int64_t nzo = Up[m] - m;
    IPXK_GRID_STRIDE(k, m) {
        const ipxint p0 = Up[k], p1 = Up[k + 1] - 1;
        const int64_t qbase = nzo - (Up[k + 1] - (k + 1));   // off-diagonal entries of the columns > k
        for (ipxint p = p0; p < p1; p++) {
            const int64_t q = qbase + (p - p0);
            keys[q] = (int)Ui[p];
            vals[q] = (int)p;
        }
        for (ipxint p = p0; p <= p1; p++) colof[p] = (int)k;
    }
}

// rp[i] = first position of key >= i in the sorted keys (i = 0..dim)
__global__ void lower_bound_kernel(int dim, int64_t nz, const int* __restrict__ sorted_keys, int* __restrict__ rp) {
    IPXK_GRID_STRIDE(i, (int64_t)dim + 1) {
        int64_t lo = 0, hi = nz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (sorted_keys[mid] < (int)i) lo = mid + 1; else hi = mid;
        }
        rp[i] = (int)lo;
    }
}

// out[i] = first position of a key >= bkeys[i] in the sorted keys
__global__ void lower_bound_keys_kernel(int nb, int64_t nz, const int* __restrict__ sorted_keys, const int* __restrict__ bkeys,
                                        int* __restrict__ out) {
    IPXK_GRID_STRIDE(i, nb) {
        const int key = bkeys[i];
        int64_t lo = 0, hi = nz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (sorted_keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        out[i] = (int)lo;
    }
}

__global__ void rows_from_perm_kernel(int64_t nz, const int* __restrict__ perm, const int* __restrict__ colof,
                                      const double* __restrict__ X, int* __restrict__ ri, double* __restrict__ rx) {
    IPXK_GRID_STRIDE(t, nz) {
        const int p = perm[t];
        ri[t] = colof[p];
        rx[t] = X[p];
    }
}

__global__ void u_diag_kernel(int m, const ipxint* __restrict__ Up, const double* __restrict__ Ux,
                              double* __restrict__ dgn) {
    IPXK_GRID_STRIDE(k, m) dgn[k] = Ux[Up[k + 1] - 1];
}

// ---- step 2 (round 3): levels in ONE launch ---------------------------------------------------------------
// The processing order of a sweep (ascending / descending unknowns) is a topological order of its dependency graph,
// so the levels can be computed the way the sweeps themselves run (trisolve.hip): level[] starts at -1 = "not known",
// wavefront w of the W resident ones takes the 64-row chunks w, w + W, ... of the processing order, a row polls the
// levels of its dependencies (L1-bypassing loads) until all are known and stores max + 1 -- the value is the flag.  The
// lowest unfinished row depends only on finished ones, so the launch always makes progress, whatever the placement;
// every workgroup must be resident (the grid is sized from the occupancy query).  Rows inside one chunk may depend on
// each other: a lane never spins on its own, the wavefront loops over "whoever is ready now".  Rows of more than
// kLevelLongRow entries are evaluated by the whole wavefront, lowest first.  A wait that exceeds the spin budget
// raises `abort` and the caller falls back to the relaxation launches below.
// (Replaces 20-60 relaxation launches per sweep: 9.1 of the 12.7 ms of kernel time of a Prepare at 1M rows.)
constexpr int kLevelLongRow = 32;
constexpr int kLevelSpin = 1 << 20;
__global__ __launch_bounds__(kBlock) void level_sweep_kernel(int dim, int ascending, const int* __restrict__ rp, const int* __restrict__ ri,
                                                             int* level, int* abort_flag) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), NW = gridDim.x * (kBlock / 64);
    const int nchunks = (dim + 63) / 64;
    for (int c = gw; c < nchunks; c += NW) {
        const int t = c * 64 + lane;
        const int i = t < dim ? (ascending ? t : dim - 1 - t) : -1;
        const int p0 = i >= 0 ? rp[i] : 0, p1 = i >= 0 ? rp[i + 1] : 0;
        const bool is_long = p1 - p0 > kLevelLongRow;
        bool finished = i < 0;
        if (i >= 0 && p1 == p0) {
            __hip_atomic_store(level + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            finished = true;
        }
        for (int spins = 0;; spins++) {
            if (!finished && !is_long) {
                int lv = 0;
                bool ready = true;
                for (int p = p0; p < p1; p++) {
                    const int l = __hip_atomic_load(level + ri[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (l < 0) { ready = false; break; }
                    lv = l + 1 > lv ? l + 1 : lv;
                }
                if (ready) {
                    __hip_atomic_store(level + i, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    finished = true;
                }
            }
            const unsigned long long lm = __ballot(!finished && is_long);
            if (lm) {                                             // the lowest unfinished long row, by all lanes
                const int src = __ffsll((long long)lm) - 1;
                const int q0 = __shfl(p0, src, 64), q1 = __shfl(p1, src, 64);
                int lv = 0;
                bool ready = true;
                for (int p = q0 + lane; p < q1; p += 64) {
                    const int l = __hip_atomic_load(level + ri[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (l < 0) ready = false;
                    else lv = l + 1 > lv ? l + 1 : lv;
                }
                if (__all(ready)) {
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) {
                        const int o = __shfl_xor(lv, d, 64);
                        lv = o > lv ? o : lv;
                    }
                    if (lane == src) {
                        __hip_atomic_store(level + i, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        finished = true;
                    }
                }
            }
            if (!__any(!finished)) break;
            if (spins > kLevelSpin ||
                ((spins & 255) == 255 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// ---- step 2: levels -----------------------------------------------------------------------
// One pass of level[i] = max(level[dep] + 1) over the rows of at most kRelaxLong entries.  Two things let a
// single launch settle many levels: the threads take the unknowns in PROCESSING order (descending sweeps from
// the far end), so that the workgroups dispatched first hold the dependencies of those dispatched later; and
// a workgroup repeats its pass while one of its own unknowns moved.
constexpr int kRelaxInner = 64;
constexpr int kRelaxLong = 64;
constexpr int kRelaxLongThreads = 1024;
__global__ __launch_bounds__(kBlock) void relax_levels_kernel(int dim, int ascending, const int* __restrict__ rp,
                                                              const int* __restrict__ ri, int* level, int* changed) {
    for (int64_t t0 = (int64_t)blockIdx.x * kBlock; t0 < dim; t0 += (int64_t)gridDim.x * kBlock) {
        const int64_t t = t0 + threadIdx.x;
        int i = t < dim ? (ascending ? (int)t : dim - 1 - (int)t) : -1;
        if (i >= 0 && rp[i + 1] - rp[i] > kRelaxLong) i = -1;          // relax_long_rows_kernel's
        int mine = i >= 0 ? level[i] : 0;
        bool moved = false;
        for (int inner = 0; inner < kRelaxInner; inner++) {
            bool ch = false;
            if (i >= 0) {
                int lv = 0;
                for (int p = rp[i]; p < rp[i + 1]; p++) {
                    const int l = __hip_atomic_load(level + ri[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                    lv = l > lv ? l : lv;
                }
                if (lv > mine) {
                    mine = lv;
                    __hip_atomic_store(level + i, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ch = true;
                }
            }
            moved |= ch;
            if (!__syncthreads_or(ch ? 1 : 0)) break;
        }
        if (moved) *changed = 1;
    }
}
// The rows of more than kRelaxLong entries (the rows of an LU's dense bump: a chain as long as the bump), one
// after the other in processing order by ONE workgroup whose threads share a row's entries: a chain among them
// settles in a single launch instead of one launch per link.
__global__ __launch_bounds__(kRelaxLongThreads) void relax_long_rows_kernel(int nlong, const int* __restrict__ rows,
                                                                             const int* __restrict__ rp,
                                                                             const int* __restrict__ ri, int* level,
                                                                             int* changed) {
    __shared__ int red[kRelaxLongThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool moved = false;
    for (int q = 0; q < nlong; q++) {
        const int i = rows[q];
        int lv = 0;
        for (int p = rp[i] + tid; p < rp[i + 1]; p += kRelaxLongThreads) {
            const int l = __hip_atomic_load(level + ri[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            lv = l > lv ? l : lv;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(lv, d, 64);
            lv = o > lv ? o : lv;
        }
        if (lane == 0) red[wave] = lv;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < kRelaxLongThreads / 64; w++) lv = red[w] > lv ? red[w] : lv;
            if (lv > level[i]) {
                __hip_atomic_store(level + i, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                moved = true;
            }
        }
        __syncthreads();          // the next row may read this one's level; red is reused
    }
    if (tid == 0 && moved) *changed = 1;
}
// the same rows in parallel, one wavefront each: enough where they do not depend on each other
__global__ __launch_bounds__(kBlock) void relax_long_rows_parallel_kernel(int nlong, const int* __restrict__ rows,
                                                                           const int* __restrict__ rp,
                                                                           const int* __restrict__ ri, int* level,
                                                                           int* changed) {
    const int lane = threadIdx.x & 63;
    for (int q = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); q < nlong; q += gridDim.x * (kBlock / 64)) {
        const int i = rows[q];
        int lv = 0;
        for (int p = rp[i] + lane; p < rp[i + 1]; p += 64) {
            const int l = __hip_atomic_load(level + ri[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            lv = l > lv ? l : lv;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(lv, d, 64);
            lv = o > lv ? o : lv;
        }
        if (lane == 0 && lv > level[i]) {
            __hip_atomic_store(level + i, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *changed = 1;
        }
    }
}
__global__ void relax_long_flag_kernel(int dim, int ascending, const int* __restrict__ rp, int* __restrict__ flag) {
    IPXK_GRID_STRIDE(t, dim) {
        const int i = ascending ? (int)t : dim - 1 - (int)t;
        flag[t] = rp[i + 1] - rp[i] > kRelaxLong ? 1 : 0;
    }
}
__global__ void relax_long_list_kernel(int dim, int ascending, const int* __restrict__ flag, const int* __restrict__ rank,
                                       int* __restrict__ rows) {
    IPXK_GRID_STRIDE(t, dim)
        if (flag[t]) rows[rank[t]] = ascending ? (int)t : dim - 1 - (int)t;
}

// ---- step 3: order by (level, descending row length) -------------------------------------------------
__global__ void level_keys_kernel(int dim, int ascending, int by_length, const int* __restrict__ level,
                                  const int* __restrict__ rp, int* __restrict__ keys, int* __restrict__ vals) {
    IPXK_GRID_STRIDE(t, dim) {
        const int i = ascending ? (int)t : dim - 1 - (int)t;
        const int len = rp[i + 1] - rp[i];
        // by_length: descending length inside a level; otherwise only long rows before short ones, each
        // part in processing order
        const int sub = by_length ? 255 - (len < 255 ? len : 255) : (len > kShortRow ? 0 : 255 - kShortRow);
        keys[t] = (level[i] << kLenKeyBits) | sub;
        vals[t] = i;
    }
}

// ---- step 4: rows into chunks ---------------------------------------------------------------
// per level l: sorted unknowns [lstart[2l], lstart[2l+1]) are its long rows, [lstart[2l+1], lstart[2l+2]) its
// short ones; lpos[2l], lpos[2l+1] = first position of either part
__global__ void place_kernel(int dim, const int* __restrict__ sorted_key, const int* __restrict__ sorted_unknown,
                             const int* __restrict__ lstart, const int* __restrict__ lpos, const int* __restrict__ lsub,
                             const int* __restrict__ rp, const double* __restrict__ dgn, int* __restrict__ order,
                             int* __restrict__ posof, double* __restrict__ diag, int* __restrict__ len, int* bad_len) {
    IPXK_GRID_STRIDE(t, dim) {
        const int key = sorted_key[t], i = sorted_unknown[t];
        const int l = key >> kLenKeyBits;
        const int part = (int)t >= lstart[2 * l + 1] ? 1 : 0;
        const int pos = lpos[2 * l + part] + ((int)t - lstart[2 * l + part]);
        order[pos] = i;
        posof[i] = pos;
        diag[pos] = dgn[i];
        const int rowlen = rp[i + 1] - rp[i];
        if (rowlen >= (1 << kLenBits)) *bad_len = 1;                // does not fit the len word: refused by the host
        len[pos] = (rowlen & ((1 << kLenBits) - 1)) | (lsub[l] << kLenBits);     // lsub: the level's place inside a merged chunk
    }
}

// entry slots a chunk needs: 64 * (its longest row, rounded up to whole steps of 8 for long rows);
// also completes the descriptor's width
__global__ void chunk_size_kernel(int nchunks, ChunkDesc* __restrict__ chunks, const int* __restrict__ len,
                                  int* __restrict__ size, int* too_long) {
    IPXK_GRID_STRIDE(c, nchunks) {
        ChunkDesc d = chunks[c];
        int mx = 0;
        for (int q = 0; q < d.npos; q++) mx = max(mx, len[d.pos0 + q] & ((1 << kLenBits) - 1));
        if (d.sub > 1 && d.width < 0 && mx > 64) *too_long = 1;     // merged long rows must fit one round of 64 entries
        if (d.width < 0) { const int steps = (mx + kLongLanes - 1) / kLongLanes; d.width = -max(steps, 1); size[c] = max(steps, 1) * 64; }
        else { d.width = mx; size[c] = mx * 64; }
        chunks[c].width = d.width;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) size[nchunks] = 0;
}

__global__ void check_monotone_kernel(int n, const int* __restrict__ a, int* bad) {
    IPXK_GRID_STRIDE(c, n) if (a[c + 1] < a[c] || a[c] < 0) *bad = 1;
}
__global__ void chunk_ent0_kernel(int nchunks, ChunkDesc* __restrict__ chunks, const int* __restrict__ ent0) {
    IPXK_GRID_STRIDE(c, nchunks) chunks[c].ent0 = ent0[c];
}

// one wavefront per chunk copies the rows of its chunk into the chunk's block; dependencies become positions
__global__ __launch_bounds__(kBlock) void pack_entries_kernel(int nchunks, const ChunkDesc* __restrict__ chunks,
                                                              const int* __restrict__ order, const int* __restrict__ posof,
                                                              const int* __restrict__ rp,
                                                              const int* __restrict__ ri, const double* __restrict__ rx,
                                                              int* __restrict__ idx, double* __restrict__ val) {
    const int lane = threadIdx.x & 63;
    for (int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); c < nchunks; c += gridDim.x * (kBlock / 64)) {
        const ChunkDesc d = chunks[c];
        const bool ell = d.width >= 0;
        const int i = order[ell ? d.pos0 + lane : d.pos0 + (lane >> 3)];
        if (i < 0) continue;
        const int p0 = rp[i], n = rp[i + 1] - p0;
        for (int e = ell ? 0 : (lane & 7); e < n; e += ell ? 1 : kLongLanes) {
            const int64_t slot = ell ? (int64_t)d.ent0 + e * 64 + lane : (int64_t)d.ent0 + (e >> 3) * 64 + lane;
            idx[slot] = posof[ri[p0 + e]];
            val[slot] = rx[p0 + e];
        }
    }
}

// ---- column-scaled value sets (splitted_normal_matrix.cc:30-39) -----------------------------------
// MODE 1 (U' sweep): unknown k gathers column k of U: every entry and the diagonal times uscale[k]
// MODE 2 (U sweep): unknown i walks row i of U: entry (i, k) times uscale[k], the diagonal times uscale[i]
template <int MODE>
__global__ __launch_bounds__(kBlock) void rescale_kernel(SweepView S, const int* __restrict__ order, int nchunks,
                                                         const double* __restrict__ uscale,
                                                         double* __restrict__ valS, double* __restrict__ diagS) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const ChunkDesc d = S.chunks[c];
    const bool ell = d.width >= 0;
    const int pos = ell ? d.pos0 + lane : d.pos0 + (lane >> 3), gl = lane & 7;
    const int r = order[pos];
    if (r < 0) { if (ell || gl == 0) diagS[pos] = 1.0; return; }
    const double own = uscale[r];
    if (ell || gl == 0) diagS[pos] = S.diag[pos] * own;
    const int len = S.len[pos] & ((1 << kLenBits) - 1);
    for (int e = ell ? 0 : gl; e < len; e += ell ? 1 : kLongLanes) {
        const int64_t slot = ell ? (int64_t)d.ent0 + e * 64 + lane : (int64_t)d.ent0 + (e >> 3) * 64 + lane;
        valS[slot] = S.val[slot] * (MODE == 1 ? own : uscale[order[S.idx[slot]]]);
    }
}

struct Scratch {   // reused by the four sweeps of a Prepare and kept from one Prepare to the next (Context::prepare_host; grow-only:
                   // hipMalloc / hipFree of a dozen buffers cost more than the kernels that used them)
    DevBuf<int> keys, vals, keys2, vals2, colof, level, lstart, lpos, lsub, csize, cent0;
    DevBuf<int> changed, too_long, bkeys, bad;
    DevBuf<unsigned char> tmp;
    DevBuf<int> rp, ri;
    DevBuf<double> rx, dgn;
    int level_grid = -1;     // workgroups of level_sweep_kernel: all resident on this context's device (-1: not asked yet)
    int* h_flag = nullptr;   // pinned
    ~Scratch() { if (h_flag) (void)hipHostFree(h_flag); }
};

void sort_pairs(Scratch& W, int64_t n, int end_bit, hipStream_t s) {
    size_t bytes = 0;
    IPXK_HIP(rocprim::radix_sort_pairs(nullptr, bytes, W.keys.get(), W.keys2.get(), W.vals.get(), W.vals2.get(),
                                       (size_t)n, 0u, (unsigned)end_bit, s));
    if (W.tmp.size() < bytes) W.tmp.resize(bytes);
    IPXK_HIP(rocprim::radix_sort_pairs(W.tmp.get(), bytes, W.keys.get(), W.keys2.get(), W.vals.get(), W.vals2.get(),
                                       (size_t)n, 0u, (unsigned)end_bit, s));
}

void exclusive_scan(Scratch& W, const int* in, int* out, size_t n, hipStream_t s) {
    size_t bytes = 0;
    IPXK_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, n, rocprim::plus<int>(), s));
    if (W.tmp.size() < bytes) W.tmp.resize(bytes);
    IPXK_HIP(rocprim::exclusive_scan(W.tmp.get(), bytes, in, out, 0, n, rocprim::plus<int>(), s));
}

int bits_for(int64_t n) {   // bits needed for values in [0, n)
    int b = 1;
    while ((int64_t(1) << b) < n) b++;
    return b;
}

// Deep dependency graphs: the relaxation needs as many rounds as there are levels, each a pass over
// all entries.  After kMaxRelaxLaunches launches the levels are computed by one sequential O(nnz) scan
// on the host instead (the unknowns' processing order is a topological order) and uploaded.
constexpr int kMaxRelaxLaunches = 1024;  // ~10 ms at 1M rows

// levels, order, packed rows and launch plan of one sweep from its natural-order row list
template <class HostLevels>
void finish_sweep(Context* c, Scratch& W, Sweep& S, bool level_launches, int dim, int64_t nz, bool ascending, bool running,
                  int scale_mode, HostLevels&& host_levels) {
    hipStream_t s = c->stream;
    S.dim = dim;
    S.running = running;
    S.scale_mode = scale_mode;
    // 2. levels: one sync-free launch (level_sweep_kernel); should it give up, rounds of relaxation launches, twice as
    // many each time (one host round trip per round)
    W.level.ensure(std::max(dim, 1));
    DevBuf<int>& changed = W.changed;
    changed.ensure(1);
    if (!W.h_flag) IPXK_HIP(hipHostMalloc(reinterpret_cast<void**>(&W.h_flag), sizeof(int)));
    bool have_levels = false;
    if (dim > 0 && !(getenv("IPXK_LEVEL_SWEEP") && getenv("IPXK_LEVEL_SWEEP")[0] == '0')) {
        if (W.level_grid < 0) {                                   // per context, i.e. per device (as the sweeps' grid, trisolve.hip)
            int dev = 0, per_cu = 0;
            hipDeviceProp_t prop;
            W.level_grid = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, level_sweep_kernel, kBlock, 0) == hipSuccess)
                // one block per CU is held back from what the query reports, at most 4 are used
                W.level_grid = prop.multiProcessorCount * std::max(0, std::min(per_cu - 1, 4));
        }
        const int resident = W.level_grid;
        if (resident > 0) {
            IPXK_HIP(hipMemsetAsync(W.level.get(), 0xff, sizeof(int) * dim, s));
            IPXK_HIP(hipMemsetAsync(changed.get(), 0, sizeof(int), s));
            const int grid = std::max(1, std::min(resident, (dim + kBlock - 1) / kBlock));
            hipLaunchKernelGGL(level_sweep_kernel, dim3(grid), dim3(kBlock), 0, s, dim, ascending ? 1 : 0, W.rp.get(), W.ri.get(),
                               W.level.get(), changed.get());
            IPXK_HIP(hipMemcpyAsync(W.h_flag, changed.get(), sizeof(int), hipMemcpyDeviceToHost, s));
            IPXK_HIP(hipStreamSynchronize(s));
            have_levels = *W.h_flag == 0;
            if (getenv("IPXK_VERBOSE")) fprintf(stderr, "ipxk: levels by one sync-free launch%s\n", have_levels ? "" : ": gave up, relaxation instead");
        }
    }
    if (!have_levels) IPXK_HIP(hipMemsetAsync(W.level.get(), 0, sizeof(int) * std::max(dim, 1), s));
    // long rows, in processing order
    int nlong = 0;
    DevBuf<int> long_rows;
    if (dim > 0 && !have_levels) {
        W.keys.ensure((size_t)dim); W.vals.ensure((size_t)dim);
        hipLaunchKernelGGL(relax_long_flag_kernel, dim3(grid_for(dim)), dim3(kBlock), 0, s, dim, ascending ? 1 : 0, W.rp.get(), W.keys.get());
        exclusive_scan(W, W.keys.get(), W.vals.get(), (size_t)dim, s);
        int last[2] = {0, 0};
        IPXK_HIP(hipMemcpyAsync(&last[0], W.keys.get() + dim - 1, sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipMemcpyAsync(&last[1], W.vals.get() + dim - 1, sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        nlong = last[0] + last[1];
        if (nlong > 0) {
            long_rows.resize((size_t)nlong);
            hipLaunchKernelGGL(relax_long_list_kernel, dim3(grid_for(dim)), dim3(kBlock), 0, s, dim, ascending ? 1 : 0, W.keys.get(),
                               W.vals.get(), long_rows.get());
        }
    }
    // a cycle = one pass over the short rows + (if any) one over the long ones, a wavefront each; every batch
    // of cycles (twice as many each time, one host round trip per batch) starts with the sequential pass over
    // the long rows, which settles a chain among them at once
    for (int launched = 0, batch = nlong > 0 ? 4 : 8; dim > 0 && !have_levels; batch = std::min(batch * 2, 64)) {
        if (launched >= kMaxRelaxLaunches) {
            std::vector<int> lv((size_t)dim, 0);
            host_levels(lv);
            W.level.upload(lv, s);
            break;
        }
        IPXK_HIP(hipMemsetAsync(changed.get(), 0, sizeof(int), s));
        for (int r = 0; r < batch; r++) {
            if (r == batch - 1) IPXK_HIP(hipMemsetAsync(changed.get(), 0, sizeof(int), s));   // only the last cycle decides
            hipLaunchKernelGGL(relax_levels_kernel, dim3(grid_for(dim)), dim3(kBlock), 0, s, dim, ascending ? 1 : 0, W.rp.get(),
                               W.ri.get(), W.level.get(), changed.get());
            if (nlong > 0 && r == 0)
                hipLaunchKernelGGL(relax_long_rows_kernel, dim3(1), dim3(kRelaxLongThreads), 0, s, nlong, long_rows.get(), W.rp.get(),
                                   W.ri.get(), W.level.get(), changed.get());
            else if (nlong > 0)
                hipLaunchKernelGGL(relax_long_rows_parallel_kernel, dim3((nlong + 3) / 4), dim3(kBlock), 0, s, nlong, long_rows.get(),
                                   W.rp.get(), W.ri.get(), W.level.get(), changed.get());
        }
        launched += batch;
        IPXK_HIP(hipMemcpyAsync(W.h_flag, changed.get(), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        if (*W.h_flag == 0) {
            if (getenv("IPXK_VERBOSE")) fprintf(stderr, "ipxk: levels settled within %d relaxation cycles (%d long rows)\n", launched, nlong);
            break;
        }
    }
    // 3. stable sort of the unknowns in processing order by (level, descending length)
    const size_t need = (size_t)std::max<int64_t>(std::max<int64_t>(dim, nz), 1);
    W.keys.ensure(need); W.vals.ensure(need); W.keys2.ensure(need); W.vals2.ensure(need);
    int nlev = 0;
    std::vector<int> lstart;     // [2*nlev + 1]: first sorted unknown of (level, long part / short part)
    if (dim > 0) {
        const char* ord = getenv("IPXK_SWEEP_ORDER");
        const int by_length = ord && std::string(ord) == "length" ? 1 : 0;
        hipLaunchKernelGGL(level_keys_kernel, dim3(grid_for(dim)), dim3(kBlock), 0, s, dim, ascending ? 1 : 0, by_length,
                           W.level.get(), W.rp.get(), W.keys.get(), W.vals.get());
        sort_pairs(W, dim, 31, s);
        // the last sorted key belongs to the deepest level
        IPXK_HIP(hipMemcpyAsync(W.h_flag, W.keys2.get() + (dim - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        nlev = (*W.h_flag >> kLenKeyBits) + 1;
        IPXK_REQUIRE(nlev < (1 << (31 - kLenKeyBits)), "dependency graph too deep for the 32-bit sort key");
        // boundaries: key (l << 8) starts level l, key (l << 8 | 255 - kShortRow) starts its short rows
        std::vector<int> bkeys((size_t)2 * nlev + 1);
        for (int l = 0; l < nlev; l++) { bkeys[2 * l] = l << kLenKeyBits; bkeys[2 * l + 1] = (l << kLenKeyBits) | (255 - kShortRow); }
        bkeys[2 * nlev] = nlev << kLenKeyBits;
        DevBuf<int>& dk = W.bkeys;
        dk.upload(bkeys, s);
        W.lstart.ensure(bkeys.size());
        hipLaunchKernelGGL(lower_bound_keys_kernel, dim3(grid_for((int64_t)bkeys.size())), dim3(kBlock), 0, s, (int)bkeys.size(),
                           (int64_t)dim, W.keys2.get(), dk.get(), W.lstart.get());
        lstart.resize(bkeys.size());
        W.lstart.download(lstart.data(), lstart.size(), s);
        IPXK_HIP(hipStreamSynchronize(s));
    }
    const bool merge_levels = [] { const char* e = getenv("IPXK_SWEEP_MERGE"); return !(e && e[0] == '0'); }();
    size_t np1 = 1, ns1 = 1;
    int npos = 0;
    int64_t slots = 0;
    auto layout = [&](bool merge_long) -> bool {
        // chunk layout (host arithmetic over the part sizes): positions and chunk -> position; widths and
        // entry offsets are completed on the device.  Consecutive tiny levels whose rows are all short (<= 64 in
        // total) or all long (<= 8 in total) share one MERGED chunk (trisolve.hpp).
        std::vector<int> lpos((size_t)2 * std::max(nlev, 1), 0), lsub((size_t)std::max(nlev, 1), 0);
        std::vector<ChunkDesc> chunks;
        S.level_chunk.assign((size_t)nlev + 1, 0);
        S.level_width.assign((size_t)nlev, 0);
        int64_t pos = 0;
        auto counts = [&](int l, int& nl, int& ns) { nl = lstart[2 * l + 1] - lstart[2 * l]; ns = lstart[2 * l + 2] - lstart[2 * l + 1]; };
        for (int l = 0; l < nlev;) {
            int nl, ns;
            counts(l, nl, ns);
            // how many levels starting at l can share a chunk
            int last = l, total = nl + ns;
            const bool long_only = ns == 0 && nl > 0, short_only = nl == 0 && ns > 0;
            const int cap = long_only ? kLongLanes : 64;
            if (merge_levels && (long_only ? merge_long : short_only) && total <= cap) {
                while (last + 1 < nlev && last + 1 - l < kMaxSubLevels) {
                    int nl2, ns2;
                    counts(last + 1, nl2, ns2);
                    const bool same = long_only ? (ns2 == 0 && nl2 > 0) : (nl2 == 0 && ns2 > 0);
                    if (!same || total + nl2 + ns2 > cap) break;
                    total += nl2 + ns2;
                    last++;
                }
            }
            if (last > l) {                                   // merged chunk for levels l..last
                const int c = (int)chunks.size();
                int p = (int)pos;
                for (int t = l; t <= last; t++) {
                    int a, b2;
                    counts(t, a, b2);
                    S.level_chunk[t] = c;
                    S.level_width[t] = a + b2;
                    lsub[t] = t - l;
                    lpos[2 * t] = p; lpos[2 * t + 1] = p;    // one of the two parts is empty
                    p += a + b2;
                }
                chunks.push_back({(int)pos, 0, long_only ? -1 : 0, cap, last - l + 1, 0, 0, 0});
                pos += cap;
                l = last + 1;
                continue;
            }
            S.level_chunk[l] = (int)chunks.size();
            S.level_width[l] = nl + ns;
            lpos[2 * l] = (int)pos;
            const int nlpad = (nl + kLongLanes - 1) / kLongLanes * kLongLanes;
            for (int q = 0; q < nlpad; q += kLongLanes) chunks.push_back({(int)pos + q, 0, -1, kLongLanes, 1, 0, 0, 0});
            pos += nlpad;
            lpos[2 * l + 1] = (int)pos;
            const int nspad = (ns + 63) / 64 * 64;
            for (int q = 0; q < nspad; q += 64) chunks.push_back({(int)pos + q, 0, 0, 64, 1, 0, 0, 0});
            pos += nspad;
            IPXK_REQUIRE(pos < (int64_t(1) << 31), "packed factor exceeds 32-bit offsets");
            l++;
        }
        S.level_chunk[nlev] = (int)chunks.size();
        npos = (int)pos;
        const int nchunks = (int)chunks.size();
        S.nlevels = nlev;
        S.npos = npos;
        S.nchunks = nchunks;
        np1 = (size_t)std::max(npos, 1);
        const size_t nc1 = (size_t)std::max(nchunks, 1);
        S.chunks.ensure(nc1);
        if (nchunks) S.chunks.upload(chunks.data(), chunks.size(), s);
        S.merged_prefix.assign((size_t)nchunks + 1, 0);
        for (int c = 0; c < nchunks; c++) S.merged_prefix[c + 1] = S.merged_prefix[c] + (chunks[c].sub > 1 ? 1 : 0);
        S.order.ensure(np1); S.diag.ensure(np1); S.len.ensure(np1); S.src.ensure(np1); S.y.ensure(np1);
        S.posof.ensure((size_t)std::max(dim, 1));
        IPXK_HIP(hipMemsetAsync(S.order.get(), 0xff, sizeof(int) * np1, s));
        IPXK_HIP(hipMemsetAsync(S.len.get(), 0, sizeof(int) * np1, s));
        if (npos > 0) hipLaunchKernelGGL(fill_double_kernel, dim3(grid_for(npos)), dim3(kBlock), 0, s, (int64_t)npos, 1.0, S.diag.get());
        slots = 0;
        if (dim > 0) {
            W.lpos.ensure(lpos.size()); W.lsub.ensure(lsub.size());
            W.lpos.upload(lpos, s);
            W.lsub.upload(lsub, s);
            DevBuf<int>& too_long = W.too_long;               // [0] merged long row beyond one round, [1] row beyond the len word, [2] slot count overflow
            too_long.ensure(3);
            IPXK_HIP(hipMemsetAsync(too_long.get(), 0, 3 * sizeof(int), s));
            hipLaunchKernelGGL(place_kernel, dim3(grid_for(dim)), dim3(kBlock), 0, s, dim, W.keys2.get(), W.vals2.get(),
                               W.lstart.get(), W.lpos.get(), W.lsub.get(), W.rp.get(), W.dgn.get(), S.order.get(), S.posof.get(),
                               S.diag.get(), S.len.get(), too_long.get() + 1);
            W.csize.ensure(nc1 + 1); W.cent0.ensure(nc1 + 1);
            hipLaunchKernelGGL(chunk_size_kernel, dim3(grid_for(nchunks)), dim3(kBlock), 0, s, nchunks, S.chunks.get(),
                               S.len.get(), W.csize.get(), too_long.get());
            exclusive_scan(W, W.csize.get(), W.cent0.get(), (size_t)nchunks + 1, s);
            // (the 32-bit running sum must not wrap: entry offsets are ints)
            hipLaunchKernelGGL(check_monotone_kernel, dim3(grid_for(nchunks)), dim3(kBlock), 0, s, nchunks, W.cent0.get(), too_long.get() + 2);
            int tl[3] = {0, 0, 0};
            too_long.download(tl, 3, s);
            if (tl[1] || tl[2])
                throw Error(IPXK_E_UNSUPPORTED, tl[1] ? "a row of a triangular factor has 2^24 entries or more: beyond the packed sweep layout"
                                                      : "the packed triangular factor needs 2^31 entry slots or more: beyond the packed sweep layout");
            if (tl[0]) return false;                          // a merged long row exceeds one round: lay out again without
            hipLaunchKernelGGL(chunk_ent0_kernel, dim3(grid_for(nchunks)), dim3(kBlock), 0, s, nchunks, S.chunks.get(), W.cent0.get());
            IPXK_HIP(hipMemcpyAsync(W.h_flag, W.cent0.get() + nchunks, sizeof(int), hipMemcpyDeviceToHost, s));
            IPXK_HIP(hipStreamSynchronize(s));
            slots = *W.h_flag;
            IPXK_REQUIRE(slots >= 0, "packed factor exceeds 32-bit offsets");
        }
        S.nentries = slots;
        ns1 = (size_t)std::max<int64_t>(slots, 1);
        S.idx.ensure(ns1); S.val.ensure(ns1);
        IPXK_HIP(hipMemsetAsync(S.idx.get(), 0, sizeof(int) * ns1, s));
        IPXK_HIP(hipMemsetAsync(S.val.get(), 0, sizeof(double) * ns1, s));
        if (nchunks > 0)
            hipLaunchKernelGGL(pack_entries_kernel, dim3(std::min(grid_for((int64_t)nchunks * 64), 2048)), dim3(kBlock), 0, s, nchunks,
                               S.chunks.get(), S.order.get(), S.posof.get(), W.rp.get(), W.ri.get(), W.rx.get(), S.idx.get(),
                               S.val.get());
        return true;
    };
    // long rows are merged only on request (IPXK_SWEEP_MERGE=long): every round of a merged chunk recomputes
    // all its rows, which for rows of 10-40 entries costs more than the hand-offs it saves (C3: +9 us per
    // forward pair)
    const bool want_long = [] { const char* e = getenv("IPXK_SWEEP_MERGE"); return e && std::string(e) == "long"; }();
    if (!layout(want_long)) {
        const bool ok = layout(false);
        IPXK_REQUIRE(ok, "chunk layout failed");
    }
    if (scale_mode) { S.valS.ensure(ns1); S.diagS.ensure(np1); }
    build_sweep_blocks(c, S, level_launches);
    plan_sweep(S, level_launches);
    IPXK_HIP(hipStreamSynchronize(s));    // host vectors uploaded above go out of scope
    if (getenv("IPXK_SWEEP_STATS")) {
        fprintf(stderr, "sweep(%s,%s): %d levels, %d positions for %d unknowns, %d chunks, %lld entry slots for %lld entries\n",
                running ? "fwd" : "trans", ascending ? "asc" : "desc", nlev, npos, dim, S.nchunks, (long long)slots, (long long)nz);
        for (const Sweep::Launch& L : S.plan)
            fprintf(stderr, "   launch chunks %d..%d %s\n", L.c0, L.c1,
                    L.kind == Sweep::kOneXcd ? "one XCD" : "all XCDs");
    }
}

}  // namespace

struct PrepareHost { Scratch W; };           // workspaces of Prepare kept by the context (context.hpp)
void destroy_prepare_host(PrepareHost* p) { delete p; }

void rescale_sweeps_device(Context* c, SplitOperator* S) {
    hipStream_t s = c->stream;
    for (Sweep* W : {&S->Ut, &S->Uf}) {
        if (W->nchunks == 0) continue;
        const int g = (W->nchunks + kBlock / 64 - 1) / (kBlock / 64);
        if (W->scale_mode == 1)
            hipLaunchKernelGGL(rescale_kernel<1>, dim3(g), dim3(kBlock), 0, s, W->view(false), W->order.get(), W->nchunks,
                               S->uscale.get(), W->valS.get(), W->diagS.get());
        else
            hipLaunchKernelGGL(rescale_kernel<2>, dim3(g), dim3(kBlock), 0, s, W->view(false), W->order.get(), W->nchunks,
                               S->uscale.get(), W->valS.get(), W->diagS.get());
    }
    IPXK_HIP(hipGetLastError());
}

// host copies of the index arrays of L and U for the fall-back level computation of finish_sweep: the caller's
// arrays, or -- factors that were computed on the device -- downloaded when (and only when) the fall-back runs
struct HostIndices {
    const ipxint *Lp = nullptr, *Li = nullptr, *Up = nullptr, *Ui = nullptr;
    std::vector<ipxint> own[4];
    void fetch(const DeviceFactors& F, int m, hipStream_t s) {
        if (Lp) return;
        own[0].resize((size_t)m + 1); own[1].resize((size_t)std::max<int64_t>(F.nzL, 1));
        own[2].resize((size_t)m + 1); own[3].resize((size_t)std::max<int64_t>(F.nzU, 1));
        IPXK_HIP(hipMemcpyAsync(own[0].data(), F.Lp, ((size_t)m + 1) * sizeof(ipxint), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipMemcpyAsync(own[1].data(), F.Li, (size_t)F.nzL * sizeof(ipxint), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipMemcpyAsync(own[2].data(), F.Up, ((size_t)m + 1) * sizeof(ipxint), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipMemcpyAsync(own[3].data(), F.Ui, (size_t)F.nzU * sizeof(ipxint), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        Lp = own[0].data(); Li = own[1].data(); Up = own[2].data(); Ui = own[3].data();
    }
};

void analyse_sweeps_device(Context* c, SplitOperator* S, const ipxint* Lp, const ipxint* Li, const double* Lx,
                           const ipxint* Up, const ipxint* Ui, const double* Ux) {
    hipStream_t s = c->stream;
    const int m = S->m;
    // factors as given
    DevBuf<ipxint> dLp, dLi, dUp, dUi;
    DevBuf<double> dLx, dUx;
    const int64_t nzL = Lp[m], nzU = Up[m];
    dLp.upload(Lp, (size_t)m + 1, s); dUp.upload(Up, (size_t)m + 1, s);
    dLi.upload(Li, (size_t)nzL, s);   dLx.upload(Lx, (size_t)nzL, s);
    dUi.upload(Ui, (size_t)nzU, s);   dUx.upload(Ux, (size_t)nzU, s);
    const DeviceFactors F{dLp.get(), dLi.get(), dUp.get(), dUi.get(), dLx.get(), dUx.get(), nzL, nzU};
    analyse_sweeps_resident(c, S, F, Lp, Li, Up, Ui);
}

void analyse_sweeps_resident(Context* c, SplitOperator* S, const DeviceFactors& F, const ipxint* hLp, const ipxint* hLi,
                             const ipxint* hUp, const ipxint* hUi) {
    hipStream_t s = c->stream;
    const int m = S->m;
    const int64_t nzL = F.nzL, nzU = F.nzU, nzUo = nzU - m;
    const bool verbose = getenv("IPXK_VERBOSE") != nullptr;
    auto now = [&] {
        if (verbose) (void)hipStreamSynchronize(s);
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    const double t0 = now();
    HostIndices H;
    H.Lp = hLp; H.Li = hLi; H.Up = hUp; H.Ui = hUi;
    struct { const ipxint* p; const ipxint* get() const { return p; } } dLp{F.Lp}, dLi{F.Li}, dUp{F.Up}, dUi{F.Ui};
    struct { const double* p; const double* get() const { return p; } } dLx{F.Lx}, dUx{F.Ux};
    if (m > 0) {
        // column pointers were checked by the producer (monotone, totals); indices are checked here
        DevBuf<int> bad(1);
        IPXK_HIP(hipMemsetAsync(bad.get(), 0, sizeof(int), s));
        hipLaunchKernelGGL(validate_factors_kernel, dim3(grid_for(m)), dim3(kBlock), 0, s, m, dLp.get(), dLi.get(),
                           dUp.get(), dUi.get(), bad.get());
        int flag = 0;
        bad.download(&flag, 1, s);
        if (flag) throw Error(IPXK_E_ARGUMENT, "L or U violates the factor contract (index out of range, L not "
                                               "strictly lower, U not upper with its diagonal last)");
    }
    const double t1 = now();
    if (!c->prepare_host) c->prepare_host = new PrepareHost;
    Scratch& W = c->prepare_host->W;
    const size_t maxnz = (size_t)std::max<int64_t>(std::max(nzL, nzU), 1);
    W.rp.ensure((size_t)m + 1); W.ri.ensure(maxnz); W.rx.ensure(maxnz);
    W.dgn.ensure(std::max(m, 1));
    const size_t nkeys = std::max<size_t>(maxnz, (size_t)m + 1);
    W.keys.ensure(nkeys); W.vals.ensure(nkeys); W.keys2.ensure(nkeys); W.vals2.ensure(nkeys); W.colof.ensure(maxnz);
    const int g = grid_for(m);
    const int rowbits = bits_for(std::max(m, 2));
    const bool ll = S->level_launches;

    // --- U' sweep: unknown k gathers the rows above the diagonal of column k, ascending
    if (m > 0)
        hipLaunchKernelGGL(ut_rows_kernel, dim3(g), dim3(kBlock), 0, s, m, dUp.get(), dUi.get(), dUx.get(),
                           W.rp.get(), W.ri.get(), W.rx.get(), W.dgn.get());
    else IPXK_HIP(hipMemsetAsync(W.rp.get(), 0, sizeof(int), s));
    finish_sweep(c, W, S->Ut, ll, m, nzUo, true, false, 1, [&](std::vector<int>& lv) {
        H.fetch(F, m, s);
        const ipxint *Up = H.Up, *Ui = H.Ui;
        for (int k = 0; k < m; k++) {                       // unknown k gathers rows i < k of column k
            int l = 0;
            for (ipxint q = Up[k]; q < Up[k + 1] - 1; q++) l = std::max(l, lv[Ui[q]] + 1);
            lv[k] = l;
        }
    });

    // --- L' sweep: unknown k gathers column k of L (rows > k), descending, unit diagonal
    hipLaunchKernelGGL(lt_rows_kernel, dim3(grid_for(std::max<int64_t>(nzL, m + 1))), dim3(kBlock), 0, s, m, nzL,
                       dLp.get(), dLi.get(), dLx.get(), W.rp.get(), W.ri.get(), W.rx.get());
    hipLaunchKernelGGL(fill_double_kernel, dim3(g), dim3(kBlock), 0, s, (int64_t)m, 1.0, W.dgn.get());
    S->Lt.newest_first = true;      // a row's first entries (rows k+1, k+2, ... of column k) are the unknowns solved last
    finish_sweep(c, W, S->Lt, ll, m, nzL, false, false, 0, [&](std::vector<int>& lv) {
        H.fetch(F, m, s);
        const ipxint *Lp = H.Lp, *Li = H.Li;
        for (int k = m - 1; k >= 0; k--) {                  // unknown k gathers rows i > k of column k
            int l = 0;
            for (ipxint q = Lp[k]; q < Lp[k + 1]; q++) l = std::max(l, lv[Li[q]] + 1);
            lv[k] = l;
        }
    });

    // --- L sweep: row i of L, ascending column order (sparse_matrix.cc:283-297)
    if (m > 0 && nzL > 0) {
        hipLaunchKernelGGL(lf_keys_kernel, dim3(g), dim3(kBlock), 0, s, m, dLp.get(), dLi.get(), W.keys.get(),
                           W.vals.get(), W.colof.get());
        sort_pairs(W, nzL, rowbits, s);
    }
    hipLaunchKernelGGL(lower_bound_kernel, dim3(grid_for(m + 1)), dim3(kBlock), 0, s, m, nzL, W.keys2.get(),
                       W.rp.get());
    if (nzL > 0)
        hipLaunchKernelGGL(rows_from_perm_kernel, dim3(grid_for(nzL)), dim3(kBlock), 0, s, nzL, W.vals2.get(),
                           W.colof.get(), dLx.get(), W.ri.get(), W.rx.get());
    finish_sweep(c, W, S->Lf, ll, m, nzL, true, true, 0, [&](std::vector<int>& lv) {
        H.fetch(F, m, s);
        const ipxint *Lp = H.Lp, *Li = H.Li;
        for (int j = 0; j < m; j++)                         // column j is final when reached: push to rows i > j
            for (ipxint q = Lp[j]; q < Lp[j + 1]; q++) lv[Li[q]] = std::max(lv[Li[q]], lv[j] + 1);
    });

    // --- U sweep: row i of U without the diagonal, DESCENDING column order (sparse_matrix.cc:267-281)
    if (m > 0) {
        hipLaunchKernelGGL(uf_keys_kernel, dim3(g), dim3(kBlock), 0, s, m, dUp.get(), dUi.get(), W.keys.get(),
                           W.vals.get(), W.colof.get());
        if (nzUo > 0) sort_pairs(W, nzUo, rowbits, s);
        hipLaunchKernelGGL(u_diag_kernel, dim3(g), dim3(kBlock), 0, s, m, dUp.get(), dUx.get(), W.dgn.get());
    }
    hipLaunchKernelGGL(lower_bound_kernel, dim3(grid_for(m + 1)), dim3(kBlock), 0, s, m, nzUo, W.keys2.get(),
                       W.rp.get());
    if (nzUo > 0)
        hipLaunchKernelGGL(rows_from_perm_kernel, dim3(grid_for(nzUo)), dim3(kBlock), 0, s, nzUo, W.vals2.get(),
                           W.colof.get(), dUx.get(), W.ri.get(), W.rx.get());
    finish_sweep(c, W, S->Uf, ll, m, nzUo, false, true, 2, [&](std::vector<int>& lv) {
        H.fetch(F, m, s);
        const ipxint *Up = H.Up, *Ui = H.Ui;
        for (int j = m - 1; j >= 0; j--)                    // descending: push to rows i < j
            for (ipxint q = Up[j]; q < Up[j + 1] - 1; q++) lv[Ui[q]] = std::max(lv[Ui[q]], lv[j] + 1);
    });
    IPXK_HIP(hipStreamSynchronize(s));
    if (verbose)
        fprintf(stderr, "ipxk: device analysis: validation of L, U %.1f ms (%.0f MB), four sweeps %.1f ms\n", (t1 - t0) * 1e3,
                ((double)(nzL + nzU) * 16 + (double)m * 24) / 1e6, (now() - t1) * 1e3);
    IPXK_HIP(hipGetLastError());
}

}  // namespace ipxk
