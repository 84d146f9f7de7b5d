// Row-gather SpMV for gfx950: out[r] = finish(init(r) (+|-) sum_p prod(x[idx[p]], val[p])).
//
// One kernel serves every sparse product on the path (A'y, A t, diag(A W A'),
// right-hand-side assembly, solution recovery, the dense-column SMW terms and
// N N') through small epilogue functors.  HBM-bound: per nonzero it moves a
// 4-byte index and an 8-byte value exactly once with coalesced loads (lane i
// reads element base+i), the x gather is served by L2 / Infinity Cache.
//
// Summation order inside a row is the storage order (the owning thread adds the
// row's LDS-staged products sequentially), which is the reference's order:
//   * pass 1 of NormalMatrix::_Apply, d += rhs[Ai[p]]*Ax[p] (normal_matrix.cc:69-70)
//   * pass 2 equals the reference's one-pass scatter because row i of the
//     row-wise copy lists columns in ascending order (normal_matrix.cc:65-74).
// Products are rounded before they are added (they pass through LDS and the
// library is built with -ffp-contract=off), like the reference's x86 -O2 build.
#pragma once

#include "device_utils.hpp"
#include "internal.hpp"

namespace ipxk {

constexpr int kLdsDoubles = kChunkNnz + kChunkNnz / 32;

// +1 double of padding per 32: rows of equal length 8 (stride 64 B) would
// otherwise hit 4 of the 64 LDS banks 8 ways in the per-row read phase.
__device__ __forceinline__ int lds_slot(int q) { return q + (q >> 5); }

// ---- epilogues ---------------------------------------------------------------
struct ProdMul {
    static __device__ __forceinline__ double prod(double xg, double v) { return xg * v; }
};
// (v * w) * v, the evaluation order of diagonal_precond.cc:37
struct ProdSquareWeighted {
    static __device__ __forceinline__ double prod(double xg, double v) { return (v * xg) * v; }
};

// out[r] = acc * w[r]  (w == nullptr: out[r] = acc)
struct EpiScale : ProdMul {
    const double* w; double* out;
    static constexpr bool kNeg = false;
    __device__ __forceinline__ double init(int) const { return 0.0; }
    __device__ __forceinline__ void finish(int r, double acc, double&) const {
        out[r] = w ? acc * w[r] : acc;
    }
};
// out[r] = y[r]*wI[r] + acc, dot += y[r]*out[r]   (wI == nullptr: no slack term)
struct EpiNormalRows : ProdMul {
    const double* wI; const double* y; double* out;
    static constexpr bool kNeg = false;
    __device__ __forceinline__ double init(int r) const { return wI ? y[r] * wI[r] : 0.0; }
    __device__ __forceinline__ void finish(int r, double acc, double& dot) const {
        out[r] = acc;
        dot += y[r] * acc;
    }
};
// out[r] = wI[r] + sum (v*w)*v
struct EpiDiagonal : ProdSquareWeighted {
    const double* wI; double* out;
    static constexpr bool kNeg = false;
    __device__ __forceinline__ double init(int r) const { return wI ? wI[r] : 0.0; }
    __device__ __forceinline__ void finish(int r, double acc, double&) const { out[r] = acc; }
};
// out[r] = (-b[r] + acc) + wI[r]*aI[r]          (kkt_solver_diag.cc:90-92)
struct EpiKktRhs : ProdMul {
    const double* b; const double* wI; const double* aI; double* out;
    static constexpr bool kNeg = false;
    __device__ __forceinline__ double init(int r) const { return -b[r]; }
    __device__ __forceinline__ void finish(int r, double acc, double&) const {
        out[r] = acc + wI[r] * aI[r];
    }
};
// out[j] = w[j]*(a[j] - acc)                     (kkt_solver_diag.cc:111-112)
struct EpiRecoverX : ProdMul {
    const double* w; const double* a; double* out;
    static constexpr bool kNeg = false;
    __device__ __forceinline__ double init(int) const { return 0.0; }
    __device__ __forceinline__ void finish(int j, double acc, double&) const {
        out[j] = w[j] * (a[j] - acc);
    }
};
// out[i] = b[i] - sum                            (kkt_solver_diag.cc:108-116)
struct EpiResidualRows : ProdMul {
    const double* b; double* out;
    static constexpr bool kNeg = true;
    __device__ __forceinline__ double init(int r) const { return b[r]; }
    __device__ __forceinline__ void finish(int r, double acc, double&) const { out[r] = acc; }
};
// out[r] = (rhs[r] - acc)/diag[r], dot += out[r]*rhs[r]   (diagonal_precond.cc:145-149)
struct EpiSmwRows : ProdMul {
    const double* rhs; const double* diag; double* out;
    static constexpr bool kNeg = false;
    __device__ __forceinline__ double init(int) const { return 0.0; }
    __device__ __forceinline__ void finish(int r, double acc, double& dot) const {
        const double d = rhs[r] - acc;
        const double l = d / diag[r];
        out[r] = l;
        dot += l * rhs[r];
    }
};
// out[j] = mask[j] * scale2[j] * (a[j] - acc)    (kkt_solver_basis.cc:101-120,178-188)
struct EpiBasisColumns : ProdMul {
    const double* scale2;   // colscale^2 on nonbasic columns, 0 elsewhere
    const double* a; double* out;
    static constexpr bool kNeg = false;
    __device__ __forceinline__ double init(int) const { return 0.0; }
    __device__ __forceinline__ void finish(int j, double acc, double&) const {
        const double s = scale2[j];
        out[j] = s != 0.0 ? (a[j] - acc) * s : 0.0;
    }
};
// ---- the kernels ---------------------------------------------------------------
// Time-tiled row-gather SpMV (layout: internal.hpp, GatherMatrix).  Workgroup w of G
// co-resident workgroups walks the phases of its rows; all workgroups are in the same
// phase at (roughly) the same time, so the slice of x being gathered stays in L2.
//
// Three-stage software pipeline over chunks of kChunkNnz entries: while chunk k's
// products are staged in LDS and added to the row accumulators, the x gathers of chunk
// k+1 and the coalesced (idx,val,count) stream loads of chunk k+2 are in flight, so
// neither the HBM nor the L2 gather latency sits on the per-chunk critical path.
// LDS is double-buffered: one barrier per chunk.
constexpr int kPerThread = kChunkNnz / kBlock;

template <int RT>
struct ChunkLoad {          // stream loads of one chunk held in registers
    int c[kPerThread];
    double v[kPerThread];
    unsigned char cb[RT];   // entry counts of my RT rows in the step (valid when the chunk opens a step)
    int start, nn, first, step;
};

template <int RT>
__device__ __forceinline__ void issue_stream(const GatherView& M, int k, int kend, int w, int tid,
                                             ChunkLoad<RT>& L) {
    L.nn = 0; L.first = 0; L.step = 0; L.start = 0;
#pragma unroll
    for (int u = 0; u < RT; u++) L.cb[u] = 0;
    if (k >= kend) {
        // past the end: harmless indices so that the (unused) gathers stay in bounds
#pragma unroll
        for (int e = 0; e < kPerThread; e++) { L.c[e] = 0; L.v[e] = 0.0; }
        return;
    }
    L.start = M.chunk_start[k];
    const int info = M.chunk_info[k];
    L.nn = info & 0xffffff;
    L.first = info >> 30;
    L.step = M.chunk_step[k];
#pragma unroll
    for (int e = 0; e < kPerThread; e++) {
        const int qq = e * kBlock + tid;
        const int pp = L.start + (qq < L.nn ? qq : 0);
        L.c[e] = __builtin_nontemporal_load(M.idx + pp);
        L.v[e] = __builtin_nontemporal_load(M.val + pp);
    }
    if (L.first) {
        // thread t owns the rows t, t + kBlock, t + 2 kBlock, ... of the workgroup (strided ownership:
        // consecutive rows -> consecutive threads, so every chunk of a step keeps many threads busy
        // whether a row has its entries spread over all phases or concentrated in one)
#pragma unroll
        for (int u = 0; u < RT; u++) L.cb[u] = M.counts[(size_t)L.step * (kBlock * RT) + u * kBlock + tid];
    }
}

template <class Epi, int RT, bool MASKED = false>
__global__ __launch_bounds__(kBlock, 4) void spmv_phased_kernel(GatherView M, const double* __restrict__ x,
                                                             Epi epi, double* dot_partials,
                                                             const int* done) {
    if (done && *done) return;
    __shared__ double lds[2][kLdsDoubles];
    __shared__ double red[kBlock / 64 + 1];
    __shared__ int wave_total[2][RT][kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.x;
    double dotpart = 0.0;
    if (M.stamps && tid == 0) {
        // tuning aid: where this workgroup runs (XCC_ID = hwreg 20, HW_ID = hwreg 4)
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        M.stamps[(size_t)M.Q * M.P * M.G + w] = ((unsigned long long)xcc << 32) | hw;
    }

    for (int q = 0; q < M.Q; q++) {
        const int wg_row0 = (q * M.G + w) * M.RWrows;
        const int row_end = min(M.nrows, wg_row0 + M.RWrows);   // rows of this workgroup
        double acc[RT];
#pragma unroll
        for (int u = 0; u < RT; u++) {
            const int r = wg_row0 + u * kBlock + tid;
            acc[u] = r < row_end ? epi.init(r) : 0.0;
        }

        const int kbeg = M.wg_chunk_ptr[q * M.G + w], kend = M.wg_chunk_ptr[q * M.G + w + 1];
        ChunkLoad<RT> cur, nxt;
        double xg[kPerThread];
        issue_stream<RT>(M, kbeg, kend, w, tid, cur);
        issue_stream<RT>(M, kbeg + 1, kend, w, tid, nxt);
#pragma unroll
        for (int e = 0; e < kPerThread; e++) {                      // gathers of chunk 0
            if (MASKED) xg[e] = cur.v[e] != 0.0 ? x[cur.c[e]] : 0.0;    // masked value array (trisolve.hip): no gather for a zero entry
            else xg[e] = x[cur.c[e]];
        }

        int buf = 0, step_n0 = 0;
        int cnt[RT], off[RT];      // my rows' entry counts and first positions within the current step
#pragma unroll
        for (int u = 0; u < RT; u++) { cnt[u] = 0; off[u] = 0; }
        __syncthreads();   // LDS buffers free (previous round)
        for (int k = kbeg; k < kend; k++) {
            if (M.stamps && tid == 0 && cur.first) M.stamps[cur.step] = wall_clock64();
            // stage the products of chunk k (waits for its gathers only)
#pragma unroll
            for (int e = 0; e < kPerThread; e++) {
                const int qq = e * kBlock + tid;
                if (qq < cur.nn) lds[buf][lds_slot(qq)] = Epi::prod(xg[e], cur.v[e]);
            }
            // gathers of chunk k+1, stream loads of chunk k+2
            const int cstart = cur.start, cnn = cur.nn, cfirst = cur.first;
            int incl[RT];
            if (cfirst) {
                // positions of my rows' segments: the step stores rows in order, row u*kBlock + t
                // belongs to thread t -> one scan over the threads per u
                step_n0 = cstart;
#pragma unroll
                for (int u = 0; u < RT; u++) {
                    cnt[u] = cur.cb[u];
                    int v = cnt[u];
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int t = __shfl_up(v, o, 64);
                        if (lane >= o) v += t;
                    }
                    incl[u] = v;
                    if (lane == 63) wave_total[buf][u][wave] = v;
                }
            }
#pragma unroll
            for (int e = 0; e < kPerThread; e++) {
                if (MASKED) xg[e] = nxt.v[e] != 0.0 ? x[nxt.c[e]] : 0.0;
                else xg[e] = x[nxt.c[e]];
            }
            cur = nxt;
            issue_stream<RT>(M, k + 2, kend, w, tid, nxt);
            __syncthreads();
            if (cfirst) {
                int base = 0;
#pragma unroll
                for (int u = 0; u < RT; u++) {
                    int before = 0, total = 0;
#pragma unroll
                    for (int ww = 0; ww < kBlock / 64; ww++) {
                        const int t = wave_total[buf][u][ww];
                        if (ww < wave) before += t;
                        total += t;
                    }
                    off[u] = base + before + incl[u] - cnt[u];
                    base += total;
                }
            }
            // add my rows' products that lie in this chunk, in storage order
            const int shift = cstart - step_n0;
#pragma unroll
            for (int u = 0; u < RT; u++) {
                const int pos0 = off[u] - shift;
                if (pos0 >= cnn || pos0 + cnt[u] <= 0) continue;
                // the part of the row's segment inside this chunk, four LDS reads in flight at a time,
                // added strictly in storage order
                const int lo = max(pos0, 0), hi = min(pos0 + cnt[u], cnn);
                for (int pos = lo; pos < hi; pos += 4) {
                    double t[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) t[j] = lds[buf][lds_slot(min(pos + j, hi - 1))];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (pos + j < hi) acc[u] = Epi::kNeg ? acc[u] - t[j] : acc[u] + t[j];
                }
            }
            buf ^= 1;
        }
#pragma unroll
        for (int u = 0; u < RT; u++) {
            const int r = wg_row0 + u * kBlock + tid;
            if (r < row_end && !(M.row_long && M.row_long[r])) epi.finish(r, acc[u], dotpart);
        }
    }
    if (dot_partials) {
        const double d = block_reduce<SumOp>(dotpart, red);
        if (tid == 0) dot_partials[blockIdx.x] = d;
    }
}

// One segment of a long row per workgroup: strided per-thread sums, fixed block tree.
template <class Epi>
__global__ __launch_bounds__(kBlock) void spmv_long_kernel(GatherView M, const double* __restrict__ x,
                                                           const int* done) {
    if (done && *done) return;
    __shared__ double red[kBlock / 64 + 1];
    const int sgm = blockIdx.x;
    double acc = 0.0;
    for (int p = M.seg_p0[sgm] + threadIdx.x; p < M.seg_p1[sgm]; p += kBlock) {
        const double v = __builtin_nontemporal_load(M.lval + p);
        acc += Epi::prod(v != 0.0 ? x[M.lidx[p]] : 0.0, v);
    }
    acc = block_reduce<SumOp>(acc, red);
    if (threadIdx.x == 0) M.long_partials[sgm] = acc;
}

// Combines the segment sums of long rows in segment order and runs the epilogue;
// its dot contribution goes to dot_partials[dot_slot].
template <class Epi>
__global__ __launch_bounds__(kBlock) void spmv_long_fixup_kernel(GatherView M, Epi epi,
                                                                 double* dot_partials, int dot_slot,
                                                                 const int* done) {
    if (done && *done) return;
    __shared__ double red[kBlock / 64 + 1];
    double dotpart = 0.0;
    for (int l = threadIdx.x; l < M.nlong; l += kBlock) {
        const int r = M.long_row[l];
        double acc = epi.init(r);
        for (int s = M.long_slot[l]; s < M.long_slot[l + 1]; s++) {
            const double t = M.long_partials[s];
            acc = Epi::kNeg ? acc - t : acc + t;
        }
        epi.finish(r, acc, dotpart);
    }
    if (dot_partials) {
        const double d = block_reduce<SumOp>(dotpart, red);
        if (threadIdx.x == 0) dot_partials[dot_slot] = d;
    }
}

// ---- XCD-sliced tile layout (internal.hpp, SlicedMatrix) -------------------------
// One tile per workgroup: the tile's entries are streamed with coalesced loads (8 in flight per
// thread, gathers issued together), their products are staged in LDS, then every thread adds up
// the products of its 4 consecutive rows in storage order and writes the 4 partial sums (32 B).
// FUSED (one slice = the whole index space): no partial vectors and no combine launch; the thread
// starts each row from epi.init, adds the row's products in storage order (the reference's order,
// bit for bit) and applies epi.finish itself; the tile's dot partial goes to dot_partials[tile].
template <class Epi, int RPT, bool FUSED, bool MASKED = false>
__global__ __launch_bounds__(kBlock) void spmv_sliced_tile_kernel(SlicedView M, const double* __restrict__ x,
                                                                  Epi epi, double* dot_partials, const int* done) {
    if (done && *done) return;
    extern __shared__ double sl_prod[];
    __shared__ int wave_sum[kBlock / 64];
    __shared__ double red[kBlock / 64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // rows per thread: a thread's row counts are one 8-, 16- or 32-bit load
    static_assert(RPT == 1 || RPT == 2 || RPT == 4, "rows per thread");
    constexpr int R = kBlock * RPT;
    constexpr int U = 8;
    const int ntiles = M.nrb * M.nslices;
    double dotpart = 0.0;
    // sliced: one tile per workgroup (tile = blockIdx.x fixes the XCD); fused: a workgroup walks
    // tiles blockIdx.x, blockIdx.x + gridDim.x, ... so that the launch has at most kMaxPartials dot partials
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int s = tile % M.nslices, rb = tile / M.nslices;
        const unsigned e0 = M.tile_ptr[tile];
        const int ne = (int)(M.tile_ptr[tile + 1] - e0);
        const unsigned char* cbase = M.cnt + (size_t)tile * R;
        const unsigned c4 = RPT == 4 ? reinterpret_cast<const unsigned*>(cbase)[tid]
                          : RPT == 2 ? (unsigned)reinterpret_cast<const unsigned short*>(cbase)[tid]
                                     : (unsigned)cbase[tid];
        for (int base = 0; base < ne; base += kBlock * U) {
            int ci[U];
            double v[U], xg[U];
            // unpredicated loads (lanes past the end re-read the tile's last entry), so that all 2*U
            // stream loads and then all U gathers of a thread are in flight together
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = min(base + u * kBlock + tid, ne - 1);
                ci[u] = __builtin_nontemporal_load(M.idx + e0 + i);
                v[u] = __builtin_nontemporal_load(M.val + e0 + i);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (MASKED) xg[u] = v[u] != 0.0 ? x[ci[u]] : 0.0;   // masked value array (trisolve.hip): a zero entry needs no gather
                else xg[u] = x[ci[u]];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = base + u * kBlock + tid;
                if (i < ne) sl_prod[lds_slot(i)] = Epi::prod(xg[u], v[u]);
            }
        }
        // exclusive scan of the per-thread entry counts -> first staged product of my rows
        const int mine = (int)((c4 & 255u) + ((c4 >> 8) & 255u) + ((c4 >> 16) & 255u) + (c4 >> 24));   // unused bytes are 0
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int nb = __shfl_up(incl, d, 64);
            if (lane >= d) incl += nb;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        int p = incl - mine;
        for (int w = 0; w < wave; w++) p += wave_sum[w];
        if (FUSED) {
#pragma unroll
            for (int q = 0; q < RPT; q++) {
                const int cq = (int)((c4 >> (8 * q)) & 255u);
                const int r = rb * R + tid * RPT + q;
                if (r < M.nrows && !(M.row_long && M.row_long[r])) {
                    double acc = epi.init(r);
                    for (int kk = 0; kk < cq; kk++) {
                        const double t = sl_prod[lds_slot(p + kk)];
                        acc = Epi::kNeg ? acc - t : acc + t;
                    }
                    epi.finish(r, acc, dotpart);
                }
                p += cq;
            }
            __syncthreads();          // the staging buffer is reused by the next tile
            continue;
        }
        double out[RPT];
#pragma unroll
        for (int q = 0; q < RPT; q++) {
            const int cq = (int)((c4 >> (8 * q)) & 255u);
            double acc = 0.0;
            for (int kk = 0; kk < cq; kk++) acc += sl_prod[lds_slot(p + kk)];
            p += cq;
            out[q] = acc;
        }
        double* dst = M.partial + (size_t)s * M.nrows_pad + (size_t)rb * R + (size_t)tid * RPT;
        if (RPT == 1) {
            dst[0] = out[0];
        } else {
#pragma unroll
            for (int q = 0; q < RPT; q += 2) reinterpret_cast<double2*>(dst)[q / 2] = make_double2(out[q], out[q + 1 < RPT ? q + 1 : q]);
        }
    }
    if (FUSED && dot_partials) {
        const double d = block_reduce<SumOp>(dotpart, red);
        if (tid == 0) dot_partials[blockIdx.x] = d;
    }
}

// ---- sorted sub-tiles (internal.hpp, SortedMatrix) ---------------------------------------------
// One workgroup of 256 threads per (row block, slice).  For every sub-tile the workgroup streams the entries in
// batches of 2048 (packed word + value, coalesced, non-temporal, 8 per thread in flight), gathers x at ASCENDING
// addresses (neighbouring lanes share lines: fewer L1->L2 requests than entries) and writes each product to its LDS
// slot (the entry's place in row order) -- batch after batch without a barrier, slots are distinct; after ONE
// barrier every thread adds the products of its RPT consecutive rows from consecutive slots to its running sums,
// which pass from one sub-tile to the next: a row's partial sum is formed in storage order exactly as in the sliced
// layout.  The partial sums leave through LDS so that the store is coalesced.
template <int RPT, class Prod, bool MASKED = false>
__global__ __launch_bounds__(kSortedThreads, kSortedThreads / 128) void spmv_sorted_tile_kernel(SortedView M, const double* __restrict__ x, const int* done) {
    if (done && *done) return;
    extern __shared__ double so_prod[];
    __shared__ int wave_sum[kSortedThreads / 64];
    static_assert(RPT == 4 || RPT == 8 || RPT == 16 || RPT == 32, "rows per thread (sliced form)");
    constexpr int U = 8;
    constexpr int kBatch = kSortedThreads * U;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, s = tile % M.nslices, rb = tile / M.nslices;
    const double* __restrict__ xs = x + (size_t)s * M.slice_elems;
    constexpr int RB = kSortedThreads * RPT;
    double acc[RPT];
#pragma unroll
    for (int q = 0; q < RPT; q++) acc[q] = 0.0;
    // Software pipeline over the batches of ALL sub-tiles of the tile: while the gathers of a batch are in flight,
    // the stream loads of the next batch (of this or the next sub-tile) are issued, so that the load -> gather ->
    // stage chain of a workgroup overlaps with itself (few, large workgroups: 8 waves per CU).
    const int sub0 = tile * M.nsub;
    unsigned pk[U];
    double v[U];
    auto stream = [&](int sub, int base) {         // (pk, v) of the batch starting at `base` of sub-tile `sub`
        const unsigned e0 = M.sub_ptr[sub];
        const int ne = (int)(M.sub_ptr[sub + 1] - e0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = min(base + u * kSortedThreads + tid, max(ne - 1, 0));
            pk[u] = __builtin_nontemporal_load(M.pack + e0 + i);
            v[u] = __builtin_nontemporal_load(M.val + e0 + i);
        }
    };
    stream(sub0, 0);
    for (int h = 0; h < M.nsub; h++) {
        const int sub = sub0 + h;
        const int ne = (int)(M.sub_ptr[sub + 1] - M.sub_ptr[sub]);
        // my RPT row counts: RPT bytes (16-byte loads)
        unsigned cw[RPT / 4];
        {
            const unsigned* cbase = reinterpret_cast<const unsigned*>(M.cnt + (size_t)sub * RB) + (size_t)tid * (RPT / 4);
#pragma unroll
            for (int q = 0; q < RPT / 4; q++) cw[q] = cbase[q];
        }
        for (int base = 0; base < ne || base == 0; base += kBatch) {
            double xg[U];
            unsigned slot[U];
            double vv[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned off = pk[u] & ((1u << kSortedOffBits) - 1u);
                slot[u] = pk[u] >> kSortedOffBits;
                vv[u] = v[u];
                if (MASKED) xg[u] = v[u] != 0.0 ? xs[off] : 0.0;
                else xg[u] = xs[off];
            }
            // next batch: of this sub-tile, else the first one of the next sub-tile
            if (base + kBatch < ne) stream(sub, base + kBatch);
            else if (h + 1 < M.nsub) stream(sub + 1, 0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = base + u * kSortedThreads + tid;
                if (i < ne) so_prod[lds_slot((int)slot[u])] = Prod::prod(xg[u], vv[u]);
            }
        }
        // first slot of my rows: exclusive scan of the per-thread entry counts over the workgroup
        int mine = 0;
#pragma unroll
        for (int q = 0; q < RPT / 4; q++) mine += (int)((cw[q] & 255u) + ((cw[q] >> 8) & 255u) + ((cw[q] >> 16) & 255u) + (cw[q] >> 24));
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int nb = __shfl_up(incl, d, 64);
            if (lane >= d) incl += nb;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        int p = incl - mine;
        for (int w = 0; w < wave; w++) p += wave_sum[w];
        // first product of every row with independent LDS reads, the rest of a row (rare) one by one
        double first[RPT];
        int p0[RPT];
#pragma unroll
        for (int q = 0; q < RPT; q++) {
            const int cq = (int)((cw[q / 4] >> (8 * (q & 3))) & 255u);
            p0[q] = p;
            first[q] = cq > 0 ? so_prod[lds_slot(p)] : 0.0;
            p += cq;
        }
#pragma unroll
        for (int q = 0; q < RPT; q++) {
            const int cq = (int)((cw[q / 4] >> (8 * (q & 3))) & 255u);
            if (cq > 0) acc[q] += first[q];
            for (int kk = 1; kk < cq; kk++) acc[q] += so_prod[lds_slot(p0[q] + kk)];
        }
        __syncthreads();                  // the staging buffer and wave_sum are reused by the next sub-tile
    }
    // coalesced store of the RB partial sums: through LDS (thread t holds rows t*RPT .. t*RPT + RPT - 1)
#pragma unroll
    for (int q = 0; q < RPT; q++) so_prod[lds_slot(tid * RPT + q)] = acc[q];
    __syncthreads();
    double* dst = M.partial + (size_t)s * M.nrows_pad + (size_t)rb * RB;
#pragma unroll
    for (int q = 0; q < RPT; q += 2) {
        const int r = (q / 2) * (2 * kSortedThreads) + 2 * tid;
        reinterpret_cast<double2*>(dst + r)[0] = make_double2(so_prod[lds_slot(r)], so_prod[lds_slot(r + 1)]);
    }
}

// The FUSED form: one slice, one sub-tile per row block, gathered indices relative to the tile's smallest one; the
// thread starts each row from epi.init, adds the row's products in storage order (the reference's order, bit for
// bit as in the phased and fused layouts) and applies epi.finish itself.  A workgroup walks tiles blockIdx.x,
// blockIdx.x + gridDim.x, ...; its dot partial goes to dot_partials[blockIdx.x].
template <class Epi, int RPT>
__global__ __launch_bounds__(kSortedThreads, kSortedThreads / 128) void spmv_sorted_fused_kernel(SortedView M, const double* __restrict__ x, Epi epi,
                                                                                           double* dot_partials, const int* done) {
    if (done && *done) return;
    extern __shared__ double so_prod[];
    __shared__ int wave_sum[kSortedThreads / 64];
    __shared__ double red[kSortedThreads / 64 + 1];
    static_assert(RPT == 1 || RPT == 2 || RPT == 4 || RPT == 8 || RPT == 16 || RPT == 32, "rows per thread");
    constexpr int U = 8;
    constexpr int kBatch = kSortedThreads * U;
    constexpr int RB = kSortedThreads * RPT;
    constexpr int CW = RPT >= 4 ? RPT / 4 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double dotpart = 0.0;
    unsigned pk[U];
    double v[U];
    auto stream = [&](int tile, int base) {
        const unsigned e0 = M.sub_ptr[tile];
        const int ne = (int)(M.sub_ptr[tile + 1] - e0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = min(base + u * kSortedThreads + tid, max(ne - 1, 0));
            pk[u] = __builtin_nontemporal_load(M.pack + e0 + i);
            v[u] = __builtin_nontemporal_load(M.val + e0 + i);
        }
    };
    if ((int)blockIdx.x < M.nrb) stream(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < M.nrb; tile += gridDim.x) {
        const int ne = (int)(M.sub_ptr[tile + 1] - M.sub_ptr[tile]);
        const double* __restrict__ xs = x + M.xmin[tile];
        unsigned cw[CW];
        {
            const unsigned char* cbase = M.cnt + (size_t)tile * RB + (size_t)tid * RPT;
            if (RPT >= 4) {
#pragma unroll
                for (int q = 0; q < CW; q++) cw[q] = reinterpret_cast<const unsigned*>(cbase)[q];
            } else if (RPT == 2) {
                cw[0] = (unsigned)reinterpret_cast<const unsigned short*>(cbase)[0];
            } else {
                cw[0] = (unsigned)cbase[0];
            }
        }
        for (int base = 0; base < ne || base == 0; base += kBatch) {
            double xg[U], vv[U];
            unsigned slot[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned off = pk[u] & ((1u << kSortedOffBits) - 1u);
                slot[u] = pk[u] >> kSortedOffBits;
                vv[u] = v[u];
                xg[u] = xs[off];
            }
            if (base + kBatch < ne) stream(tile, base + kBatch);
            else if (tile + (int)gridDim.x < M.nrb) stream(tile + gridDim.x, 0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = base + u * kSortedThreads + tid;
                if (i < ne) so_prod[lds_slot((int)slot[u])] = Epi::prod(xg[u], vv[u]);
            }
        }
        int mine = 0;
#pragma unroll
        for (int q = 0; q < CW; q++) mine += (int)((cw[q] & 255u) + ((cw[q] >> 8) & 255u) + ((cw[q] >> 16) & 255u) + (cw[q] >> 24));
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int nb = __shfl_up(incl, d, 64);
            if (lane >= d) incl += nb;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        int p = incl - mine;
        for (int w = 0; w < wave; w++) p += wave_sum[w];
#pragma unroll
        for (int q = 0; q < RPT; q++) {
            const int cq = (int)((cw[q / 4] >> (8 * (q & 3))) & 255u);
            const int r = tile * RB + tid * RPT + q;
            if (r < M.nrows && !(M.row_long && M.row_long[r])) {
                double acc = epi.init(r);
                for (int kk = 0; kk < cq; kk += 4) {           // four LDS reads in flight, added strictly in storage order
                    double t[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) t[j] = so_prod[lds_slot(p + min(kk + j, cq - 1))];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (kk + j < cq) acc = Epi::kNeg ? acc - t[j] : acc + t[j];
                }
                epi.finish(r, acc, dotpart);
            }
            p += cq;
        }
        __syncthreads();                  // the staging buffer and wave_sum are reused by the next tile
    }
    if (dot_partials) {
        double d = dotpart;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
        if (lane == 0) red[wave] = d;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < kSortedThreads / 64; w++) t += red[w];
            dot_partials[blockIdx.x] = t;
        }
    }
}

// ---- accumulated tiles (internal.hpp, AccMatrix) -----------------------------------------------
// One workgroup of 512 threads per (row block, slice); the RB row sums live in LDS.  Batch after batch (at most 2048
// entries, 4 per thread): packed word + value streamed non-temporally, x gathered at ascending addresses (neighbouring
// lanes share lines), one barrier, one ds_add_f64 per entry -- the layout guarantees that a batch holds at most one
// entry of a row, so the adds of a batch hit distinct addresses and a row is summed in batch order.  The stream loads
// of batch k+1 are in flight while the gathers of batch k return.  The row sums leave as one partial vector per slice.
template <class Prod>
__global__ __launch_bounds__(kAccThreads) void spmv_acc_tile_kernel(AccView M, const double* __restrict__ x, const int* done) {
    if (done && *done) return;
    extern __shared__ double ac_sum[];
    constexpr int U = kAccPerThread, T = kAccThreads;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, s = tile % M.nslices, rb = tile / M.nslices;
    const double* __restrict__ xs = x + (size_t)s * M.slice_elems;
    const unsigned b0 = M.tile_batch[tile];
    const int nb = (int)(M.tile_batch[tile + 1] - b0);
    // first entry of batch k of this tile (wave-uniform: scalar loads), fetched well ahead of its use
    auto first = [&](int k) -> unsigned { return M.bptr[b0 + min(k, nb)]; };
    constexpr int NE = 6;
    unsigned e[NE];                       // e[i] = first(k + i) for the current batch k
#pragma unroll
    for (int i = 0; i < NE; i++) e[i] = first(i);
    // Three-stage software pipeline over the batches: while the products of batch k are added to the row sums, the
    // gathers of batch k + 1 and the stream loads of batches k + 2 and k + 3 are in flight -- an iteration waits only for
    // loads issued a whole iteration earlier (the iteration time was one dependent round trip stream -> gather before:
    // 2 us per batch of 2048 entries).  Three stream buffers and two gather buffers rotate; the loop is unrolled six times.
    unsigned pk[3][U];
    double v[3][U], xg[2][U];
    auto stream = [&](unsigned (&pkb)[U], double (&vb)[U], unsigned e0, unsigned e1) {       // entries [e0, e1) of a batch (none: e0 == e1)
        const int ne = (int)(e1 - e0);
        if (ne <= 0) return;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = min(u * T + tid, ne - 1);
            pkb[u] = __builtin_nontemporal_load(M.pack + e0 + i);
            vb[u] = __builtin_nontemporal_load(M.val + e0 + i);
        }
    };
    auto gather = [&](double (&xgb)[U], const unsigned (&pkb)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) xgb[u] = xs[pkb[u] & ((1u << kSortedOffBits) - 1u)];
    };
#pragma unroll
    for (int d = 0; d < 3; d++) {
#pragma unroll
        for (int u = 0; u < U; u++) { pk[d][u] = 0u; v[d][u] = 0.0; }
        stream(pk[d], v[d], e[d], e[d + 1]);
    }
    for (int r = tid; r < M.RB; r += T) ac_sum[r] = 0.0;
    if (nb > 0) gather(xg[0], pk[0]);
    for (int k0 = 0; k0 < nb; k0 += 6) {
#pragma unroll
        for (int d = 0; d < 6; d++) {
            const int k = k0 + d;
            if (k < nb) {
                const int ne = (int)(e[1] - e[0]);
                if (k + 1 < nb) gather(xg[(d + 1) & 1], pk[(d + 1) % 3]);          // batch k + 1
                const unsigned enew = first(k + NE);
                __syncthreads();                  // the adds of the previous batch (first batch: the zeroing) are done
                // (a plain read-modify-write -- the rows of a batch are distinct -- measured 3 % slower than ds_add_f64)
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (u * T + tid < ne)
                        __hip_atomic_fetch_add(ac_sum + (pk[d % 3][u] >> kSortedOffBits), Prod::prod(xg[d & 1][u], v[d % 3][u]), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                stream(pk[d % 3], v[d % 3], e[3], e[4]);               // batch k + 3 into the buffer just emptied
#pragma unroll
                for (int i = 0; i + 1 < NE; i++) e[i] = e[i + 1];
                e[NE - 1] = enew;
            }
        }
    }
    __syncthreads();
    double* dst = M.partial + (size_t)s * M.nrows_pad + (size_t)rb * M.RB;
    for (int r = 2 * tid; r < M.RB; r += 2 * T) reinterpret_cast<double2*>(dst + r)[0] = make_double2(ac_sum[r], ac_sum[r + 1]);
}

// ---- plain rows (small matrices) --------------------------------------------------------------
// The matrix as it is (the device's plain copy, layout_device.hip): 8 lanes per row, lane k takes entries k, k + 8, ...
// (coalesced: a wavefront reads 64 consecutive entries), the first lane of the group adds the products in storage
// order (the reference's order, bit for bit as in the phased and fused layouts).  No LDS, no barrier, no tile
// tables: a pass is three dependent round trips (row pointers -> entries -> gathers).  For matrices of a few hundred
// thousand entries, where a pass of the tiled layouts is 10 us of latency chains for 10 MB (C2: 50k x 100k).
template <class Epi>
__global__ __launch_bounds__(kBlock) void spmv_rowgroup_kernel(int nrows, const int* __restrict__ ptr, const int* __restrict__ idx,
                                                               const double* __restrict__ val, const unsigned char* __restrict__ row_long,
                                                               const double* __restrict__ x, Epi epi, double* dot_partials, const int* done) {
    if (done && *done) return;
    __shared__ double red[kBlock / 64 + 1];
    const int lane8 = threadIdx.x & 7, lane = threadIdx.x & 63, g0 = lane & ~7;
    double dotpart = 0.0;
    const int ngroups = (gridDim.x * kBlock) >> 3;
    for (int r = (blockIdx.x * kBlock + threadIdx.x) >> 3; r < nrows; r += ngroups) {
        const bool live = !(row_long && row_long[r]);
        const int p0 = ptr[r], p1 = live ? ptr[r + 1] : p0;
        double acc = lane8 == 0 ? epi.init(r) : 0.0;
        for (int p = p0; p < p1; p += 8) {
            const int q = p + lane8;
            const double prod = q < p1 ? Epi::prod(x[idx[q]], val[q]) : 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const double t = __shfl(prod, g0 + k, 64);
                if (p + k < p1) acc = Epi::kNeg ? acc - t : acc + t;
            }
        }
        if (live && lane8 == 0) epi.finish(r, acc, dotpart);
    }
    if (dot_partials) {
        const double d = block_reduce<SumOp>(dotpart, red);
        if (threadIdx.x == 0) dot_partials[blockIdx.x] = d;
    }
}

// The FUSED form: one slice, a tile per row block, gathered indices relative to the tile's smallest one; the row sums
// start from epi.init, the products are added batch by batch (ascending address = storage order of sorted rows) and the
// kernel applies epi.finish itself; the tile's dot partial goes to dot_partials[tile].  The pipeline is the sliced
// form's: adds of batch k, gathers of k + 1, stream loads of k + 2 and k + 3 in flight.
template <class Epi>
__global__ __launch_bounds__(kAccThreads) void spmv_acc_fused_kernel(AccView M, const double* __restrict__ x, Epi epi, double* dot_partials,
                                                                     const int* done) {
    if (done && *done) return;
    extern __shared__ double ac_sum[];
    __shared__ double red[kAccThreads / 64];
    constexpr int U = kAccPerThread, T = kAccThreads;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const double* __restrict__ xs = x + M.xmin[tile];
    const unsigned b0 = M.tile_batch[tile];
    const int nb = (int)(M.tile_batch[tile + 1] - b0);
    auto first = [&](int k) -> unsigned { return M.bptr[b0 + min(k, nb)]; };
    constexpr int NE = 6;
    unsigned e[NE];
#pragma unroll
    for (int i = 0; i < NE; i++) e[i] = first(i);
    unsigned pk[3][U];
    double v[3][U], xg[2][U];
    auto stream = [&](unsigned (&pkb)[U], double (&vb)[U], unsigned e0, unsigned e1) {
        const int ne = (int)(e1 - e0);
        if (ne <= 0) return;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = min(u * T + tid, ne - 1);
            pkb[u] = __builtin_nontemporal_load(M.pack + e0 + i);
            vb[u] = __builtin_nontemporal_load(M.val + e0 + i);
        }
    };
    auto gather = [&](double (&xgb)[U], const unsigned (&pkb)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) xgb[u] = xs[pkb[u] & ((1u << kSortedOffBits) - 1u)];
    };
#pragma unroll
    for (int d = 0; d < 3; d++) {
#pragma unroll
        for (int u = 0; u < U; u++) { pk[d][u] = 0u; v[d][u] = 0.0; }
        stream(pk[d], v[d], e[d], e[d + 1]);
    }
    const int r0 = tile * M.RB;
    for (int r = tid; r < M.RB; r += T) ac_sum[r] = r0 + r < M.nrows ? epi.init(r0 + r) : 0.0;
    if (nb > 0) gather(xg[0], pk[0]);
    for (int k0 = 0; k0 < nb; k0 += 6) {
#pragma unroll
        for (int d = 0; d < 6; d++) {
            const int k = k0 + d;
            if (k < nb) {
                const int ne = (int)(e[1] - e[0]);
                if (k + 1 < nb) gather(xg[(d + 1) & 1], pk[(d + 1) % 3]);
                const unsigned enew = first(k + NE);
                __syncthreads();
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (u * T + tid < ne) {
                        const double p = Epi::prod(xg[d & 1][u], v[d % 3][u]);
                        __hip_atomic_fetch_add(ac_sum + (pk[d % 3][u] >> kSortedOffBits), Epi::kNeg ? -p : p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                stream(pk[d % 3], v[d % 3], e[3], e[4]);
#pragma unroll
                for (int i = 0; i + 1 < NE; i++) e[i] = e[i + 1];
                e[NE - 1] = enew;
            }
        }
    }
    __syncthreads();
    double dotpart = 0.0;
    for (int r = tid; r < M.RB; r += T)
        if (r0 + r < M.nrows) epi.finish(r0 + r, ac_sum[r], dotpart);
    if (dot_partials) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dotpart += __shfl_down(dotpart, o, 64);
        if ((tid & 63) == 0) red[tid >> 6] = dotpart;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < T / 64; w++) t += red[w];
            dot_partials[tile] = t;
        }
    }
}

// out[r] = finish(init(r) (+|-) partial[0][r] (+|-) partial[1][r] ...), slices in ascending order
template <class Epi>
__global__ __launch_bounds__(kBlock) void spmv_sliced_combine_kernel(SlicedView M, Epi epi, double* dot_partials,
                                                                     const int* done) {
    if (done && *done) return;
    __shared__ double red[kBlock / 64 + 1];
    double dotpart = 0.0;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < M.nrows; r += gridDim.x * kBlock) {
        if (M.row_long && M.row_long[r]) continue;       // finished by the long-row fix-up kernel
        double acc = epi.init(r);
        for (int s = 0; s < M.nslices; s++) {
            const double t = __builtin_nontemporal_load(M.partial + (size_t)s * M.nrows_pad + r);
            acc = Epi::kNeg ? acc - t : acc + t;
        }
        epi.finish(r, acc, dotpart);
    }
    if (dot_partials) {
        const double d = block_reduce<SumOp>(dotpart, red);
        if (threadIdx.x == 0) dot_partials[blockIdx.x] = d;
    }
}

// COMPACT: use the compacted copy of the tiles (GatherMatrix::compact_tiles) -- the plain kernels on fewer entries
template <class Epi, bool MASKED = false, bool COMPACT = false>
inline void launch_spmv_sliced(const GatherMatrix& M, const double* x, const Epi& epi, double* dot_partials,
                               const int* done, hipStream_t s) {
    static_assert(!(MASKED && COMPACT), "the compacted copy needs no mask");
    if (M.use_acc && !MASKED && !COMPACT) {
        // accumulated tiles + the sliced layout's combine on their partial vectors
        const AccView W = M.acc_view();
        const size_t lds = (size_t)W.RB * sizeof(double);
        static bool lds_attr_set = false;       // per instantiation: dynamic LDS beyond 64 KB has to be allowed once
        if (!lds_attr_set) {
            IPXK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spmv_acc_tile_kernel<Epi>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(kAccMaxRows * sizeof(double))));
            lds_attr_set = true;
        }
        hipLaunchKernelGGL((spmv_acc_tile_kernel<Epi>), dim3(W.nrb * W.nslices), dim3(kAccThreads), lds, s, W, x, done);
        SlicedView C = M.sliced_view(0);
        C.nrows_pad = W.nrows_pad; C.partial = W.partial;
        hipLaunchKernelGGL(spmv_sliced_combine_kernel<Epi>, dim3(M.combine_grid()), dim3(kBlock), 0, s, C, epi, dot_partials, done);
        if (M.nlong > 0) {
            const GatherView G = M.view(false);
            hipLaunchKernelGGL(spmv_long_kernel<Epi>, dim3(M.nseg), dim3(kBlock), 0, s, G, x, done);
            hipLaunchKernelGGL(spmv_long_fixup_kernel<Epi>, dim3(1), dim3(kBlock), 0, s, G, epi, dot_partials, M.combine_grid(), done);
        }
        return;
    }
    const SlicedView V = M.sliced_view(COMPACT ? 2 : MASKED ? 1 : 0);
    const size_t lds = (size_t)(M.sliced.max_tile + M.sliced.max_tile / 32 + 1) * sizeof(double);
    const dim3 grid(V.nrb * V.nslices), block(kBlock);
    if (V.nslices == 1) {       // fused: the tile kernel is the whole product
        const dim3 fgrid(M.fused_grid());
        if (V.R == kBlock * 4) hipLaunchKernelGGL((spmv_sliced_tile_kernel<Epi, 4, true, MASKED>), fgrid, block, lds, s, V, x, epi, dot_partials, done);
        else if (V.R == kBlock * 2) hipLaunchKernelGGL((spmv_sliced_tile_kernel<Epi, 2, true, MASKED>), fgrid, block, lds, s, V, x, epi, dot_partials, done);
        else hipLaunchKernelGGL((spmv_sliced_tile_kernel<Epi, 1, true, MASKED>), fgrid, block, lds, s, V, x, epi, dot_partials, done);
    } else {
        if (V.R == kBlock * 4) hipLaunchKernelGGL((spmv_sliced_tile_kernel<Epi, 4, false, MASKED>), grid, block, lds, s, V, x, epi, dot_partials, done);
        else if (V.R == kBlock * 2) hipLaunchKernelGGL((spmv_sliced_tile_kernel<Epi, 2, false, MASKED>), grid, block, lds, s, V, x, epi, dot_partials, done);
        else hipLaunchKernelGGL((spmv_sliced_tile_kernel<Epi, 1, false, MASKED>), grid, block, lds, s, V, x, epi, dot_partials, done);
        hipLaunchKernelGGL(spmv_sliced_combine_kernel<Epi>, dim3(M.combine_grid()), dim3(kBlock), 0, s, V, epi,
                           dot_partials, done);
    }
    if (M.nlong > 0) {          // rows of more than kMaxRowLen entries: segment sums + ordered fix-up
        const GatherView G = M.view(MASKED || COMPACT);
        hipLaunchKernelGGL(spmv_long_kernel<Epi>, dim3(M.nseg), block, 0, s, G, x, done);
        hipLaunchKernelGGL(spmv_long_fixup_kernel<Epi>, dim3(1), block, 0, s, G, epi, dot_partials,
                           V.nslices == 1 ? M.fused_grid() : M.combine_grid(), done);
    }
}

// Launches the SpMV (+ long-row kernels when the matrix has long rows).  Returns
// the number of dot partials written (0 when dot_partials == nullptr).
template <class Epi, bool MASKED = false>
inline int launch_spmv(const GatherMatrix& M, const double* x, const Epi& epi, double* dot_partials,
                       const int* done, hipStream_t s) {
    if (M.use_acc_fused && !MASKED) {
        const AccView W = M.acc_fused_view();
        static bool lds_attr_set = false;
        if (!lds_attr_set) {
            IPXK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spmv_acc_fused_kernel<Epi>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(kAccMaxRows * sizeof(double))));
            lds_attr_set = true;
        }
        hipLaunchKernelGGL((spmv_acc_fused_kernel<Epi>), dim3(W.nrb), dim3(kAccThreads), (size_t)W.RB * sizeof(double), s, W, x, epi, dot_partials, done);
        return dot_partials ? W.nrb : 0;
    }
    if (M.use_plain && !MASKED) {
        const int grid = M.plain_grid();
        hipLaunchKernelGGL(spmv_rowgroup_kernel<Epi>, dim3(grid), dim3(kBlock), 0, s, M.nrows, M.csr_ptr, M.csr_idx, M.csr_val,
                           M.nlong > 0 ? M.row_long.get() : nullptr, x, epi, dot_partials, done);
        if (M.nlong > 0) {
            const GatherView G = M.view(false);
            hipLaunchKernelGGL(spmv_long_kernel<Epi>, dim3(M.nseg), dim3(kBlock), 0, s, G, x, done);
            hipLaunchKernelGGL(spmv_long_fixup_kernel<Epi>, dim3(1), dim3(kBlock), 0, s, G, epi, dot_partials, grid, done);
        }
        return dot_partials ? grid + (M.nlong > 0 ? 1 : 0) : 0;
    }
    if (M.use_sorted_fused && !MASKED) {
        const SortedView W = M.sorted_view();
        const size_t lds = (size_t)(M.sorted.max_sub + M.sorted.max_sub / 32 + 1) * sizeof(double);
        const dim3 grid(M.sorted_fused_grid()), block(kSortedThreads);
        switch (W.RB / kSortedThreads) {
            case 32: hipLaunchKernelGGL((spmv_sorted_fused_kernel<Epi, 32>), grid, block, lds, s, W, x, epi, dot_partials, done); break;
            case 16: hipLaunchKernelGGL((spmv_sorted_fused_kernel<Epi, 16>), grid, block, lds, s, W, x, epi, dot_partials, done); break;
            case 8: hipLaunchKernelGGL((spmv_sorted_fused_kernel<Epi, 8>), grid, block, lds, s, W, x, epi, dot_partials, done); break;
            case 4: hipLaunchKernelGGL((spmv_sorted_fused_kernel<Epi, 4>), grid, block, lds, s, W, x, epi, dot_partials, done); break;
            case 2: hipLaunchKernelGGL((spmv_sorted_fused_kernel<Epi, 2>), grid, block, lds, s, W, x, epi, dot_partials, done); break;
            default: hipLaunchKernelGGL((spmv_sorted_fused_kernel<Epi, 1>), grid, block, lds, s, W, x, epi, dot_partials, done); break;
        }
        if (M.nlong > 0) {
            const GatherView G = M.view(false);
            hipLaunchKernelGGL(spmv_long_kernel<Epi>, dim3(M.nseg), dim3(kBlock), 0, s, G, x, done);
            hipLaunchKernelGGL(spmv_long_fixup_kernel<Epi>, dim3(1), dim3(kBlock), 0, s, G, epi, dot_partials, M.sorted_fused_grid(), done);
        }
        return dot_partials ? M.sorted_fused_grid() + (M.nlong > 0 ? 1 : 0) : 0;
    }
    if (M.use_sliced) {
        if (MASKED && M.compact.valid) launch_spmv_sliced<Epi, false, true>(M, x, epi, dot_partials, done, s);
        else launch_spmv_sliced<Epi, MASKED>(M, x, epi, dot_partials, done, s);
        return dot_partials ? M.num_partials() : 0;
    }
    const GatherView V = M.view(MASKED);
    const dim3 grid(M.G), block(kBlock);
    switch (M.RT) {
        case 1: hipLaunchKernelGGL((spmv_phased_kernel<Epi, 1, MASKED>), grid, block, 0, s, V, x, epi, dot_partials, done); break;
        case 2: hipLaunchKernelGGL((spmv_phased_kernel<Epi, 2, MASKED>), grid, block, 0, s, V, x, epi, dot_partials, done); break;
        case 4: hipLaunchKernelGGL((spmv_phased_kernel<Epi, 4, MASKED>), grid, block, 0, s, V, x, epi, dot_partials, done); break;
        default: hipLaunchKernelGGL((spmv_phased_kernel<Epi, 8, MASKED>), grid, block, 0, s, V, x, epi, dot_partials, done); break;
    }
    if (M.nlong > 0) {
        hipLaunchKernelGGL(spmv_long_kernel<Epi>, dim3(M.nseg), block, 0, s, V, x, done);
        hipLaunchKernelGGL(spmv_long_fixup_kernel<Epi>, dim3(1), block, 0, s, V, epi, dot_partials, M.G, done);
    }
    return dot_partials ? M.num_partials() : 0;
}

}  // namespace ipxk
