// Basis LU factorization on the device behind the reference's
// LuFactorization contract, src/lu_factorization.h:21-58:
//     B[rowperm,colperm] = (L+I)*U,  L strictly lower without its diagonal, U upper with the diagonal last in
//     each column, indices sorted; dependent columns replaced by unit columns and listed.
// It takes the place of the reference's kernel src/basiclu_kernel.cc:31-82 (BASICLU, third party); the call
// sites are ForrestTomlin::_Factorize (src/forrest_tomlin.cc:28-30) and, through LuUpdate::Factorize,
// Basis::Factorize (src/basis.cc:116-156).
//
// LP bases are nearly triangular, and that is what this code is built for:
//   1. SINGLETON ROUNDS.  A column with one entry in the active rows is a pivot without arithmetic (its other
//      entries are entries of U); so is a row with one entry in the active columns (the rest of the pivot column,
//      divided by the pivot, is a column of L; nothing fills in).  All current column singletons, then all
//      current row singletons, are taken per round -- they are independent of each other -- until neither
//      exists.  Integer work over the CSC and a row-wise copy of B: one thread per column (row), conflicts (two
//      singleton columns in one row, two singleton rows in one column) settled by atomic min / max, the pivot
//      order inside a round fixed by a prefix sum over the indices, so the result does not depend on timing.
//   2. THE BUMP that remains (for LP bases: tens to a few thousand rows) is factorized as a DENSE matrix with
//      partial pivoting, right-looking, in panels of 32 columns: a one-workgroup panel kernel (pivot search,
//      scaling, update inside the panel), a kernel that finishes the panel's rows of U, and a tiled update of
//      the trailing matrix.  Every entry receives its updates one pivot at a time in pivot order, products
//      rounded before they are subtracted: the arithmetic of the plain column-by-column elimination, bit for
//      bit.  Rows are never swapped (a row carries the step at which it was pivoted).
//   3. ASSEMBLY.  Every entry of B outside the bump and every nonzero of the factorized bump is keyed by
//      (pivot stage of its column, pivot stage of its row); one radix sort per factor orders the columns and,
//      inside them, the rows.
// The rules (tolerances, tie breaks, order of the dependent columns) are restated on the CPU by the test
// infrastructure, which the tests compare against entry by entry; the reference's own
// LuFactorization::Factorize (stability estimate) and ForrestTomlin judge the factors there as well.
//   2b. TEARING.  A bump of more than IPXK_LU_BUMP_MAX rows (default 8192) is not factorized densely as it stands.
//      LP bumps are sparse and nearly triangular themselves -- a few columns (the ones recent basis exchanges
//      brought in) block the singleton rounds, and singleton peeling is all-or-nothing along dependency chains.
//      Whenever the rounds stall, the T active columns with the most active entries (ties: smaller index) are set
//      aside as SPIKES (T = 1, doubled up to 1024 while a tear frees fewer than 64 pivots, back to 1 otherwise)
//      and the rounds go on without them until no active column is left: the bump-and-spike ordering of
//      Hellerman and Rarick in rounds.  Round pivots still cost no arithmetic outside the spikes.  The spikes
//      receive the updates of the row singleton pivots in pivot order: a forward substitution with the L columns
//      found so far, run for 64 spikes at a time as a dense dim x 64 block (one wavefront per row, one lane per
//      spike, rows of one half-round in one launch, every row's products subtracted in pivot order).  Their
//      entries in pivoted rows become entries of U, their entries in the never-pivoted rows form the dense block
//      of step 2.
//   2c. ELIMINATION ROUNDS (round 4).  If the spikes themselves would exceed the limit, the basis is no longer refused: the
//      factorization starts again and, when the singleton rounds stall, eliminates the bump SPARSELY.  The active submatrix
//      becomes a compact problem of its own (rows / columns renumbered in order, CSC + a row-wise index).  A round
//        * picks one candidate per column: among the entries that pass the absolute and the relative pivot threshold, the one in
//          the shortest row; its Markowitz cost is (row count - 1)(column count - 1);
//        * lets the candidates of cost <= max(4, 2 x the cheapest, the quartile of all costs rounded up to 2^b - 1) compete: a row
//          keeps its best (cost, then column index), and a contender wins unless a better one has an entry in its pivot row or
//          its pivot row in this column -- the winners' pivots form a DIAGONAL block, so eliminating them together is exact
//          (Schur complement = A22 - sum over winners of column x row / pivot), and a singleton is simply a winner of cost 0;
//        * writes the updates -(a_i'j / pivot) a_ij' behind the surviving entries (which keep their order: the renumbering is
//          monotone), sorts the updates, merges the two sorted lists (stable: entry first, then the updates in ascending order of
//          the winner's column), sums equal positions in that order with the product rounded before it is subtracted, drops
//          sums that are exactly zero, and rebuilds pointers, row index and counts.  Entries whose row or column was pivoted
//          leave the matrix for the list E (indices in B, value at that time): E replaces B in the assembly of step 3.
//      The rounds end at IPXK_LU_SPARSE_MIN columns (512), or -- once the rest fits the dense code -- after two rounds in a row
//      that each eliminated fewer than 1 / 256 of the columns (what is left has no large independent pivot sets any more), or
//      when no column has an acceptable pivot; they are GIVEN UP (IPXK_E_UNSUPPORTED) when the current matrix grows beyond
//      IPXK_LU_SPARSE_FILL_MAX (8) x nnz(B) + 2^20 entries -- bounded work.  The rest is factorized densely as in step 2, its columns in ascending order
//      of their number of entries (fewer nonzeros in the dense factors).  Only if THAT rest exceeds the limit is the basis
//      refused (IPXK_E_UNSUPPORTED).  IPXK_LU_SPARSE=1: elimination rounds instead of tearing from the start; =0: never.
//      Every rule is restated sequentially on the CPU by the test infrastructure; the factors are bit-identical to that restatement.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_merge.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <chrono>
#include <climits>
#include <cstdlib>

#include "context.hpp"
#include "trisolve.hpp"

namespace ipxk {

namespace {

using u64 = unsigned long long;
constexpr int kPanel = 32;           // columns per panel of the dense elimination
constexpr int kPanelThreads = 1024;
constexpr u64 kNoKey = ~0ull;

int grid_for(int64_t n) { return (int)std::min<int64_t>(4096, std::max<int64_t>(1, (n + kBlock - 1) / kBlock)); }
int bits_for(int64_t n) { int b = 1; while ((int64_t(1) << b) < n) b++; return b; }

#define IPXK_GRID_STRIDE(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

struct Tmp {
    DevBuf<unsigned char> bytes;
    void* need(size_t n) { if (bytes.size() < n) bytes.resize(n); return bytes.get(); }
};

void scan_exclusive(Tmp& T, const int* in, int* out, size_t n, hipStream_t s) {
    size_t bytes = 0;
    IPXK_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, n, rocprim::plus<int>(), s));
    IPXK_HIP(rocprim::exclusive_scan(T.need(bytes), bytes, in, out, 0, n, rocprim::plus<int>(), s));
}

// grows a buffer to at least `need` elements, keeping its first `used` ones
template <class T>
void grow_keep(DevBuf<T>& b, size_t used, size_t need, hipStream_t s) {
    if (need <= b.size()) return;
    DevBuf<T> nb(std::max(need, 2 * b.size()));
    if (used) IPXK_HIP(hipMemcpyAsync(nb.get(), b.get(), used * sizeof(T), hipMemcpyDeviceToDevice, s));
    IPXK_HIP(hipStreamSynchronize(s));
    b = std::move(nb);
}

// ---- row-wise copy ------------------------------------------------------------------------------
__global__ void lu_expand_kernel(int dim, const int* __restrict__ Bp, const int* __restrict__ Bi, int* __restrict__ colof,
                                 int* __restrict__ keys, int* __restrict__ pos, int* __restrict__ rc, int* __restrict__ cc,
                                 int* bad) {
    IPXK_GRID_STRIDE(j, dim) {
        cc[j] = Bp[j + 1] - Bp[j];
        for (int p = Bp[j]; p < Bp[j + 1]; p++) {
            const int i = Bi[p];
            if (i < 0 || i >= dim) { *bad = 1; continue; }
            colof[p] = (int)j;
            keys[p] = i;
            pos[p] = p;
            atomicAdd(rc + i, 1);
        }
    }
}
__global__ void lu_rows_kernel(int64_t nb, const int* __restrict__ pos_sorted, const int* __restrict__ colof, int* __restrict__ Rj) {
    IPXK_GRID_STRIDE(q, nb) Rj[q] = colof[pos_sorted[q]];
}
__global__ void lu_fill_int_kernel(int64_t n, int v, int* a) { IPXK_GRID_STRIDE(i, n) a[i] = v; }
__global__ void lu_fill_u64_kernel(int64_t n, u64 v, u64* a) { IPXK_GRID_STRIDE(i, n) a[i] = v; }

// ---- singleton rounds ---------------------------------------------------------------------------
// While the rounds run, rstage / cstage hold -1 for active rows / columns and the TAG of the round that pivoted
// them otherwise (2 * iteration for a column round, + 1 for a row round); the dense pivot stages are assigned
// afterwards by one sort of (tag, column index | row index): the order inside a round is by index whatever the
// order in which the threads ran.
struct Rounds {
    int dim;
    const int *Bp, *Bi;
    const double* Bx;
    const int *Rp, *Rj, *Rpos;
    int *rstage, *cstage, *rc, *cc;
    double* pivot;
    unsigned char* ckind;
    int *cand, *claim, *pivrow;
    u64 *cand_bits, *claim_abs;
    int* counters;        // [8 + b]: iteration b of the batch found a pivot
    double abstol, pivottol;
};

__global__ void lu_col_find_kernel(Rounds R) {
    IPXK_GRID_STRIDE(j, R.dim) {
        int cr = -1;
        if (R.cstage[j] == -1 && R.cc[j] == 1) {          // -1: active; -2: a spike (torn), never a round pivot
            for (int p = R.Bp[j]; p < R.Bp[j + 1]; p++) {
                const int i = R.Bi[p];
                if (R.rstage[i] >= 0) continue;
                if (fabs(R.Bx[p]) >= R.abstol) { cr = i; atomicMin(R.claim + i, (int)j); }
                break;
            }
        }
        R.cand[j] = cr;
    }
}
__global__ __launch_bounds__(kBlock) void lu_col_commit_kernel(Rounds R, int tag, int slot) {
    IPXK_GRID_STRIDE(j, R.dim) {
        if (!(R.cand[j] >= 0 && R.claim[R.cand[j]] == (int)j)) continue;
        R.counters[slot] = 1;                      // "this iteration found a pivot" (same value from every writer)
        const int i = R.cand[j];
        for (int p = R.Bp[j]; p < R.Bp[j + 1]; p++)
            if (R.Bi[p] == i) R.pivot[j] = R.Bx[p];
        R.cstage[j] = tag;
        R.rstage[i] = tag;
        R.pivrow[j] = i;
        R.ckind[j] = 1;
        R.claim[i] = INT_MAX;
        // row i leaves the active submatrix: its other columns lose an active entry (none of them is a pivot of
        // this round: such a column would have had two active entries)
        for (int q = R.Rp[i]; q < R.Rp[i + 1]; q++) {
            const int j2 = R.Rj[q];
            if (j2 != (int)j && R.cstage[j2] < 0) atomicSub(R.cc + j2, 1);
        }
    }
}
__global__ void lu_row_find_kernel(Rounds R) {
    IPXK_GRID_STRIDE(i, R.dim) {
        int cj = -1;
        u64 bits = 0;
        if (R.rstage[i] < 0 && R.rc[i] == 1) {
            for (int q = R.Rp[i]; q < R.Rp[i + 1]; q++) {
                const int j = R.Rj[q];
                if (R.cstage[j] != -1) continue;
                const double a = fabs(R.Bx[R.Rpos[q]]);
                double colmax = 0.0;
                for (int p = R.Bp[j]; p < R.Bp[j + 1]; p++)
                    if (R.rstage[R.Bi[p]] < 0) colmax = fmax(colmax, fabs(R.Bx[p]));
                if (a >= R.abstol && a >= R.pivottol * colmax) {
                    cj = j;
                    bits = (u64)__double_as_longlong(a);
                    atomicMax(R.claim_abs + j, bits);
                }
                break;
            }
        }
        R.cand[i] = cj;
        R.cand_bits[i] = bits;
    }
}
__global__ void lu_row_pick_kernel(Rounds R) {
    IPXK_GRID_STRIDE(i, R.dim) {
        const int j = R.cand[i];
        if (j >= 0 && R.cand_bits[i] == R.claim_abs[j]) atomicMin(R.claim + j, (int)i);
    }
}
__global__ __launch_bounds__(kBlock) void lu_row_commit_kernel(Rounds R, int tag, int slot) {
    IPXK_GRID_STRIDE(i, R.dim) {
        if (!(R.cand[i] >= 0 && R.claim[R.cand[i]] == (int)i)) continue;
        R.counters[slot] = 1;
        const int j = R.cand[i];
        R.rstage[i] = tag;
        R.cstage[j] = tag;
        R.pivrow[j] = (int)i;
        R.ckind[j] = 2;
        R.claim[j] = INT_MAX;          // (a loser reads the winner's index or this: neither is its own)
        R.claim_abs[j] = 0;
        // column j leaves: its other active rows lose an active entry (they become entries of L)
        for (int p = R.Bp[j]; p < R.Bp[j + 1]; p++) {
            const int r = R.Bi[p];
            if (r == (int)i) R.pivot[j] = R.Bx[p];
            else if (R.rstage[r] < 0) atomicSub(R.rc + r, 1);
        }
    }
}
// # column / row singletons: block-wise sums of the kinds, one atomic per workgroup
__global__ __launch_bounds__(kBlock) void lu_count_kinds_kernel(int dim, const unsigned char* __restrict__ ckind, int* counters) {
    __shared__ int s1, s2;
    if (threadIdx.x == 0) { s1 = 0; s2 = 0; }
    __syncthreads();
    int n1 = 0, n2 = 0;
    IPXK_GRID_STRIDE(j, dim) { n1 += ckind[j] == 1; n2 += ckind[j] == 2; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { n1 += __shfl_xor(n1, d, 64); n2 += __shfl_xor(n2, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s1, n1); atomicAdd(&s2, n2); }
    __syncthreads();
    if (threadIdx.x == 0) { if (s1) atomicAdd(counters + 1, s1); if (s2) atomicAdd(counters + 2, s2); }
}
// tags -> sort keys; after the sort: dense stages
__global__ void lu_stage_keys_kernel(int dim, const int* __restrict__ cstage, const int* __restrict__ pivrow,
                                     u64* __restrict__ keys, int* __restrict__ vals) {
    IPXK_GRID_STRIDE(j, dim) {
        const int tag = cstage[j];
        keys[j] = tag < 0 ? kNoKey : ((u64)(unsigned)tag << 32) | (unsigned)((tag & 1) ? pivrow[j] : (int)j);
        vals[j] = (int)j;
    }
}
__global__ void lu_stage_assign_kernel(int npiv, const int* __restrict__ order, const int* __restrict__ pivrow,
                                       int* __restrict__ cstage, int* __restrict__ rstage) {
    IPXK_GRID_STRIDE(q, npiv) {
        const int j = order[q];
        cstage[j] = (int)q;
        rstage[pivrow[j]] = (int)q;
    }
}

// ---- tearing ------------------------------------------------------------------------------------
// candidates for spikes: active columns by descending number of active entries, then ascending index
__global__ void lu_tear_keys_kernel(int dim, const int* __restrict__ cstage, const int* __restrict__ cc, u64* __restrict__ keys) {
    IPXK_GRID_STRIDE(j, dim)
        keys[j] = cstage[j] == -1 ? ((u64)(0x7fffffffu - (unsigned)cc[j]) << 32) | (unsigned)j : kNoKey;
}
// the first `take` candidates become spikes: they leave the active columns, their active rows lose an entry
__global__ void lu_tear_apply_kernel(Rounds R, const u64* __restrict__ sorted, int take) {
    IPXK_GRID_STRIDE(t, take) {
        const int j = (int)(sorted[t] & 0xffffffffull);
        R.cstage[j] = -2;
        for (int p = R.Bp[j]; p < R.Bp[j + 1]; p++) {
            const int i = R.Bi[p];
            if (R.rstage[i] < 0) atomicSub(R.rc + i, 1);
        }
    }
}
// the entries of L found by the rounds, keyed (stage of the row | stage of the pivot): an entry (r, j) of a row
// singleton column j below its pivot, i.e. r was still active when j was pivoted.  Rows that were never pivoted
// count as stages npiv.. in ascending row order.
__global__ void lu_lentry_keys_kernel(int64_t nb, const int* __restrict__ colof, const int* __restrict__ Bi,
                                      const double* __restrict__ Bx, const unsigned char* __restrict__ ckind,
                                      const int* __restrict__ pivrow, const int* __restrict__ rstage, const int* __restrict__ cstage,
                                      const int* __restrict__ rloc, const double* __restrict__ pivot, int npiv,
                                      u64* __restrict__ key, double* __restrict__ val) {
    IPXK_GRID_STRIDE(p, nb) {
        const int j = colof[p], r = Bi[p];
        key[p] = kNoKey;
        val[p] = 0.0;
        if (ckind[j] != 2 || r == pivrow[j]) continue;
        const int rs = rstage[r] >= 0 ? rstage[r] : npiv + rloc[r];
        if (rs < cstage[j]) continue;                          // pivoted before j: an entry of U
        key[p] = ((u64)(unsigned)rs << 32) | (unsigned)cstage[j];
        val[p] = Bx[p] / pivot[j];
    }
}
// L entries of the rows of each half-round (tag): which launches of the substitution have work
__global__ void lu_tagwork_kernel(int ntags, const ipxint* __restrict__ tagptr, const ipxint* __restrict__ lrp, int* __restrict__ work) {
    IPXK_GRID_STRIDE(t, ntags) work[t] = (int)(lrp[tagptr[t + 1]] - lrp[tagptr[t]]);
}
constexpr int kSpikeBatch = 64;      // spikes per dense block: one lane each
// X[stage of row][lane] = entries of the batch's spikes (X zero before)
// (all spike kernels: blockIdx.y = the batch of 64 spikes within a group of batches that travel together -- they are independent --,
// its block of X `xs` doubles behind the previous one's; kb = # spikes in all)
__global__ void lu_spike_scatter_kernel(int kb, int c0, size_t xs, const int* __restrict__ bcol, const int* __restrict__ Bp,
                                        const int* __restrict__ Bi, const double* __restrict__ Bx, const int* __restrict__ rstage,
                                        const int* __restrict__ rloc, int npiv, double* __restrict__ X) {
    const int l = blockIdx.x;
    c0 += kSpikeBatch * blockIdx.y;
    X += xs * blockIdx.y;
    if (c0 + l >= kb) return;
    const int j = bcol[c0 + l];
    for (int p = Bp[j] + threadIdx.x; p < Bp[j + 1]; p += blockDim.x) {
        const int r = Bi[p];
        const int rs = rstage[r] >= 0 ? rstage[r] : npiv + rloc[r];
        X[(size_t)rs * kSpikeBatch + l] = Bx[p];
    }
}
// rows [s0, s1) of the substitution: x[r] -= l_rj * x[stage of j] for the row's entries of L in pivot order, products
// rounded before they are subtracted (the elimination's own arithmetic); one wavefront per row, one lane per spike
__global__ __launch_bounds__(kBlock) void lu_spike_round_kernel(int s0, int s1, size_t xs, const ipxint* __restrict__ lrp, const u64* __restrict__ lkey,
                                                                const double* __restrict__ lval, double* __restrict__ X) {
    const int lane = threadIdx.x & 63;
    X += xs * blockIdx.y;
    for (int64_t r = (int64_t)s0 + (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); r < s1; r += (int64_t)gridDim.x * (kBlock / 64)) {
        const ipxint e0 = lrp[r], e1 = lrp[r + 1];
        if (e0 == e1) continue;
        double acc = X[(size_t)r * kSpikeBatch + lane];
        for (ipxint e = e0; e < e1; e++) {
            const size_t src = (size_t)(lkey[e] & 0xffffffffull);
            const double prod = lval[e] * X[src * kSpikeBatch + lane];
            acc = acc - prod;
        }
        X[(size_t)r * kSpikeBatch + lane] = acc;
    }
}
// The same for a RUN of consecutive half-rounds [t0, t1) that are small (a few rows each): one workgroup takes them one after the
// other with a barrier in between, instead of one launch per half-round -- thousands of tears leave thousands of half-rounds of
// a handful of rows (a 12 000 x 30 000 LP through the drop-in solver: 2.2 million launches of the kernel above, 37 % of all
// kernel time and more in launch latency).  A row's arithmetic is the same: its entries in pivot order, products rounded first.
constexpr int kSpikeRunThreads = 1024;
__global__ __launch_bounds__(kSpikeRunThreads) void lu_spike_run_kernel(int t0, int t1, size_t xs, const ipxint* __restrict__ tagptr,
                                                                        const ipxint* __restrict__ lrp, const u64* __restrict__ lkey,
                                                                        const double* __restrict__ lval, double* X) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    X += xs * blockIdx.y;
    for (int t = t0; t < t1; t++) {
        const int64_t s0 = tagptr[t], s1 = tagptr[t + 1];
        for (int64_t r = s0 + wave; r < s1; r += kSpikeRunThreads / 64) {
            const ipxint e0 = lrp[r], e1 = lrp[r + 1];
            if (e0 == e1) continue;
            double acc = X[(size_t)r * kSpikeBatch + lane];
            for (ipxint e = e0; e < e1; e++) {
                const size_t src = (size_t)(lkey[e] & 0xffffffffull);
                const double prod = lval[e] * X[src * kSpikeBatch + lane];
                acc = acc - prod;
            }
            X[(size_t)r * kSpikeBatch + lane] = acc;
        }
        __syncthreads();               // the rows of the next half-round read what this one wrote
    }
}
// the batch's part of the dense block: rows that were never pivoted
__global__ void lu_spike_dense_kernel(int kb, int c0, size_t xs, int npiv, const double* __restrict__ X, double* __restrict__ D) {
    c0 += kSpikeBatch * blockIdx.y;
    X += xs * blockIdx.y;
    const int nlanes = min(kSpikeBatch, kb - c0);
    IPXK_GRID_STRIDE(e, (int64_t)kb * nlanes) {
        const int l = (int)(e % nlanes), t = (int)(e / nlanes);
        D[(size_t)(c0 + l) * kb + t] = X[(size_t)(npiv + t) * kSpikeBatch + l];
    }
}
// the batch's entries in pivoted rows (future entries of U): counted, then appended as (bump column, stage, value)
__global__ void lu_spike_count_kernel(int kb, int c0, size_t xs, int npiv, const double* __restrict__ X, int* count) {
    c0 += kSpikeBatch * blockIdx.y;
    X += xs * blockIdx.y;
    const int nlanes = min(kSpikeBatch, kb - c0);
    int mine = 0;
    IPXK_GRID_STRIDE(e, (int64_t)npiv * kSpikeBatch) mine += (int)(e % kSpikeBatch) < nlanes && X[e] != 0.0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}
__global__ void lu_spike_append_kernel(int kb, int c0, size_t xs, int npiv, const double* __restrict__ X, int* cursor,
                                       int* __restrict__ spk_c, int* __restrict__ spk_s, double* __restrict__ spk_v) {
    c0 += kSpikeBatch * blockIdx.y;
    X += xs * blockIdx.y;
    const int nlanes = min(kSpikeBatch, kb - c0);
    IPXK_GRID_STRIDE(e, (int64_t)npiv * kSpikeBatch) {
        const int l = (int)(e % kSpikeBatch);
        const double v = X[e];
        if (l >= nlanes || v == 0.0) continue;
        const int at = atomicAdd(cursor, 1);           // the order is irrelevant: the entries are sorted by key later
        spk_c[at] = c0 + l;
        spk_s[at] = (int)(e / kSpikeBatch);
        spk_v[at] = v;
    }
}

// ---- bump ---------------------------------------------------------------------------------------
__global__ void lu_active_flag_kernel(int dim, const int* __restrict__ stage, int* __restrict__ flag) {
    IPXK_GRID_STRIDE(i, dim) flag[i] = stage[i] < 0 ? 1 : 0;
}
__global__ void lu_compact_kernel(int dim, const int* __restrict__ flag, const int* __restrict__ rank, int* __restrict__ loc,
                                  int* __restrict__ list) {
    IPXK_GRID_STRIDE(i, dim) {
        loc[i] = flag[i] ? rank[i] : -1;
        if (flag[i]) list[rank[i]] = (int)i;
    }
}
__global__ __launch_bounds__(kBlock) void lu_dense_fill_kernel(int kb, const int* __restrict__ bcol, const int* __restrict__ Bp,
                                                               const int* __restrict__ Bi, const double* __restrict__ Bx,
                                                               const int* __restrict__ rloc, double* __restrict__ D) {
    const int lane = threadIdx.x & 63;                            // a wavefront per bump column
    for (int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); c < kb; c += gridDim.x * (kBlock / 64)) {
        const int j = bcol[c];
        for (int p = Bp[j] + lane; p < Bp[j + 1]; p += 64) {
            const int r = rloc[Bi[p]];
            if (r >= 0) D[(size_t)c * kb + r] = Bx[p];
        }
    }
}

struct Dense {
    int kb;
    double* D;             // column-major kb x kb
    int *brstep, *bcstep;  // pivot step of a bump row / column, -1 while unpivoted / for a dependent column
    int* bstep;            // [0] # pivots so far; [1] # pivots of the current (sub-)panel; [3] # pivots of the outer panel before it
                           // (look-ahead: two sets of bstep / prow / pcol, used by the outer panels alternately)
    int *prow, *pcol;      // rows / columns of the current panel's pivots
    double abstol;
};

// One panel of columns [c0, c1): partial pivoting (largest |entry| among the unpivoted rows, ties: smaller
// row), scaling, update of the panel's later columns.  One workgroup; the panel lives in L2.
__global__ __launch_bounds__(kPanelThreads) void lu_panel_kernel(Dense A, int c0, int c1) {
    __shared__ double red_v[kPanelThreads / 64];
    __shared__ int red_r[kPanelThreads / 64];
    __shared__ double su[kPanel];
    __shared__ int s_pr;
    __shared__ double s_piv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kb = A.kb;
    int np = 0;
    int step = A.bstep[0];
    for (int c = c0; c < c1; c++) {
        double* col = A.D + (size_t)c * kb;
        double best = 0.0;
        int br = INT_MAX;
        for (int r = tid; r < kb; r += kPanelThreads)
            if (A.brstep[r] < 0) {
                const double a = fabs(col[r]);
                if (a > best) { best = a; br = r; }
            }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double ov = __shfl_xor(best, d, 64);
            const int orr = __shfl_xor(br, d, 64);
            if (ov > best || (ov == best && orr < br)) { best = ov; br = orr; }
        }
        if (lane == 0) { red_v[wave] = best; red_r[wave] = br; }
        __syncthreads();
        if (tid == 0) {
            double bv = 0.0;
            int r = INT_MAX;
            for (int w = 0; w < kPanelThreads / 64; w++)
                if (red_v[w] > bv || (red_v[w] == bv && red_r[w] < r)) { bv = red_v[w]; r = red_r[w]; }
            if (r == INT_MAX || !(bv >= A.abstol) || bv == 0.0) {
                s_pr = -1;
                A.bcstep[c] = -1;
            } else {
                s_pr = r;
                s_piv = col[r];
                A.brstep[r] = step;
                A.bcstep[c] = step;
                A.prow[np] = r;
                A.pcol[np] = c;
            }
        }
        __syncthreads();
        const int pr = s_pr;
        if (pr < 0) continue;                 // dependent column (uniform over the workgroup)
        const double piv = s_piv;
        np++;
        step++;
        if (tid < c1 - c - 1) su[tid] = A.D[(size_t)(c + 1 + tid) * kb + pr];
        __syncthreads();
        for (int r = tid; r < kb; r += kPanelThreads) {
            if (A.brstep[r] >= 0) continue;   // pivoted rows (this step's included) keep their values
            const double l = col[r] / piv;
            col[r] = l;
            for (int c2 = c + 1; c2 < c1; c2++) {
                const double u = su[c2 - c - 1];
                if (u != 0.0) A.D[(size_t)c2 * kb + r] -= l * u;
            }
        }
        __syncthreads();
    }
    if (tid == 0) { A.bstep[0] = step; A.bstep[1] = np; }
}

// The same for bumps of at most kPanelThreads rows: a thread owns one row of the panel in registers, the pivot
// row travels through LDS; the panel is read and written once.  Same arithmetic, same order.  (The column steps
// are instantiated one by one: v[] must be indexed by constants to stay in registers.)
// (two copies of everything, used alternately by consecutive pivot steps: a step then needs two barriers, not four --
// every thread combines the wavefronts' candidates itself, and no barrier has to protect the buffers for the next step)
struct PanelShared {
    double red_v[2][kPanelThreads / 64];
    int red_r[2][kPanelThreads / 64];
    double su[2][kPanel];
};
// the pivot row of a step from the wavefronts' candidates (largest |entry|, ties: smaller row); -1: none / no column
__device__ __forceinline__ int panel_pivot_row(const PanelShared& sh, int par, bool col, double abstol, bool* dependent) {
    // (every lane reads one wavefront's candidate and the sixteen are combined by shuffles: the same selection -- a total order --
    // as a scan of all sixteen by every thread, at a third of the LDS traffic: the scan was 1 us of a 4.5 us pivot step)
    static_assert(kPanelThreads / 64 == 16, "sixteen wavefronts");
    double bv = sh.red_v[par][threadIdx.x & 15];
    int rr = sh.red_r[par][threadIdx.x & 15];
    if (!(bv > 0.0)) { bv = 0.0; rr = INT_MAX; }          // (no candidate, or not a number: never a pivot)
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
        const double ov = __shfl_xor(bv, d, 64);
        const int orr = __shfl_xor(rr, d, 64);
        if (ov > bv || (ov == bv && orr < rr)) { bv = ov; rr = orr; }
    }
    *dependent = col && (rr == INT_MAX || !(bv >= abstol) || bv == 0.0);
    return col && !*dependent ? rr : -1;
}
template <int T>
__device__ __forceinline__ void panel_small_steps(const Dense& A, PanelShared& sh, double (&v)[kPanel], int c0, int c1, int r,
                                                  bool& active, int& np, int& step) {
    if constexpr (T < kPanel) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const bool col = c0 + T < c1;                  // uniform
        double best = (col && active) ? fabs(v[T]) : 0.0;
        int br = best > 0.0 ? r : INT_MAX;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double ov = __shfl_xor(best, d, 64);
            const int orr = __shfl_xor(br, d, 64);
            if (ov > best || (ov == best && orr < br)) { best = ov; br = orr; }
        }
        constexpr int par = T & 1;
        if (lane == 0) { sh.red_v[par][wave] = best; sh.red_r[par][wave] = br; }
        __syncthreads();
        bool dependent;
        const int pr = panel_pivot_row(sh, par, col, A.abstol, &dependent);      // uniform over the workgroup
        if (tid == 0) {
            if (dependent) A.bcstep[c0 + T] = -1;
            else if (pr >= 0) {
                A.brstep[pr] = step;
                A.bcstep[c0 + T] = step;
                A.prow[np] = pr;
                A.pcol[np] = c0 + T;
            }
        }
        if (pr >= 0 && r == pr) {
            active = false;
#pragma unroll
            for (int t2 = 0; t2 < kPanel; t2++) sh.su[par][t2] = v[t2];
        }
        __syncthreads();
        if (pr >= 0) { np++; step++; }
        if (pr >= 0 && active) {
            const double l = v[T] / sh.su[par][T];
            v[T] = l;
#pragma unroll
            for (int t2 = T + 1; t2 < kPanel; t2++) {
                const double u = sh.su[par][t2];
                if (c0 + t2 < c1 && u != 0.0) v[t2] -= l * u;
            }
        }
        panel_small_steps<T + 1>(A, sh, v, c0, c1, r, active, np, step);
    }
}
__global__ __launch_bounds__(kPanelThreads) void lu_panel_small_kernel(Dense A, int c0, int c1) {
    __shared__ PanelShared sh;
    const int kb = A.kb, r = threadIdx.x;
    const bool have = r < kb;
    bool active = have && A.brstep[r] < 0;
    double v[kPanel];
#pragma unroll
    for (int t = 0; t < kPanel; t++) v[t] = (have && c0 + t < c1) ? A.D[(size_t)(c0 + t) * kb + r] : 0.0;
    int np = 0;
    int step = A.bstep[0];
    panel_small_steps<0>(A, sh, v, c0, c1, r, active, np, step);
    if (have) {
#pragma unroll
        for (int t = 0; t < kPanel; t++)
            if (c0 + t < c1) A.D[(size_t)(c0 + t) * kb + r] = v[t];
    }
    if (threadIdx.x == 0) { A.bstep[0] = step; A.bstep[1] = np; }
}

// Bumps of more than kPanelThreads rows: panels of kNarrow columns, a thread owns R rows of the panel in registers
// (R * kPanelThreads >= rows).  Same arithmetic, same order; the two kernels that follow a panel take the number
// of its pivots from bstep[1], so they serve both panel widths.
constexpr int kNarrow = 8;            // panel width with 4 rows per thread (bumps of 2049 .. 4096 rows)
constexpr int kNarrowWide = 16;       // ... with 2 rows per thread (1025 .. 2048 rows): half the panels, the same registers
constexpr int kNarrowDeep = 4;        // ... with 8 rows per thread (4097 .. 8192 rows)
constexpr int kNarrowHuge = 2;        // ... with 16 rows per thread (8193 .. 16384 rows: the dense fall-back of a bump that tearing cannot cut down)
constexpr int kNarrowGiant = 1;       // ... with 32 rows per thread (16385 .. 32768 rows: what the elimination rounds leave of the bump of an IPM basis of 50 000 rows and more)
constexpr int kDenseHardMax = 32 * 1024;
constexpr int kNarrowWideMax = 16;    // the widest sub-panel
template <int R, int W, int T>
__device__ __forceinline__ void panel_multi_steps(const Dense& A, PanelShared& sh, double (&v)[R][W], int c0, int c1,
                                                  unsigned& active, int& np, int& step) {
    if constexpr (T < W) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const bool col = c0 + T < c1;                  // uniform
        double best = 0.0;
        int br = INT_MAX;
#pragma unroll
        for (int q = 0; q < R; q++) {
            const double a = (col && ((active >> q) & 1u)) ? fabs(v[q][T]) : 0.0;
            if (a > best) { best = a; br = tid + q * kPanelThreads; }      // rows ascend with q: the first maximum stays
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double ov = __shfl_xor(best, d, 64);
            const int orr = __shfl_xor(br, d, 64);
            if (ov > best || (ov == best && orr < br)) { best = ov; br = orr; }
        }
        constexpr int par = T & 1;
        if (lane == 0) { sh.red_v[par][wave] = best; sh.red_r[par][wave] = br; }
        __syncthreads();
        bool dependent;
        const int pr = panel_pivot_row(sh, par, col, A.abstol, &dependent);      // uniform over the workgroup
        if (tid == 0) {
            if (dependent) A.bcstep[c0 + T] = -1;
            else if (pr >= 0) {
                A.brstep[pr] = step;
                A.bcstep[c0 + T] = step;
                A.prow[np] = pr;
                A.pcol[np] = c0 + T;
            }
        }
        if (pr >= 0 && (pr % kPanelThreads) == tid) {
            const int qp = pr / kPanelThreads;
#pragma unroll
            for (int q = 0; q < R; q++)
                if (q == qp) {
                    active &= ~(1u << q);
#pragma unroll
                    for (int t2 = 0; t2 < W; t2++) sh.su[par][t2] = v[q][t2];
                }
        }
        __syncthreads();
        if (pr >= 0) {
            np++; step++;
#pragma unroll
            for (int q = 0; q < R; q++)
                if ((active >> q) & 1u) {
                    const double l = v[q][T] / sh.su[par][T];
                    v[q][T] = l;
#pragma unroll
                    for (int t2 = T + 1; t2 < W; t2++) {
                        const double u = sh.su[par][t2];
                        if (c0 + t2 < c1 && u != 0.0) v[q][t2] -= l * u;
                    }
                }
        }
        panel_multi_steps<R, W, T + 1>(A, sh, v, c0, c1, active, np, step);
    }
}
// Two-level panels (round 4): the kernel factorizes a SUB-panel [c0, c1) of an outer panel of kPanel columns; its pivots
// are appended to the outer panel's list (first_inner: the list starts again), bstep[3] = # pivots of the outer panel
// before this sub-panel, bstep[1] = # pivots of this sub-panel.  The sub-panel's update is applied to the rest of
// the outer panel only; the whole trailing matrix is updated once per outer panel with all its pivots (in pivot
// order, one rounded product at a time: every entry still receives exactly the arithmetic of the column-by-column
// elimination).  Before: a full-matrix update per 8- or 16-column panel, and bumps of more than 4096 rows went
// through lu_panel_kernel (one workgroup, the panel in L2: 1.2 ms per panel, 0.3 s for a 6000-row bump).
// (usub / c1o: the previous sub-panel's rows of U in the rest of the outer panel, columns [c0, c1o), which lu_subpanel_update_kernel
// left in a side buffer -- every one of its workgroups needs the rows as they were -- are written to their places here first.)
template <int R, int W>
__global__ __launch_bounds__(kPanelThreads) void lu_panel_multi_kernel(Dense A, int c0, int c1, int first_inner, const double* __restrict__ usub = nullptr,
                                                                       int c1o = 0, const int* __restrict__ step_src = nullptr) {
    __shared__ PanelShared sh;
    const int kb = A.kb, tid = threadIdx.x;
    const int base = first_inner ? 0 : A.bstep[3] + A.bstep[1];
    if (usub && !first_inner) {
        const int pf = A.bstep[3], pn = A.bstep[1], nc = c1o - c0;
        for (int e = tid; e < pn * nc; e += kPanelThreads) {
            const int t = e / nc, x = e - t * nc;
            if (t > 0) A.D[(size_t)(c0 + x) * kb + A.prow[pf + t]] = usub[t * kPanel + x];      // (the first pivot's row is unchanged)
        }
        __syncthreads();
    }
    A.prow += base; A.pcol += base;
    unsigned active = 0, have = 0;
    double v[R][W];
#pragma unroll
    for (int q = 0; q < R; q++) {
        const int r = tid + q * kPanelThreads;
        if (r < kb) { have |= 1u << q; if (A.brstep[r] < 0) active |= 1u << q; }
#pragma unroll
        for (int t = 0; t < W; t++) v[q][t] = (r < kb && c0 + t < c1) ? A.D[(size_t)(c0 + t) * kb + r] : 0.0;
    }
    int np = 0;
    int step = (first_inner && step_src) ? step_src[0] : A.bstep[0];       // (look-ahead: the count so far is in the other set)
    panel_multi_steps<R, W, 0>(A, sh, v, c0, c1, active, np, step);
#pragma unroll
    for (int q = 0; q < R; q++)
        if ((have >> q) & 1u) {
            const int r = tid + q * kPanelThreads;
#pragma unroll
            for (int t = 0; t < W; t++)
                if (c0 + t < c1) A.D[(size_t)(c0 + t) * kb + r] = v[q][t];
        }
    if (tid == 0) { A.bstep[0] = step; A.bstep[1] = np; A.bstep[3] = base; }
}

// COOPERATIVE OUTER PANEL (round 5).  The two-level scheme above spends a one-workgroup launch (24 us) per sub-panel of 2 ... 16 columns
// plus a launch (15 us) that carries the sub-panel's update to the rest of the outer panel: 316 us per 32 columns at 8000 rows, 620 us
// beyond 8192 rows -- 70 of the 80 ms of a 7350-row block, and the largest item of a whole LP solve
// (profiles/r05_lp_dropin_24000_kernel_summary_before_eta_rework.txt).  Here the WHOLE outer panel of kPanel columns is factorized by ONE launch
// of G <= 64 workgroups of 256 threads that share the rows (R = 1 / 2 rows of the panel per thread in registers: up to 8192 / 32 768 rows).  Per column ONE exchange: every workgroup publishes its best candidate (|entry|, row) TOGETHER with that row's 32
// panel entries (write-through stores, drained, then one agent-scope add to a counter); everyone polls the counter, reads the G
// messages past L1, takes the same winner (largest |entry|, ties: smaller row -- a total order, so the choice does not depend on G)
// and has the pivot row with it.  No second exchange, no sub-panels, no side buffer.  Every entry still receives its updates one pivot at
// a time in pivot order, products rounded before they are subtracted: the factors equal the other kernels' bit for bit.
// All G workgroups must be resident at once: G <= 64 (two per compute unit fit) on 256 compute units, nothing else on the stream (the
// look-ahead's late update runs under a CU mask that leaves 32 units free); a poll that does not see its word within kCoopSpinLimit polls raises an abort flag
// that ends every workgroup, and the factorization fails loudly instead of hanging.
constexpr int kCoopThreads = 256;
constexpr int kCoopMaxG = 64;
constexpr int kCoopSlot = kPanel + 2;                 // a message: |entry|, row, the row's kPanel entries
constexpr int kCoopSpinLimit = 1 << 22;
struct CoopShared {
    double red_v[kCoopThreads / 64];
    int red_r[kCoopThreads / 64];
    double row[kPanel];
    double slots[kCoopMaxG * kCoopSlot];
    int abort;
    int part, G;            // this workgroup among the participants, their number
    int plain;              // all participants share one XCD (checked): messages and resets by plain stores that stay in its L2
};
struct Coop {
    double* slots;          // [5][G][kCoopSlot], every word the sentinel or a message
    int set0;               // the set of this launch's first step (the steps of a factorization take the five sets in turn)
    int* abort_flag;
    int xcd_mode;           // 1: only the workgroups with blockIdx % 8 == 0 take part (one XCD under the round-robin dispatch of gfx950)
    unsigned epoch;         // of the placement check
    unsigned long long* xcc_slots;
};
constexpr long long kCoopSentinel = 0x7ff8dead5eed0001LL;       // a quiet NaN with a payload of its own
template <int R, int T>
__device__ __forceinline__ void coop_steps(const Dense& A, CoopShared& sh, const Coop& C, double (&v)[R][kPanel], int c0, int c1, unsigned& active,
                                           int& np, int& step, bool& dead) {
    if constexpr (T < kPanel) {
        if (c0 + T >= c1 || dead) return;              // uniform over the grid
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = sh.G, part = sh.part;
        const bool plain = sh.plain != 0;
        const int row0 = part * R * kCoopThreads;
        if (wave == 1 + (T & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the slot this wave reset two steps ago (see the exchange)
        double best = 0.0;
        int br = INT_MAX;
#pragma unroll
        for (int q = 0; q < R; q++) {
            const double a = ((active >> q) & 1u) ? fabs(v[q][T]) : 0.0;
            if (a > best) { best = a; br = row0 + q * kCoopThreads + tid; }       // rows ascend with q: the first maximum stays
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double ov = __shfl_xor(best, d, 64);
            const int orr = __shfl_xor(br, d, 64);
            if (ov > best || (ov == best && orr < br)) { best = ov; br = orr; }
        }
        if (lane == 0) { sh.red_v[wave] = best; sh.red_r[wave] = br; }
        __syncthreads();
        double bv = 0.0;
        int rr = INT_MAX;
#pragma unroll
        for (int w = 0; w < kCoopThreads / 64; w++) {
            const double ov = sh.red_v[w];
            const int orr = sh.red_r[w];
            if (ov > bv || (ov == bv && orr < rr)) { bv = ov; rr = orr; }
        }
        if (!(bv > 0.0)) { bv = 0.0; rr = INT_MAX; }       // (no candidate, or not a number: never a pivot)
        if (rr != INT_MAX && (rr - row0) % kCoopThreads == tid) {
            const int qo = (rr - row0) / kCoopThreads;
#pragma unroll
            for (int q = 0; q < R; q++)
                if (q == qo) {
#pragma unroll
                    for (int t2 = 0; t2 < kPanel; t2++) sh.row[t2] = v[q][t2];
                }
        }
        __syncthreads();
        // ---- the exchange: a message IS its own flag.  FIVE sets of message slots are used in turn; a slot holds a sentinel (a NaN
        // pattern no candidate, row index or matrix entry is) until its workgroup writes the step's message there, word by word with
        // write-through stores and nothing else -- no drain, no counter: the readers poll every word past L1 until it is not the
        // sentinel.  A slot is reset three steps before its next use (it held the messages of step t - 2, and by the time a workgroup
        // has read all messages of step t everyone has published t - 1, i.e. finished reading t - 2); waves 1 and 2 take turns, and the
        // wave that reset a slot at step t waits for that store at the START of step t + 2 -- two steps later, so the wait is free --
        // in front of the barriers that precede the publication of step t + 2.  So whoever has seen a workgroup's message of step u
        // finds that workgroup's slot of step u + 1 reset or already written, never stale.
        const int set = (C.set0 + T) % 5;
        double* mine = C.slots + ((size_t)set * G + part) * kCoopSlot;
        if (wave == 0 && lane < kCoopSlot) {
            const double x = lane == 0 ? bv : lane == 1 ? __longlong_as_double((long long)rr) : sh.row[lane - 2];
            if (plain) __hip_atomic_store(mine + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // stays in the XCD's L2
            else __hip_atomic_store(mine + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);               // write-through
        }
        const double* all = C.slots + (size_t)set * G * kCoopSlot;
        int gone = 0;
        {
            // all of a thread's words are requested at once (independent loads: one round trip); only those still holding the
            // sentinel are asked for again
            constexpr int kPer = (kCoopMaxG * kCoopSlot + kCoopThreads - 1) / kCoopThreads;
            const int nw = G * kCoopSlot;
            double x[kPer];
#pragma unroll
            for (int k = 0; k < kPer; k++) {
                const int e = tid + k * kCoopThreads;
                x[k] = e < nw ? __hip_atomic_load(all + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
            int spins = 0;
            for (;;) {
                bool pending = false;
#pragma unroll
                for (int k = 0; k < kPer; k++) pending |= __double_as_longlong(x[k]) == kCoopSentinel;
                if (!pending) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kCoopSpinLimit || ((spins & 1023) == 0 && __hip_atomic_load(C.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(C.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gone = 1;
                    break;
                }
#pragma unroll
                for (int k = 0; k < kPer; k++)
                    if (__double_as_longlong(x[k]) == kCoopSentinel) x[k] = __hip_atomic_load(all + tid + k * kCoopThreads, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < kPer; k++) {
                const int e = tid + k * kCoopThreads;
                if (e < nw) sh.slots[e] = x[k];
            }
        }
        if (gone) sh.abort = 1;
        __syncthreads();
        if (sh.abort) { dead = true; return; }
        if (wave == 1 + (T & 1) && lane < kCoopSlot) {
            double* ahead = C.slots + ((size_t)((set + 3) % 5) * G + part) * kCoopSlot;
            if (plain) __hip_atomic_store(ahead + lane, __longlong_as_double(kCoopSentinel), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(ahead + lane, __longlong_as_double(kCoopSentinel), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- the same winner everywhere
        double wv = lane < G ? sh.slots[lane * kCoopSlot] : 0.0;
        int wr = lane < G ? (int)__double_as_longlong(sh.slots[lane * kCoopSlot + 1]) : INT_MAX;
        int wg = lane;
        if (!(wv > 0.0)) { wv = 0.0; wr = INT_MAX; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double ov = __shfl_xor(wv, d, 64);
            const int orr = __shfl_xor(wr, d, 64);
            const int og = __shfl_xor(wg, d, 64);
            if (ov > wv || (ov == wv && orr < wr)) { wv = ov; wr = orr; wg = og; }
        }
        static_assert(kCoopMaxG <= 64, "the winner is combined over the 64 lanes of a wavefront");
        wv = __shfl(wv, 0, 64); wr = __shfl(wr, 0, 64); wg = __shfl(wg, 0, 64);
        const bool dependent = wr == INT_MAX || !(wv >= A.abstol) || wv == 0.0;
        const int pr = dependent ? -1 : wr;
        if (part == 0 && tid == 0) {
            if (dependent) A.bcstep[c0 + T] = -1;
            else {
                A.brstep[pr] = step;
                A.bcstep[c0 + T] = step;
                A.prow[np] = pr;
                A.pcol[np] = c0 + T;
            }
        }
        if (pr >= 0) {
            const double* su = sh.slots + wg * kCoopSlot + 2;
            if (pr >= row0 && pr < row0 + R * kCoopThreads && (pr - row0) % kCoopThreads == tid) active &= ~(1u << ((pr - row0) / kCoopThreads));
            np++; step++;
            const double piv = su[T];
#pragma unroll
            for (int q = 0; q < R; q++)
                if ((active >> q) & 1u) {
                    const double l = v[q][T] / piv;
                    v[q][T] = l;
#pragma unroll
                    for (int t2 = T + 1; t2 < kPanel; t2++) {
                        const double u = su[t2];
                        if (c0 + t2 < c1 && u != 0.0) v[q][t2] -= l * u;
                    }
                }
        }
        // (sh.slots / sh.row / red_* are rewritten only after the next step's first barrier, which every thread reaches after this read)
        coop_steps<R, T + 1>(A, sh, C, v, c0, c1, active, np, step, dead);
    }
}
template <int R>
__global__ __launch_bounds__(kCoopThreads) __attribute__((amdgpu_waves_per_eu(1, 2))) void lu_panel_coop_kernel(Dense A, Coop C, int c0, int c1, const int* __restrict__ step_src) {
    __shared__ CoopShared sh;
    if (C.xcd_mode && (blockIdx.x & 7)) return;
    const int kb = A.kb, tid = threadIdx.x;
    const int part = C.xcd_mode ? blockIdx.x >> 3 : blockIdx.x, G = C.xcd_mode ? (gridDim.x + 7) >> 3 : gridDim.x;
    const int row0 = part * R * kCoopThreads;
    if (tid == 0) { sh.abort = 0; sh.part = part; sh.G = G; sh.plain = 0; }
    if (C.xcd_mode && tid < 64) {
        // the placement is an observation, not a contract (as for the one-XCD runs of the sweeps, trisolve.hip): every participant
        // publishes the XCD it runs on and reads everybody else's; plain stores only if all agree -- all see the same ids and decide alike
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xff;
        if (tid == 0) __hip_atomic_store(C.xcc_slots + part, ((unsigned long long)C.epoch << 32) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        for (int i = tid; i < G; i += 64) {
            unsigned long long w;
            int spins = 0;
            while (((w = __hip_atomic_load(C.xcc_slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != C.epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kCoopSpinLimit) { __hip_atomic_store(C.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            same &= (unsigned)(w & 0xff) == xcc && (w >> 32) == C.epoch;
        }
        same = __all(same);
        if (tid == 0) sh.plain = same ? 1 : 0;
    }
    unsigned active = 0, have = 0;
    double v[R][kPanel];
#pragma unroll
    for (int q = 0; q < R; q++) {
        const int r = row0 + q * kCoopThreads + tid;
        if (r < kb) { have |= 1u << q; if (A.brstep[r] < 0) active |= 1u << q; }
#pragma unroll
        for (int t = 0; t < kPanel; t++) v[q][t] = (r < kb && c0 + t < c1) ? A.D[(size_t)(c0 + t) * kb + r] : 0.0;
    }
    __syncthreads();
    int np = 0;
    int step = step_src ? step_src[0] : A.bstep[0];       // (look-ahead: the count so far is in the other set)
    bool dead = false;
    coop_steps<R, 0>(A, sh, C, v, c0, c1, active, np, step, dead);
    if (dead) return;                                      // nothing was written: the host finds the abort flag
#pragma unroll
    for (int q = 0; q < R; q++)
        if ((have >> q) & 1u) {
            const int r = row0 + q * kCoopThreads + tid;
#pragma unroll
            for (int t = 0; t < kPanel; t++)
                if (c0 + t < c1) A.D[(size_t)(c0 + t) * kb + r] = v[q][t];
        }
    if (part == 0 && tid == 0) { A.bstep[0] = step; A.bstep[1] = np; A.bstep[3] = 0; }
}

// The panel's rows of U in the trailing columns: row prow[t] of column c2 receives the updates of the panel's
// earlier pivots, in pivot order.  One thread per trailing column.
// mode 0: the pivots of the last panel call, prow[0 .. bstep[1]) (one-level panels); 1: those of the last SUB-panel,
// prow[bstep[3] .. bstep[3] + bstep[1]); 2: all pivots of the outer panel, prow[0 .. bstep[3] + bstep[1]).  Columns [c1, cend).
__device__ __forceinline__ void panel_pivots(const Dense& A, int mode, int* first, int* np) {
    *first = mode == 1 ? A.bstep[3] : 0;
    *np = mode == 2 ? A.bstep[3] + A.bstep[1] : A.bstep[1];
}
__global__ __launch_bounds__(kBlock) void lu_panel_rows_kernel(Dense A, int c1, int cend, int mode, double* __restrict__ ubuf = nullptr, int ldu = 0) {
    __shared__ double l11[kPanel][kPanel];
    __shared__ int prow[kPanel];
    int first, np;
    panel_pivots(A, mode, &first, &np);
    A.prow += first; A.pcol += first;
    const int kb = A.kb;
    for (int e = threadIdx.x; e < kPanel * kPanel; e += kBlock) {
        const int t2 = e / kPanel, t = e % kPanel;
        l11[t2][t] = (t < t2 && t2 < np) ? A.D[(size_t)A.pcol[t] * kb + A.prow[t2]] : 0.0;
    }
    if (threadIdx.x < kPanel) prow[threadIdx.x] = threadIdx.x < np ? A.prow[threadIdx.x] : 0;
    __syncthreads();
    if (np == 0) return;
    IPXK_GRID_STRIDE(cc, cend - c1) {
        double* col = A.D + (size_t)(c1 + cc) * kb;
        double v[kPanel];
#pragma unroll
        for (int t = 0; t < kPanel; t++) v[t] = t < np ? col[prow[t]] : 0.0;
#pragma unroll
        for (int t = 0; t < kPanel; t++) {
            const double u = v[t];
            if (t < np && u != 0.0) {
#pragma unroll
                for (int t2 = t + 1; t2 < kPanel; t2++)
                    if (t2 < np) v[t2] -= l11[t2][t] * u;
            }
            // (keeps the LDS reads of the later pivots from being hoisted up here: all 496 at once need 256 registers and 684
            // bytes of scratch per lane -- the kernel took 38-54 us; with the fence 40 registers)
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int t = 1; t < kPanel; t++)
            if (t < np) col[prow[t]] = v[t];
        if (ubuf) {                       // the finished rows of U, pivot by pivot, contiguous along the columns (MFMA trailing update)
#pragma unroll
            for (int t = 0; t < kPanel; t++) ubuf[(size_t)t * ldu + cc] = t < np ? v[t] : 0.0;
        }
    }
}

// Trailing update on the matrix cores (round 4): D[r][c] -= sum_t L[r][t] U[t][c] over the np <= 32 pivots of an outer
// panel as v_mfma_f64_16x16x4_f64 products, for bumps of more than kMfmaMinRows rows.  The transposed product is
// formed (A operand = U', from the compact copy the rows kernel leaves; B operand = L, a column of D per pivot), so that
// the lane index of a result runs along the ROWS of D: loads and stores of a tile are 128-byte segments of D's columns.
// A workgroup takes 64 rows x 64 columns, a wavefront 16 rows x 64 columns (L fragment loaded once, 8 k-steps).  Rows
// pivoted already keep their values (their entries are entries of U).  The sums are accumulated by the matrix unit
// (fused, k ascending): no longer the one-rounded-product-at-a-time arithmetic of the restatement -- the factors are
// judged by the stability estimate (src/lu_factorization.cc:87-127) and agree with the restatement's to ~1e-13.
typedef double lu_d4 __attribute__((ext_vector_type(4)));
// (look-ahead: the update of columns [c1, cend) may run while the next outer panel is being factorized; a row that panel pivots
// meanwhile carries a step >= this panel's count bstep[0] and is still live for THIS update.  ubuf's columns start at cu.)
__global__ __launch_bounds__(kBlock) void lu_trailing_mfma_kernel(Dense A, const double* __restrict__ ubuf, int ldu, int c1, int cend, int cu) {
    const int np = A.bstep[3] + A.bstep[1], kb = A.kb;
    if (np == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int r = blockIdx.x * 64 + wave * 16 + li;               // this lane's row of D (B operand / result column)
    const int rc = min(r, kb - 1);
    const int rs = A.brstep[rc];
    const bool live = r < kb && (rs < 0 || rs >= A.bstep[0]);
    // all rows of the wavefront's 16 pivoted already: nothing to do
    if (__ballot(live) == 0ull) return;
    double lf[8];                                                 // L[r][t = 4 ks + lk]
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
        const int t = 4 * ks + lk;
        lf[ks] = t < np ? A.D[(size_t)A.pcol[t] * kb + rc] : 0.0;
    }
    const int cb = c1 + blockIdx.y * 64;
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
        const int c0 = cb + ct * 16;
        if (c0 >= cend) break;
        const int ca = min(c0 + li, cend - 1) - cu;               // A operand: column c0 + li of the trailing part
        lu_d4 acc;
#pragma unroll
        for (int q = 0; q < 4; q++) {                             // result q: column c0 + lk + 4 q, row r
            const int c = min(c0 + lk + 4 * q, cend - 1);
            acc[q] = A.D[(size_t)c * kb + rc];
        }
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const double u = -ubuf[(size_t)(4 * ks + lk) * ldu + ca];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(u, lf[ks], acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = c0 + lk + 4 * q;
            if (live && c < cend) A.D[(size_t)c * kb + r] = acc[q];
        }
    }
}

// Trailing update: D[r][c2] -= sum over the panel's pivots t (in order, one rounded product at a time) of
// multiplier[r][t] * U[t][c2], for the rows not pivoted yet.  64 x 64 tile per workgroup, 4 x 4 per thread.
__global__ __launch_bounds__(kBlock) void lu_trailing_kernel(Dense A, int c1, int cend, int mode) {
    __shared__ double Ls[kPanel][64];
    __shared__ double Us[kPanel][64];
    __shared__ int live[64];
    int first, np;
    panel_pivots(A, mode, &first, &np);
    A.prow += first; A.pcol += first;
    const int kb = A.kb;
    if (np == 0) return;
    const int r0 = blockIdx.x * 64, cb = c1 + blockIdx.y * 64;
    const int tid = threadIdx.x;
    for (int e = tid; e < kPanel * 64; e += kBlock) {
        const int t = e / 64, x = e % 64;
        const int r = r0 + x, c2 = cb + x;
        Ls[t][x] = (t < np && r < kb) ? A.D[(size_t)A.pcol[t] * kb + r] : 0.0;
        Us[t][x] = (t < np && c2 < cend) ? A.D[(size_t)c2 * kb + A.prow[t]] : 0.0;
    }
    if (tid < 64) live[tid] = (r0 + tid < kb && A.brstep[r0 + tid] < 0) ? 1 : 0;
    __syncthreads();
    // a thread's 4 x 4 entries in registers, the pivots in the outer loop: sixteen independent chains instead of one (each entry
    // still receives its products one at a time in pivot order, and none for a zero of U)
    const int tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
    bool on[4][4];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int xc = ty + 16 * b, xr = tx + 16 * a;
            on[a][b] = cb + xc < cend && live[xr];
            acc[a][b] = on[a][b] ? A.D[(size_t)(cb + xc) * kb + r0 + xr] : 0.0;
        }
    for (int t = 0; t < np; t++) {
        double l[4], u[4];
#pragma unroll
        for (int a = 0; a < 4; a++) l[a] = Ls[t][tx + 16 * a];
#pragma unroll
        for (int b = 0; b < 4; b++) u[b] = Us[t][ty + 16 * b];
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const double next = acc[a][b] - l[a] * u[b];
                acc[a][b] = u[b] != 0.0 ? next : acc[a][b];
            }
    }
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int a = 0; a < 4; a++)
            if (on[a][b]) A.D[(size_t)(cb + ty + 16 * b) * kb + r0 + tx + 16 * a] = acc[a][b];
}

// The update of the rest of the outer panel, columns [c1, cend), by the pivots of the last SUB-panel, in one launch (before:
// lu_panel_rows_kernel mode 1 + lu_trailing_kernel mode 1).  Every workgroup (64 rows) forms the sub-panel's rows of U for those
// columns itself in LDS -- np <= 16 pivots x <= 28 columns, the arithmetic of lu_panel_rows_kernel -- and updates its rows with
// them; workgroup 0 leaves the rows of U in `usub` ([t][x], kPanel apart), and the next lu_panel_multi_kernel writes them to
// their places: written here, they would race with the other workgroups' reads of the rows as they were.
__global__ __launch_bounds__(kBlock) void lu_subpanel_update_kernel(Dense A, int c1, int cend, double* __restrict__ usub) {
    __shared__ double l11[kNarrowWideMax][kNarrowWideMax];
    __shared__ double Us[kNarrowWideMax][kPanel];
    __shared__ double Ls[kNarrowWideMax][64];
    __shared__ int prow[kNarrowWideMax];
    __shared__ int live[64];
    const int first = A.bstep[3], np = A.bstep[1], kb = A.kb, nc = cend - c1, tid = threadIdx.x;
    if (np == 0 || nc <= 0) return;
    A.prow += first; A.pcol += first;
    const int r0 = blockIdx.x * 64;
    for (int e = tid; e < np * np; e += kBlock) {
        const int t2 = e / np, t = e - t2 * np;
        l11[t2][t] = t < t2 ? A.D[(size_t)A.pcol[t] * kb + A.prow[t2]] : 0.0;
    }
    for (int e = tid; e < np * nc; e += kBlock) {
        const int t = e / nc, x = e - t * nc;
        Us[t][x] = A.D[(size_t)(c1 + x) * kb + A.prow[t]];
    }
    for (int e = tid; e < np * 64; e += kBlock) {
        const int t = e / 64, x = e & 63;
        Ls[t][x] = r0 + x < kb ? A.D[(size_t)A.pcol[t] * kb + r0 + x] : 0.0;
    }
    if (tid < np) prow[tid] = A.prow[tid];
    if (tid < 64) live[tid] = (r0 + tid < kb && A.brstep[r0 + tid] < 0) ? 1 : 0;
    __syncthreads();
    if (tid < nc) {                       // the rows of U of column c1 + tid, pivot after pivot
        for (int t = 0; t < np; t++) {
            const double u = Us[t][tid];
            if (u != 0.0)
                for (int t2 = t + 1; t2 < np; t2++) Us[t2][tid] -= l11[t2][t] * u;
        }
        if (blockIdx.x == 0)
            for (int t = 0; t < np; t++) usub[t * kPanel + tid] = Us[t][tid];
    }
    __syncthreads();
    // 64 rows x nc columns: a thread takes a row and every fourth column
    const int xr = tid & 63;
    if (!live[xr]) return;
    for (int xc = tid >> 6; xc < nc; xc += kBlock / 64) {
        double* d = A.D + (size_t)(c1 + xc) * kb + r0 + xr;
        double acc = *d;
        for (int t = 0; t < np; t++) {
            const double u = Us[t][xc];
            if (u != 0.0) acc -= Ls[t][xr] * u;
        }
        *d = acc;
    }
}

// ---- elimination rounds ---------------------------------------------------------------------------
// (2c in the header comment.)  The CURRENT matrix of a round is a compact problem of its own: rows and columns renumbered to the
// active ones in ascending order (grow / gcol: their indices in B), every entry active, CSC plus a row-wise index.
struct Sparse {
    int dim;
    const int *Bp, *Bi, *colof;
    const double* Bx;
    const int *Rp, *Rj, *Rpos;
    const int *rc, *cc;
    int *candrow, *cost;
    u64 *key, *rowbest;
    int* stats;           // [0] cheapest cost, [1] # candidates, [2] cost limit, [3] # winners, [8 + b] # candidates of cost < 2^b
    double abstol, pivottol;
};
constexpr int kSpStats = 48;
// one candidate per column: among its entries that pass the absolute and the relative threshold, the one in the shortest row
// (ties: larger |entry|, then smaller row); cost = (row count - 1)(column count - 1).  One wavefront per column.
__global__ __launch_bounds__(kBlock) void sp_cand_kernel(Sparse S) {
    __shared__ int s_hist[33], s_min, s_n;
    if (threadIdx.x < 33) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_min = INT_MAX; s_n = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int64_t j = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); j < S.dim; j += (int64_t)gridDim.x * (kBlock / 64)) {
        const int p0 = S.Bp[j], p1 = S.Bp[j + 1];
        double colmax = 0.0;
        for (int p = p0 + lane; p < p1; p += 64) colmax = fmax(colmax, fabs(S.Bx[p]));
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) colmax = fmax(colmax, __shfl_xor(colmax, d, 64));
        int bi = INT_MAX, brc = INT_MAX;
        double ba = 0.0;
        const double rel = S.pivottol * colmax;
        for (int p = p0 + lane; p < p1; p += 64) {
            const double a = fabs(S.Bx[p]);
            if (!(a >= S.abstol && a >= rel)) continue;
            const int i = S.Bi[p], r = S.rc[i];
            if (r < brc || (r == brc && (a > ba || (a == ba && i < bi)))) { bi = i; brc = r; ba = a; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int oi = __shfl_xor(bi, d, 64), orc = __shfl_xor(brc, d, 64);
            const double oa = __shfl_xor(ba, d, 64);
            if (oi != INT_MAX && (orc < brc || (orc == brc && (oa > ba || (oa == ba && oi < bi))))) { bi = oi; brc = orc; ba = oa; }
        }
        if (lane) continue;
        S.candrow[j] = bi == INT_MAX ? -1 : bi;
        S.key[j] = kNoKey;
        if (bi == INT_MAX) continue;
        const long long c64 = (long long)(brc - 1) * (long long)(p1 - p0 - 1);
        const int c = (int)(c64 < 0x7fffffffLL ? c64 : 0x7fffffffLL);
        S.cost[j] = c;
        atomicMin(&s_min, c);
        atomicAdd(&s_hist[c == 0 ? 0 : 32 - __clz(c)], 1);
        atomicAdd(&s_n, 1);
    }
    __syncthreads();
    if (threadIdx.x < 33 && s_hist[threadIdx.x]) atomicAdd(S.stats + 8 + threadIdx.x, s_hist[threadIdx.x]);
    if (threadIdx.x == 0 && s_n) { atomicMin(S.stats + 0, s_min); atomicAdd(S.stats + 1, s_n); }
}
// the candidates that cost at most max(4, twice the cheapest) compete, and at least a quarter of all candidates (the smallest
// power of two that admits so many)
__global__ void sp_limit_kernel(int* stats) {
    if (blockIdx.x || threadIdx.x) return;
    const long long ncand = stats[1];
    long long limit = -1;
    if (ncand > 0) {
        limit = 2LL * stats[0] > 4 ? 2LL * stats[0] : 4;
        long long cum = 0;
        for (int b = 0; b <= 32; b++) {
            cum += stats[8 + b];
            if (cum * 4 >= ncand) { const long long q = (1LL << b) - 1; if (q > limit) limit = q; break; }
        }
        if (limit > 0x7fffffffLL) limit = 0x7fffffffLL;
    }
    stats[2] = (int)limit;
}
// a row keeps its best candidate (cost, then column index)
__global__ void sp_key_kernel(Sparse S) {
    const int limit = S.stats[2];
    IPXK_GRID_STRIDE(j, S.dim) {
        if (S.candrow[j] < 0 || S.cost[j] > limit) continue;
        const u64 k = ((u64)(unsigned)S.cost[j] << 32) | (unsigned)j;
        S.key[j] = k;
        atomicMin(S.rowbest + S.candrow[j], k);
    }
}
// a contender (the best of its row) wins unless a better contender has an entry in its pivot row or its pivot row in this
// column: the winners' pivots form a diagonal block.  nupd: the products a winner sends out (pivot row x pivot column, all pairs).
// One wavefront per column.
__global__ __launch_bounds__(kBlock) void sp_win_kernel(Sparse S, int* __restrict__ winner, int* __restrict__ nupd, int* bad) {
    const int lane = threadIdx.x & 63;
    for (int64_t j = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); j < S.dim; j += (int64_t)gridDim.x * (kBlock / 64)) {
        const u64 k = S.key[j];
        bool win = false;
        int i = -1;
        if (k != kNoKey && S.rowbest[S.candrow[j]] == k) {           // (uniform over the wavefront)
            i = S.candrow[j];
            bool lose = false;
            for (int q = S.Rp[i] + lane; q < S.Rp[i + 1]; q += 64) {
                const int j2 = S.Rj[q];
                if (j2 == (int)j) continue;
                const u64 k2 = S.key[j2];
                if (k2 < k && S.rowbest[S.candrow[j2]] == k2) lose = true;       // (k2 < k implies k2 is a key)
            }
            for (int p = S.Bp[j] + lane; p < S.Bp[j + 1]; p += 64)
                if (S.Bi[p] != i && S.rowbest[S.Bi[p]] < k) lose = true;
            win = __ballot(lose) == 0;
        }
        if (lane) continue;
        winner[j] = win ? 1 : 0;
        int n = 0;
        if (win) {
            const long long n64 = (long long)S.rc[i] * S.cc[j];
            if (n64 > (1LL << 28)) *bad = 1; else n = (int)n64;
            atomicAdd(reinterpret_cast<unsigned long long*>(S.stats + 42), (unsigned long long)n);       // the round's total in 64 bits (the int scan below wraps beyond 2^31)
            atomicAdd(S.stats + 3, 1);
            atomicAdd(S.stats + 4, S.rc[i] + S.cc[j] - 1);       // the entries of the pivot row and column leave the matrix
        }
        nupd[j] = n;
    }
}
constexpr int kSpRowsBySort = 1 << 18;
__global__ void sp_rowkeys_kernel(int64_t nnz, const int* __restrict__ Bi, int* __restrict__ keys, int* __restrict__ pos) {
    IPXK_GRID_STRIDE(p, nnz) { keys[p] = Bi[p]; pos[p] = (int)p; }
}
// the row-wise index of the current matrix.  The order of a row's entries is left to the atomics: nothing depends on it (the
// winners' test is a conjunction over the row, and a winner's updates have distinct positions, so the stable sort puts them in
// the same places whatever the order in which they were written).
__global__ void sp_rowcount_kernel(int64_t nnz, const int* __restrict__ Bi, int* __restrict__ rc) {
    IPXK_GRID_STRIDE(p, nnz) atomicAdd(rc + Bi[p], 1);
}
__global__ void sp_rowfill_kernel(int64_t nnz, const int* __restrict__ Bi, const int* __restrict__ colof, const int* __restrict__ Rp,
                                  int* __restrict__ cursor, int* __restrict__ Rj, int* __restrict__ Rpos) {
    IPXK_GRID_STRIDE(p, nnz) {
        const int i = Bi[p];
        const int at = Rp[i] + atomicAdd(cursor + i, 1);
        Rj[at] = colof[p];
        Rpos[at] = (int)p;
    }
}
struct SparseGlobal {     // where a round's pivots are recorded: the arrays of the singleton rounds, indexed as B is
    int *rstage, *cstage, *pivrow;
    double* pivot;
    unsigned char* ckind;
};
__global__ void sp_commit_kernel(Sparse S, const int* __restrict__ winner, const int* __restrict__ grow, const int* __restrict__ gcol,
                                 int tag, SparseGlobal G, int* __restrict__ rstL, int* __restrict__ cstL, double* __restrict__ pivl) {
    IPXK_GRID_STRIDE(j, S.dim) {
        if (!winner[j]) continue;
        const int i = S.candrow[j];
        double piv = 0.0;
        for (int p = S.Bp[j]; p < S.Bp[j + 1]; p++)
            if (S.Bi[p] == i) piv = S.Bx[p];
        pivl[j] = piv;
        rstL[i] = tag;
        cstL[j] = tag;
        const int gi = grow[i], gj = gcol[j];
        G.rstage[gi] = tag;
        G.cstage[gj] = tag;
        G.pivrow[gj] = gi;
        G.pivot[gj] = piv;
        G.ckind[gj] = 5;
    }
}
// the indices in B of the rows (columns) that stay
__global__ void sp_map_kernel(int n, const int* __restrict__ list, const int* __restrict__ gold, int* __restrict__ gnew) {
    IPXK_GRID_STRIDE(r, n) gnew[r] = gold ? gold[list[r]] : list[r];
}
// the entries of the current matrix: those whose row and column stay are keyed (new column | new row) for the next matrix, the
// others join the list of finished entries with their indices in B and their present values
__global__ void sp_entries_kernel(int64_t nnz, const int* __restrict__ Bi, const int* __restrict__ colof, const double* __restrict__ Bx,
                                  const int* __restrict__ newrow, const int* __restrict__ newcol, const int* __restrict__ grow,
                                  const int* __restrict__ gcol, u64* __restrict__ key, double* __restrict__ val, int* __restrict__ keep,
                                  int* cursor, int* __restrict__ Erow, int* __restrict__ Ecol, double* __restrict__ Eval) {
    IPXK_GRID_STRIDE(p, nnz) {
        const int i = Bi[p], j = colof[p];
        const int nr = newrow[i], nc = newcol[j];
        const bool stays = nr >= 0 && nc >= 0;
        if (keep) keep[p] = stays ? 1 : 0;
        const u64 leaving = __ballot(!stays);                 // one atomic per wavefront (the order in E is irrelevant: the assembly sorts by key)
        if (stays) {
            key[p] = ((u64)(unsigned)nc << 32) | (unsigned)nr;
            val[p] = Bx[p];
        } else {
            key[p] = kNoKey;
            val[p] = 0.0;
            const int lane = threadIdx.x & 63, leader = __ffsll((long long)leaving) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(cursor, __popcll(leaving));
            base = __shfl(base, leader, 64);
            const int at = base + __popcll(leaving & ((1ull << lane) - 1));
            Erow[at] = grow ? grow[i] : i;
            Ecol[at] = gcol ? gcol[j] : j;
            Eval[at] = Bx[p];
        }
    }
}
__global__ void sp_carry_kernel(int64_t nnz, const int* __restrict__ keep, const int* __restrict__ pos, const u64* __restrict__ key,
                                const double* __restrict__ val, u64* __restrict__ okey, double* __restrict__ oval) {
    IPXK_GRID_STRIDE(p, nnz) {
        if (!keep[p]) continue;
        okey[pos[p]] = key[p];
        oval[pos[p]] = val[p];
    }
}
// the updates -(a_i'j / pivot) * a_ij' of the winners, behind the entries, winner after winner in ascending order of the column
// (the sort is stable: equal positions are then summed in that order); one thread per product, its winner found in the offsets
__global__ __launch_bounds__(kBlock) void sp_updates_kernel(Sparse S, const int* __restrict__ uoff, int64_t nupd, const double* __restrict__ pivl,
                                                            const int* __restrict__ newrow, const int* __restrict__ newcol, int64_t base,
                                                            u64* __restrict__ key, double* __restrict__ val) {
    IPXK_GRID_STRIDE(t, nupd) {
        int lo = 0, hi = S.dim;                       // the last column j with uoff[j] <= t (it has products: uoff[j + 1] > t)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (uoff[mid] <= (int)t) lo = mid; else hi = mid;
        }
        const int j = lo, i = S.candrow[j], p0 = S.Bp[j], q0 = S.Rp[i], nq = S.Rp[i + 1] - q0;
        const int u = (int)t - uoff[j];
        const int pi = u / nq, qi = u - pi * nq;
        const int p = p0 + pi, q = q0 + qi;
        const int i2 = S.Bi[p], j2 = S.Rj[q];
        if (i2 == i || j2 == j) { key[base + t] = kNoKey; val[base + t] = 0.0; continue; }
        const double l = S.Bx[p] / pivl[j];
        const double prod = l * S.Bx[S.Rpos[q]];
        key[base + t] = ((u64)(unsigned)newcol[j2] << 32) | (unsigned)newrow[i2];
        val[base + t] = -prod;
    }
}
// equal keys are summed in order; a sum that is exactly zero leaves the pattern
__global__ void sp_combine_kernel(int64_t n, const u64* __restrict__ key, const double* __restrict__ val, int* __restrict__ flag,
                                  double* __restrict__ sum) {
    IPXK_GRID_STRIDE(e, n + 1) {
        int keep = 0;
        if (e < n) {
            const u64 k = key[e];
            if (k != kNoKey && (e == 0 || key[e - 1] != k)) {
                double acc = val[e];
                for (int64_t f = e + 1; f < n && key[f] == k; f++) acc = acc + val[f];
                sum[e] = acc;
                keep = acc != 0.0;
            }
        }
        flag[e] = keep;
    }
}
__global__ void sp_compact_kernel(int64_t n, const u64* __restrict__ key, const int* __restrict__ flag, const int* __restrict__ pos,
                                  const double* __restrict__ sum, int* __restrict__ Bi, int* __restrict__ colof, double* __restrict__ Bx) {
    IPXK_GRID_STRIDE(e, n) {
        if (!flag[e]) continue;
        const int at = pos[e];
        Bi[at] = (int)(key[e] & 0xffffffffull);
        colof[at] = (int)(key[e] >> 32);
        Bx[at] = sum[e];
    }
}
// column pointers from the (sorted) columns of the entries
__global__ void sp_colptr_kernel(int dim, int nnz, const int* __restrict__ colof, int* __restrict__ Bp) {
    IPXK_GRID_STRIDE(j, (int64_t)dim + 1) {
        int lo = 0, hi = nnz;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (colof[mid] < (int)j) lo = mid + 1; else hi = mid;
        }
        Bp[j] = lo;
    }
}
__global__ void sp_colcount_kernel(int dim, const int* __restrict__ Bp, int* __restrict__ cc) {
    IPXK_GRID_STRIDE(j, dim) cc[j] = Bp[j + 1] - Bp[j];
}
// the dense block's columns in ascending order of their number of entries (ties: index)
__global__ void sp_colorder_keys_kernel(int kb, const int* __restrict__ cc, u64* __restrict__ keys) {
    IPXK_GRID_STRIDE(c, kb) keys[c] = ((u64)(unsigned)cc[c] << 32) | (unsigned)c;
}
__global__ void sp_colorder_apply_kernel(int kb, const u64* __restrict__ sorted, const int* __restrict__ gcol, int* __restrict__ bcol,
                                         int* __restrict__ cloc, int* __restrict__ cposl) {
    IPXK_GRID_STRIDE(t, kb) {
        const int c = (int)(sorted[t] & 0xffffffffull);
        bcol[t] = gcol[c];
        cloc[gcol[c]] = (int)t;
        cposl[c] = (int)t;
    }
}
__global__ void sp_dense_fill_kernel(int64_t nnz, int kb, const int* __restrict__ Bi, const int* __restrict__ colof, const double* __restrict__ Bx,
                                     const int* __restrict__ cposl, double* __restrict__ D) {
    IPXK_GRID_STRIDE(p, nnz) D[(size_t)cposl[colof[p]] * kb + Bi[p]] = Bx[p];
}

// stages of the bump's pivots; dependent columns and left-over rows are flagged for the ranking that follows
__global__ void lu_bump_stage_kernel(int kb, int base, const int* __restrict__ step, const int* __restrict__ list,
                                     int* __restrict__ stage, int* __restrict__ flag, unsigned char* kind) {
    IPXK_GRID_STRIDE(x, kb) {
        const int s = step[x];
        flag[x] = s < 0 ? 1 : 0;
        if (s >= 0) {
            stage[list[x]] = base + s;
            if (kind) kind[list[x]] = 3;
        }
    }
}
__global__ void lu_bump_rest_kernel(int kb, int base, const int* __restrict__ flag, const int* __restrict__ rank,
                                    const int* __restrict__ list, int* __restrict__ stage, unsigned char* kind,
                                    ipxint* dependent) {
    IPXK_GRID_STRIDE(x, kb) {
        if (!flag[x]) continue;
        stage[list[x]] = base + rank[x];
        if (kind) kind[list[x]] = 4;
        if (dependent) dependent[rank[x]] = base + rank[x];
    }
}

// ---- assembly -----------------------------------------------------------------------------------
struct Assemble {
    int dim, kb;
    const int *Bp, *Bi, *colof;
    const double* Bx;
    const int *rstage, *cstage, *rloc, *cloc, *brow, *bcol, *bcstep;
    const double *pivot, *D;
    const unsigned char* ckind;
    u64 *lkey, *ukey;
    double *lval, *uval;
    int tearing;           // the bump's columns are spikes: their entries above the dense block come from the substitution
    int shift;             // keys are (stage of the column << shift) | stage of the row, 2^shift > dim: the sorts run over 2 * shift bits
};
__global__ void lu_keys_sparse_kernel(Assemble A, int64_t nb) {
    IPXK_GRID_STRIDE(p, nb) {
        const int j = A.colof[p], i = A.Bi[p];
        if (A.ckind[j] == 4) continue;                          // replaced by a unit column
        if (A.cloc[j] >= 0 && (A.tearing || A.rloc[i] >= 0)) continue;   // bump x bump: from the dense result; a spike: lu_keys_spike_kernel
        const int k = A.cstage[j], s = A.rstage[i];
        const u64 key = ((u64)(unsigned)k << A.shift) | (unsigned)s;
        if (s <= k) { A.ukey[p] = key; A.uval[p] = A.Bx[p]; }
        else { A.lkey[p] = key; A.lval[p] = A.Bx[p] / A.pivot[j]; }
    }
}
__global__ void lu_keys_dense_kernel(Assemble A, int64_t nb) {
    const int kb = A.kb;
    IPXK_GRID_STRIDE(e, (int64_t)kb * kb) {
        const int c = (int)(e / kb), r = (int)(e % kb);
        if (A.bcstep[c] < 0) continue;
        const int k = A.cstage[A.bcol[c]], s = A.rstage[A.brow[r]];
        const double v = A.D[e];
        const u64 key = ((u64)(unsigned)k << A.shift) | (unsigned)s;
        if (s == k) { A.ukey[nb + e] = key; A.uval[nb + e] = v; }
        else if (v != 0.0) {
            if (s < k) { A.ukey[nb + e] = key; A.uval[nb + e] = v; }
            else { A.lkey[nb + e] = key; A.lval[nb + e] = v; }
        }
    }
}
__global__ void lu_keys_spike_kernel(Assemble A, int64_t off, int nspk, const int* __restrict__ spk_c, const int* __restrict__ spk_s,
                                     const double* __restrict__ spk_v) {
    IPXK_GRID_STRIDE(e, nspk) {
        const int c = spk_c[e];
        if (A.bcstep[c] < 0) continue;                          // a dependent spike is replaced by a unit column
        const int k = A.cstage[A.bcol[c]];
        A.ukey[off + e] = ((u64)(unsigned)k << A.shift) | (unsigned)spk_s[e];
        A.uval[off + e] = spk_v[e];
    }
}
__global__ void lu_keys_unit_kernel(Assemble A, int64_t off) {
    IPXK_GRID_STRIDE(j, A.dim) {
        if (A.ckind[j] != 4) continue;
        const int k = A.cstage[j];
        A.ukey[off + j] = ((u64)(unsigned)k << A.shift) | (unsigned)k;
        A.uval[off + j] = 1.0;
    }
}
// column pointers of a factor from its sorted keys: ptr[k] = first key >= (k << shift), k = 0..dim
__global__ void lu_colptr_kernel(int dim, int64_t n, const u64* __restrict__ keys, ipxint* __restrict__ ptr, int shift = 32) {
    IPXK_GRID_STRIDE(k, (int64_t)dim + 1) {
        const u64 want = (u64)k << shift;
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < want) lo = mid + 1; else hi = mid;
        }
        ptr[k] = lo;
    }
}
__global__ void lu_rowidx_kernel(int64_t nz, const u64* __restrict__ keys, ipxint* __restrict__ idx, int shift = 32) {
    IPXK_GRID_STRIDE(q, nz) idx[q] = (ipxint)(keys[q] & ((1ull << shift) - 1));
}
__global__ void lu_perm_kernel(int dim, const int* __restrict__ stage, ipxint* __restrict__ perm, int* bad) {
    IPXK_GRID_STRIDE(i, dim) {
        const int k = stage[i];
        if (k < 0 || k >= dim) *bad = 1; else perm[k] = i;
    }
}
// compact CSC of B = AI[:, basis] from the structural matrix resident on the device (slack columns: unit)
__global__ void lu_basis_count_kernel(int m, int n, const ipxint* __restrict__ basis, const int* __restrict__ Ap,
                                      int* __restrict__ cnt, int* bad) {
    IPXK_GRID_STRIDE(k, m) {
        const ipxint j = basis[k];
        if (j < 0 || j >= (ipxint)n + m) { *bad = 1; cnt[k] = 0; continue; }
        cnt[k] = j < n ? Ap[j + 1] - Ap[j] : 1;
    }
}
__global__ void lu_basis_fill_kernel(int m, int n, const ipxint* __restrict__ basis, const int* __restrict__ Ap,
                                     const int* __restrict__ Ai, const double* __restrict__ Ax, const int* __restrict__ Bp,
                                     int* __restrict__ Bi, double* __restrict__ Bx) {
    IPXK_GRID_STRIDE(k, m) {
        const ipxint j = basis[k];
        int q = Bp[k];
        if (j < n) {
            for (int p = Ap[j]; p < Ap[j + 1]; p++, q++) { Bi[q] = Ai[p]; Bx[q] = Ax[p]; }
        } else {
            Bi[q] = (int)(j - n);
            Bx[q] = 1.0;
        }
    }
}

void sort_keys(Tmp& T, u64* keys, u64* keys2, double* vals, double* vals2, size_t n, int end_bit, hipStream_t s) {
    size_t bytes = 0;
    IPXK_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys, keys2, vals, vals2, n, 0u, (unsigned)end_bit, s));
    IPXK_HIP(rocprim::radix_sort_pairs(T.need(bytes), bytes, keys, keys2, vals, vals2, n, 0u, (unsigned)end_bit, s));
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

// workspaces of a factorization, kept from one call to the next (grow-only)
struct LuWork {
    DevBuf<int> colof, keys, pos, keys2, Rpos, Rj, Rp, rstage, cstage, rc, cc, cand, flag, rank, claim, pivrow, counters;
    DevBuf<int> rloc, cloc, brow, bcol, brstep, bcstep, bstep, prow, pcol;
    DevBuf<double> ubuf;               // [kPanel][kb] the outer panel's rows of U, contiguous (MFMA trailing update)
    DevBuf<double> usub;               // [sub-panel pivot][kPanel] a sub-panel's rows of U in the rest of the outer panel
    DevBuf<double> coop_slots;         // cooperative outer panel: the workgroups' messages, [5][kCoopMaxG][kCoopSlot]
    DevBuf<unsigned> coop_bar;         // [1] abort flag
    DevBuf<unsigned long long> coop_xcc;   // placement check of the one-XCD form
    unsigned coop_epoch = 0;
    DevBuf<u64> cand_bits, claim_abs, skey, skey2, lkey, lkey2, ukey, ukey2;
    DevBuf<double> pivot, D, lval, lval2, uval, uval2;
    DevBuf<unsigned char> ckind;
    DevBuf<u64> tkey, tkey2;          // tearing: candidate keys
    DevBuf<ipxint> lrp, tagptr;       // tearing: row pointers of the L entries by stage; first stage of each half-round
    DevBuf<int> tagwork, spk_c, spk_s;
    DevBuf<double> X, spk_v;
    DevBuf<int> Bp, Bi, cnt;          // B = AI[:, basis] (ipxk_lu_factorize_basis)
    DevBuf<double> Bx;
    // elimination rounds: the current matrix (two copies, used alternately), its row-wise index, the round's scratch, and the
    // list E of the entries that have left the current matrix (indices in B, value at that time)
    struct Sp {
        DevBuf<int> Bp[2], Bi[2], colof[2], grow[2], gcol[2];
        DevBuf<double> Bx[2];
        DevBuf<int> Rp, Rj, Rpos, rc, cc, k32a, k32b, pos;
        DevBuf<int> candrow, cost, winner, nupd, uoff, rstL, cstL, flag, rank, newrow, newcol, listr, listc, stats, cflag, cpos, cposl, ecur;
        DevBuf<u64> key, rowbest, skey, skey2, ukey, ukey2, okey, okey2;
        DevBuf<double> pivl, sval, sval2, uval, uval2, csum;
        DevBuf<int> Erow, Ecol;
        DevBuf<double> Eval;
    } sp;
    Tmp T;
    int* h = nullptr;                 // pinned: counters read back per batch of rounds
    // look-ahead of the dense LU: the trailing update beyond the next outer panel runs on a second stream
    hipStream_t s2 = nullptr;
    hipEvent_t ev_rows[2] = {nullptr, nullptr}, ev_trail[2] = {nullptr, nullptr};
    ~LuWork() {
        if (h) (void)hipHostFree(h);
        for (hipEvent_t e : {ev_rows[0], ev_rows[1], ev_trail[0], ev_trail[1]}) if (e) (void)hipEventDestroy(e);
        if (s2) (void)hipStreamDestroy(s2);
    }
};

struct LuState {
    LuWork work;
    int dim = 0;
    int64_t lnz = 0, unz = 0;
    int ndep = 0;
    int bump_start = 0, bump_size = 0;      // pivot stages [bump_start, bump_start + bump_size) came from the dense bump
    bool valid = false, from_basis = false;
    DevBuf<ipxint> Lp, Li, Up, Ui, rowperm, colperm, dependent, basis;
    DevBuf<double> Lx, Ux;
    // plain CSC of the structural matrix (32-bit), uploaded at the first ipxk_lu_factorize_basis
    DevBuf<int> Ap, Ai;
    DevBuf<double> Ax;
    bool have_A = false;
    // what the resident factorization was computed with, and its statistics: ipxk_lu_factorize hands the factors of EXACTLY the
    // matrix it is given out again instead of computing them a second time (lu_reuse_resident)
    double pivottol_used = 0.0;
    bool strict_used = false;
    ipxk_lu_info last_info{};
    bool view = false;                  // ipxk_lu_get_factors returns colperm_view (the caller's column order) instead of colperm
    DevBuf<ipxint> colperm_view;
    DevBuf<int> reuse_cand, reuse_mark, reuse_sigma, reuse_flag;
    long generation = 0;                // counts the factorizations actually computed in this context
};

void destroy_lu(LuState* S) { delete S; }

static LuState* lu_state(Context* c) {
    if (!c->lu) c->lu = new LuState;
    return c->lu;
}

namespace {
__global__ void sp_stats_init_kernel(int* stats) {
    if (blockIdx.x == 0 && threadIdx.x < kSpStats) stats[threadIdx.x] = threadIdx.x == 0 ? INT_MAX : 0;
}
struct SparseOut {
    int kb = 0, cur = 0, pivots = 0, rounds = 0;
    int64_t nnz = 0, ne = 0;
};
// sorted (key, value) pairs -> the next current matrix in copy `dst`: equal keys summed in order, exact zeros dropped, column
// pointers, row-wise index, counts.  Reads back the number of entries (and the length of E).
void sp_finish_matrix(hipStream_t s, LuWork& W, const u64* skeys, const double* svals, int64_t n, int dimL, int dst, int* h, int64_t* nnz_out,
                      int64_t* ne_out) {
    LuWork::Sp& P = W.sp;
    Tmp& T = W.T;
    P.cflag.ensure((size_t)n + 1); P.cpos.ensure((size_t)n + 1); P.csum.ensure((size_t)std::max<int64_t>(n, 1));
    hipLaunchKernelGGL(sp_combine_kernel, dim3(grid_for(n + 1)), dim3(kBlock), 0, s, n, skeys, svals, P.cflag.get(), P.csum.get());
    scan_exclusive(T, P.cflag.get(), P.cpos.get(), (size_t)n + 1, s);
    IPXK_HIP(hipMemcpyAsync(h, P.cpos.get() + n, sizeof(int), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipMemcpyAsync(h + 1, P.ecur.get(), sizeof(int), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    const int64_t nnz = h[0];
    *nnz_out = nnz;
    *ne_out = h[1];
    const size_t z1 = (size_t)std::max<int64_t>(nnz, 1), d1 = (size_t)std::max(dimL, 1);
    P.Bi[dst].ensure(z1); P.colof[dst].ensure(z1); P.Bx[dst].ensure(z1); P.Bp[dst].ensure(d1 + 1);
    if (n > 0)
        hipLaunchKernelGGL(sp_compact_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, n, skeys, P.cflag.get(), P.cpos.get(), P.csum.get(),
                           P.Bi[dst].get(), P.colof[dst].get(), P.Bx[dst].get());
    hipLaunchKernelGGL(sp_colptr_kernel, dim3(grid_for(dimL + 1)), dim3(kBlock), 0, s, dimL, (int)nnz, P.colof[dst].get(), P.Bp[dst].get());
    // rows
    P.rc.ensure(d1 + 1); P.cc.ensure(d1); P.Rp.ensure(d1 + 1);
    for (DevBuf<int>* b : {&P.Rpos, &P.Rj}) b->ensure(z1);
    IPXK_HIP(hipMemsetAsync(P.rc.get(), 0, (d1 + 1) * sizeof(int), s));
    if (dimL > 0) hipLaunchKernelGGL(sp_colcount_kernel, dim3(grid_for(dimL)), dim3(kBlock), 0, s, dimL, P.Bp[dst].get(), P.cc.get());
    // the row-wise index: by atomics for small matrices, by a sort for large ones (a row with thousands of entries serializes
    // its atomics).  Nothing depends on the order of a row's entries.
    const bool rows_by_sort = nnz >= (int64_t)kSpRowsBySort;
    if (rows_by_sort) {
        for (DevBuf<int>* b : {&P.k32a, &P.k32b, &P.pos}) b->ensure(z1);
        hipLaunchKernelGGL(sp_rowkeys_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, nnz, P.Bi[dst].get(), P.k32a.get(), P.pos.get());
        size_t bytes = 0;
        const unsigned bits = (unsigned)bits_for(std::max(dimL, 2));
        IPXK_HIP(rocprim::radix_sort_pairs(nullptr, bytes, P.k32a.get(), P.k32b.get(), P.pos.get(), P.Rpos.get(), (size_t)nnz, 0u, bits, s));
        IPXK_HIP(rocprim::radix_sort_pairs(T.need(bytes), bytes, P.k32a.get(), P.k32b.get(), P.pos.get(), P.Rpos.get(), (size_t)nnz, 0u, bits, s));
        hipLaunchKernelGGL(lu_rows_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, nnz, P.Rpos.get(), P.colof[dst].get(), P.Rj.get());
        hipLaunchKernelGGL(sp_colptr_kernel, dim3(grid_for(dimL + 1)), dim3(kBlock), 0, s, dimL, (int)nnz, P.k32b.get(), P.Rp.get());
        hipLaunchKernelGGL(sp_colcount_kernel, dim3(grid_for(dimL)), dim3(kBlock), 0, s, dimL, P.Rp.get(), P.rc.get());
        return;
    }
    if (nnz > 0) hipLaunchKernelGGL(sp_rowcount_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, nnz, P.Bi[dst].get(), P.rc.get());
    scan_exclusive(T, P.rc.get(), P.Rp.get(), d1 + 1, s);
    if (nnz > 0) {
        P.k32a.ensure(d1);
        IPXK_HIP(hipMemsetAsync(P.k32a.get(), 0, d1 * sizeof(int), s));
        hipLaunchKernelGGL(sp_rowfill_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, nnz, P.Bi[dst].get(), P.colof[dst].get(), P.Rp.get(), P.k32a.get(),
                           P.Rj.get(), P.Rpos.get());
    }
}

// 2c. ELIMINATION ROUNDS.  The singleton rounds have stalled with `nact` active rows and columns of B (rstage / cstage < 0).
// Until at most sparse_min are left (or no column has an acceptable pivot), a round picks pivots of low Markowitz cost that
// form a diagonal block, eliminates them at once and builds the next current matrix (see the kernels).  On return the current
// matrix (copy out.cur: out.kb rows, out.nnz entries, local indices = rank among the rows / columns of B that are still
// active) is what the dense code takes over, and W.sp.E* (out.ne entries) replaces B in the assembly.
SparseOut sparse_rounds(hipStream_t s, LuWork& W, int dim, int64_t nb, const int* Bi, const int* colof, const double* Bx, SparseGlobal G,
                        int nact, int sparse_min, int kb_max, int slow_den, int fill_max, double dense_at, bool fill_to_dense, int* rounds, double abstol,
                        double pivottol, int* h) {
    LuWork::Sp& P = W.sp;
    Tmp& T = W.T;
    SparseOut out;
    const int g = grid_for(dim);
    const size_t d1 = (size_t)std::max(dim, 1);
    const bool verbose = getenv("IPXK_VERBOSE") && atoi(getenv("IPXK_VERBOSE")) >= 2;
    P.stats.ensure(kSpStats); P.ecur.ensure(1);
    IPXK_HIP(hipMemsetAsync(P.ecur.get(), 0, sizeof(int), s));
    IPXK_HIP(hipMemsetAsync(P.stats.get(), 0, kSpStats * sizeof(int), s));
    // the active submatrix of B, renumbered; everything else of B is finished
    W.rloc.ensure(d1); W.cloc.ensure(d1); W.flag.ensure(d1); W.rank.ensure(d1);
    int dimL = nact, cur = 0;
    P.grow[0].ensure((size_t)std::max(dimL, 1)); P.gcol[0].ensure((size_t)std::max(dimL, 1));
    hipLaunchKernelGGL(lu_active_flag_kernel, dim3(g), dim3(kBlock), 0, s, dim, G.rstage, W.flag.get());
    scan_exclusive(T, W.flag.get(), W.rank.get(), (size_t)dim, s);
    hipLaunchKernelGGL(lu_compact_kernel, dim3(g), dim3(kBlock), 0, s, dim, W.flag.get(), W.rank.get(), W.rloc.get(), P.grow[0].get());
    hipLaunchKernelGGL(lu_active_flag_kernel, dim3(g), dim3(kBlock), 0, s, dim, G.cstage, W.flag.get());
    scan_exclusive(T, W.flag.get(), W.rank.get(), (size_t)dim, s);
    hipLaunchKernelGGL(lu_compact_kernel, dim3(g), dim3(kBlock), 0, s, dim, W.flag.get(), W.rank.get(), W.cloc.get(), P.gcol[0].get());
    const size_t nz1 = (size_t)std::max<int64_t>(nb, 1);
    for (DevBuf<u64>* b : {&P.skey, &P.skey2}) b->ensure(nz1);
    for (DevBuf<double>* b : {&P.sval, &P.sval2}) b->ensure(nz1);
    P.Erow.ensure(nz1); P.Ecol.ensure(nz1); P.Eval.ensure(nz1);
    hipLaunchKernelGGL(sp_entries_kernel, dim3(grid_for(nb)), dim3(kBlock), 0, s, nb, Bi, colof, Bx, W.rloc.get(), W.cloc.get(), (const int*)nullptr,
                       (const int*)nullptr, P.skey.get(), P.sval.get(), (int*)nullptr, P.ecur.get(), P.Erow.get(), P.Ecol.get(), P.Eval.get());
    int64_t nnz = 0, ne = 0;
    if (nb > 0) sort_keys(T, P.skey.get(), P.skey2.get(), P.sval.get(), P.sval2.get(), (size_t)nb, 32 + bits_for(std::max(dimL, 2)), s);
    sp_finish_matrix(s, W, P.skey2.get(), P.sval2.get(), nb, dimL, cur, h, &nnz, &ne);
    int slow = 0;
    while (dimL > sparse_min) {
        // (the rounds stop early once the current matrix fits the dense code and two rounds in a row have each eliminated fewer
        // than 1 / slow_den of the columns: what is left has no large sets of independent pivots any more)
        if (slow_den > 0 && dimL <= kb_max && slow >= 2) break;
        // ... or once it fits the dense code and holds more than dense_at x dimL^2 entries: a dense matrix in sparse storage
        if (dense_at > 0.0 && dimL <= kb_max && (double)nnz > dense_at * (double)dimL * (double)dimL) break;
        const size_t l1 = (size_t)dimL;
        for (DevBuf<int>* b : {&P.candrow, &P.cost, &P.winner, &P.rstL, &P.cstL, &P.flag, &P.rank, &P.newrow, &P.newcol, &P.listr, &P.listc}) b->ensure(l1);
        P.nupd.ensure(l1 + 1); P.uoff.ensure(l1 + 1); P.key.ensure(l1); P.rowbest.ensure(l1); P.pivl.ensure(l1);
        Sparse S{dimL, P.Bp[cur].get(), P.Bi[cur].get(), P.colof[cur].get(), P.Bx[cur].get(), P.Rp.get(), P.Rj.get(), P.Rpos.get(), P.rc.get(),
                 P.cc.get(), P.candrow.get(), P.cost.get(), P.key.get(), P.rowbest.get(), P.stats.get(), abstol, pivottol};
        const int gl = grid_for(dimL);
        hipLaunchKernelGGL(sp_stats_init_kernel, dim3(1), dim3(64), 0, s, P.stats.get());
        hipLaunchKernelGGL(lu_fill_u64_kernel, dim3(gl), dim3(kBlock), 0, s, (int64_t)dimL, kNoKey, P.rowbest.get());
        hipLaunchKernelGGL(lu_fill_int_kernel, dim3(gl), dim3(kBlock), 0, s, (int64_t)dimL, -1, P.rstL.get());
        hipLaunchKernelGGL(lu_fill_int_kernel, dim3(gl), dim3(kBlock), 0, s, (int64_t)dimL, -1, P.cstL.get());
        IPXK_HIP(hipMemsetAsync(P.nupd.get(), 0, (l1 + 1) * sizeof(int), s));
        const int gw = (int)std::min<int64_t>(4096, ((int64_t)dimL + kBlock / 64 - 1) / (kBlock / 64));     // a wavefront per column
        hipLaunchKernelGGL(sp_cand_kernel, dim3(gw), dim3(kBlock), 0, s, S);
        hipLaunchKernelGGL(sp_limit_kernel, dim3(1), dim3(64), 0, s, P.stats.get());
        hipLaunchKernelGGL(sp_key_kernel, dim3(gl), dim3(kBlock), 0, s, S);
        hipLaunchKernelGGL(sp_win_kernel, dim3(gw), dim3(kBlock), 0, s, S, P.winner.get(), P.nupd.get(), P.stats.get() + 40);
        scan_exclusive(T, P.nupd.get(), P.uoff.get(), l1 + 1, s);
        IPXK_HIP(hipMemcpyAsync(h, P.stats.get(), kSpStats * sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipMemcpyAsync(h + kSpStats, P.uoff.get() + dimL, sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        const int nwin = h[3];
        int64_t nupd = h[kSpStats];
        const int64_t nupd64 = (int64_t)(((unsigned long long)(unsigned)h[43] << 32) | (unsigned long long)(unsigned)h[42]);
        if (h[40] || nupd64 + nnz >= (int64_t(1) << 31)) {
            // more updates than 32-bit positions hold (a round of a matrix that has become dense): the dense code takes what is left if
            // it can (round 5's policy), else the rounds are given up
            if (fill_to_dense && dimL <= kb_max) break;
            throw Error(IPXK_E_UNSUPPORTED, "LU: an elimination round would send out more updates than 32-bit positions hold");
        }
        nupd = nupd64;
        if (verbose)
            fprintf(stderr, "ipxk: elimination round %d: active %d nnz %lld cheapest %d limit %d candidates %d winners %d updates %lld\n", out.rounds + 1, dimL,
                    (long long)nnz, h[0], h[2], h[1], nwin, (long long)nupd);
        if (nwin == 0) break;                      // no column has an acceptable pivot: what is left goes to the dense block
        const int tag = 2 * (*rounds);
        (*rounds)++;
        out.rounds++;
        out.pivots += nwin;
        hipLaunchKernelGGL(sp_commit_kernel, dim3(gl), dim3(kBlock), 0, s, S, P.winner.get(), P.grow[cur].get(), P.gcol[cur].get(), tag, G, P.rstL.get(),
                           P.cstL.get(), P.pivl.get());
        // the rows and columns that stay, renumbered in order
        const int dimN = dimL - nwin, nxt = 1 - cur;
        P.grow[nxt].ensure((size_t)std::max(dimN, 1)); P.gcol[nxt].ensure((size_t)std::max(dimN, 1));
        hipLaunchKernelGGL(lu_active_flag_kernel, dim3(gl), dim3(kBlock), 0, s, dimL, P.rstL.get(), P.flag.get());
        scan_exclusive(T, P.flag.get(), P.rank.get(), l1, s);
        hipLaunchKernelGGL(lu_compact_kernel, dim3(gl), dim3(kBlock), 0, s, dimL, P.flag.get(), P.rank.get(), P.newrow.get(), P.listr.get());
        hipLaunchKernelGGL(lu_active_flag_kernel, dim3(gl), dim3(kBlock), 0, s, dimL, P.cstL.get(), P.flag.get());
        scan_exclusive(T, P.flag.get(), P.rank.get(), l1, s);
        hipLaunchKernelGGL(lu_compact_kernel, dim3(gl), dim3(kBlock), 0, s, dimL, P.flag.get(), P.rank.get(), P.newcol.get(), P.listc.get());
        if (dimN > 0) {
            hipLaunchKernelGGL(sp_map_kernel, dim3(grid_for(dimN)), dim3(kBlock), 0, s, dimN, P.listr.get(), P.grow[cur].get(), P.grow[nxt].get());
            hipLaunchKernelGGL(sp_map_kernel, dim3(grid_for(dimN)), dim3(kBlock), 0, s, dimN, P.listc.get(), P.gcol[cur].get(), P.gcol[nxt].get());
        }
        // the entries that stay keep their order (the renumbering is monotone): compacted, then merged with the sorted updates
        // (the merge is stable: at equal positions the entry comes first, then the updates in the order of the winners)
        const int64_t ncarry = nnz - h[4], n = ncarry + nupd;
        IPXK_REQUIRE(ncarry >= 0 && n < (int64_t(1) << 31), "LU: the current matrix of an elimination round exceeds 32-bit positions");
        const size_t nz1 = (size_t)std::max<int64_t>(std::max(n, nnz), 1), nu1 = (size_t)std::max<int64_t>(nupd, 1);
        for (DevBuf<u64>* b : {&P.skey, &P.skey2}) b->ensure(nz1);
        for (DevBuf<double>* b : {&P.sval, &P.sval2}) b->ensure(nz1);
        for (DevBuf<u64>* b : {&P.ukey, &P.ukey2}) b->ensure(nu1);
        for (DevBuf<double>* b : {&P.uval, &P.uval2}) b->ensure(nu1);
        P.cflag.ensure((size_t)nnz + 1); P.cpos.ensure((size_t)nnz + 1);
        grow_keep(P.Erow, (size_t)ne, (size_t)(ne + nnz), s);
        grow_keep(P.Ecol, (size_t)ne, (size_t)(ne + nnz), s);
        grow_keep(P.Eval, (size_t)ne, (size_t)(ne + nnz), s);
        if (nnz > 0) {
            hipLaunchKernelGGL(sp_entries_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, nnz, P.Bi[cur].get(), P.colof[cur].get(), P.Bx[cur].get(),
                               P.newrow.get(), P.newcol.get(), P.grow[cur].get(), P.gcol[cur].get(), P.skey.get(), P.sval.get(), P.cflag.get(),
                               P.ecur.get(), P.Erow.get(), P.Ecol.get(), P.Eval.get());
            scan_exclusive(T, P.cflag.get(), P.cpos.get(), (size_t)nnz, s);
            hipLaunchKernelGGL(sp_carry_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, nnz, P.cflag.get(), P.cpos.get(), P.skey.get(), P.sval.get(),
                               P.skey2.get(), P.sval2.get());
        }
        const u64* mkeys = P.skey2.get();
        const double* mvals = P.sval2.get();
        if (nupd > 0) {
            hipLaunchKernelGGL(sp_updates_kernel, dim3(grid_for(nupd)), dim3(kBlock), 0, s, S, P.uoff.get(), nupd, P.pivl.get(), P.newrow.get(),
                               P.newcol.get(), (int64_t)0, P.ukey.get(), P.uval.get());
            sort_keys(T, P.ukey.get(), P.ukey2.get(), P.uval.get(), P.uval2.get(), (size_t)nupd, 32 + bits_for(std::max(dimN, 2)), s);
            size_t bytes = 0;
            IPXK_HIP(rocprim::merge(nullptr, bytes, P.skey2.get(), P.ukey2.get(), P.skey.get(), P.sval2.get(), P.uval2.get(), P.sval.get(), (size_t)ncarry,
                                    (size_t)nupd, rocprim::less<u64>(), s));
            IPXK_HIP(rocprim::merge(T.need(bytes), bytes, P.skey2.get(), P.ukey2.get(), P.skey.get(), P.sval2.get(), P.uval2.get(), P.sval.get(),
                                    (size_t)ncarry, (size_t)nupd, rocprim::less<u64>(), s));
            mkeys = P.skey.get();
            mvals = P.sval.get();
        }
        sp_finish_matrix(s, W, mkeys, mvals, n, dimN, nxt, h, &nnz, &ne);
        slow = (int64_t)nwin * slow_den < (int64_t)dimL ? slow + 1 : 0;
        cur = nxt;
        dimL = dimN;
        // bounded work: a bump whose elimination fills in beyond fill_max x nnz(B) (+ 2^20) is refused, and the caller's CPU
        // kernel takes over (measured: such bases take minutes here -- the 1M-row basis of scripts/gpu_maxvol_bench.py with a
        // tightened pivot tolerance: 14 941 rounds, 291 s, 103 x fill)
        // (round 5's policy: room for a current matrix the density rule will end -- dense_at x kb_max^2 entries -- whatever nnz(B) is)
        const int64_t fill_bound = std::max<int64_t>((int64_t)fill_max * nb + (1 << 20), fill_to_dense ? (int64_t)(dense_at * (double)kb_max * (double)kb_max) : 0);
        if (fill_max > 0 && nnz > fill_bound) {
            if (fill_to_dense && dimL <= kb_max) break;             // (round 5: the dense code takes what is left if it can)
            char msg[200];
            snprintf(msg, sizeof msg, "LU: after %d elimination rounds the bump holds %lld entries in %d columns, beyond the bound of the rounds "
                     "(IPXK_LU_SPARSE_FILL_MAX)", out.rounds, (long long)nnz, dimL);
            throw Error(IPXK_E_UNSUPPORTED, msg);
        }
    }
    out.kb = dimL; out.cur = cur; out.nnz = nnz; out.ne = ne;
    return out;
}
}  // namespace

// B as compact 32-bit CSC on the device -> factors in S
static void lu_factorize_device(Context* c, LuState* S, int dim, int64_t nb_in, const int* Bp, const int* Bi,
                                const double* Bx, double pivottol, bool strict, ipxk_lu_info* info, int after_failed_tear = 0) {
    hipStream_t s = c->stream;
    int64_t nb = nb_in;
    S->valid = false;
    S->view = false;
    S->dim = dim;
    ipxk_lu_info I{};
    const double abstol = strict ? 1e-3 : 1e-14;      // kLuDependencyTol (src/ipx_internal.h:26) / BASICLU's default
    if (const char* dir = getenv("IPXK_LU_DUMP")) {   // study aid: the bases of a run as flat files (dim, nnz, Bp, Bi, Bx), every IPXK_LU_DUMP_EVERY-th
        static int calls = 0, written = 0;
        const int every = getenv("IPXK_LU_DUMP_EVERY") ? std::max(1, atoi(getenv("IPXK_LU_DUMP_EVERY"))) : 1;
        if (!after_failed_tear && nb > dim && calls++ % every == 0 && written < 64) {
            std::vector<int> hp((size_t)dim + 1), hi((size_t)nb);
            std::vector<double> hx((size_t)nb);
            IPXK_HIP(hipMemcpy(hp.data(), Bp, hp.size() * sizeof(int), hipMemcpyDeviceToHost));
            IPXK_HIP(hipMemcpy(hi.data(), Bi, hi.size() * sizeof(int), hipMemcpyDeviceToHost));
            IPXK_HIP(hipMemcpy(hx.data(), Bx, hx.size() * sizeof(double), hipMemcpyDeviceToHost));
            char path[512];
            snprintf(path, sizeof path, "%s/basis_%03d.bin", dir, written++);
            if (FILE* f = fopen(path, "wb")) {
                const int64_t head[2] = {dim, nb};
                fwrite(head, sizeof(int64_t), 2, f);
                fwrite(hp.data(), sizeof(int), hp.size(), f); fwrite(hi.data(), sizeof(int), hi.size(), f); fwrite(hx.data(), sizeof(double), hx.size(), f);
                fclose(f);
            }
        }
    }
    const double t0 = now_s();
    LuWork& W = S->work;
    Tmp& T = W.T;
    const size_t d1 = (size_t)std::max(dim, 1), nz1 = (size_t)std::max<int64_t>(nb, 1);
    DevBuf<int> &colof = W.colof, &keys = W.keys, &pos = W.pos, &keys2 = W.keys2, &Rpos = W.Rpos, &Rj = W.Rj, &Rp = W.Rp;
    DevBuf<int> &rstage = W.rstage, &cstage = W.cstage, &rc = W.rc, &cc = W.cc, &cand = W.cand, &flag = W.flag, &rank = W.rank,
                &claim = W.claim, &pivrow = W.pivrow, &counters = W.counters;
    DevBuf<u64> &cand_bits = W.cand_bits, &claim_abs = W.claim_abs;
    DevBuf<double>& pivot = W.pivot;
    DevBuf<unsigned char>& ckind = W.ckind;
    for (DevBuf<int>* b : {&colof, &keys, &pos, &keys2, &Rpos, &Rj}) b->ensure(nz1);
    for (DevBuf<int>* b : {&rstage, &cstage, &rc, &cc, &cand, &flag, &rank, &claim, &pivrow}) b->ensure(d1);
    Rp.ensure(d1 + 1); counters.ensure(32);
    cand_bits.ensure(d1); claim_abs.ensure(d1); pivot.ensure(d1); ckind.ensure(d1);
    if (!W.h) IPXK_HIP(hipHostMalloc(reinterpret_cast<void**>(&W.h), 64 * sizeof(int)));
    int* h = W.h;
    IPXK_HIP(hipMemsetAsync(rc.get(), 0, d1 * sizeof(int), s));
    IPXK_HIP(hipMemsetAsync(counters.get(), 0, 32 * sizeof(int), s));
    IPXK_HIP(hipMemsetAsync(claim_abs.get(), 0, d1 * sizeof(u64), s));
    IPXK_HIP(hipMemsetAsync(ckind.get(), 0, d1, s));
    IPXK_HIP(hipMemsetAsync(pivot.get(), 0, d1 * sizeof(double), s));
    const int g = grid_for(dim);
    if (dim > 0) {
        hipLaunchKernelGGL(lu_fill_int_kernel, dim3(g), dim3(kBlock), 0, s, (int64_t)dim, -1, rstage.get());
        hipLaunchKernelGGL(lu_fill_int_kernel, dim3(g), dim3(kBlock), 0, s, (int64_t)dim, -1, cstage.get());
        hipLaunchKernelGGL(lu_fill_int_kernel, dim3(g), dim3(kBlock), 0, s, (int64_t)dim, INT_MAX, claim.get());
        hipLaunchKernelGGL(lu_expand_kernel, dim3(g), dim3(kBlock), 0, s, dim, Bp, Bi, colof.get(), keys.get(), pos.get(),
                           rc.get(), cc.get(), counters.get() + 7);
        if (nb > 0) {
            size_t bytes = 0;
            IPXK_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys.get(), keys2.get(), pos.get(), Rpos.get(), (size_t)nb, 0u,
                                               (unsigned)bits_for(std::max(dim, 2)), s));
            IPXK_HIP(rocprim::radix_sort_pairs(T.need(bytes), bytes, keys.get(), keys2.get(), pos.get(), Rpos.get(), (size_t)nb,
                                               0u, (unsigned)bits_for(std::max(dim, 2)), s));
            hipLaunchKernelGGL(lu_rows_kernel, dim3(grid_for(nb)), dim3(kBlock), 0, s, nb, Rpos.get(), colof.get(), Rj.get());
        }
        scan_exclusive(T, rc.get(), Rp.get(), (size_t)dim, s);
        const int nb32 = (int)nb;
        IPXK_HIP(hipMemcpyAsync(Rp.get() + dim, &nb32, sizeof(int), hipMemcpyHostToDevice, s));
        IPXK_HIP(hipStreamSynchronize(s));             // nb32 is a stack variable
    }
    // ---- 1. singleton rounds
    Rounds R{dim, Bp, Bi, Bx, Rp.get(), Rj.get(), Rpos.get(), rstage.get(), cstage.get(), rc.get(), cc.get(), pivot.get(),
             ckind.get(), cand.get(), claim.get(), pivrow.get(), cand_bits.get(), claim_abs.get(), counters.get(),
             abstol, pivottol};
    int rounds = 0;
    const int batch = 8;
    int kb_max = 8192;      // (4096 until round 3: the IPM's bases on random 12 000-row LPs end in bumps of 8000 rows)
    if (const char* e = getenv("IPXK_LU_BUMP_MAX")) kb_max = std::max(0, atoi(e));
    bool tearing = false;
    int ntorn = 0, tear_width = 1, npiv_at_tear = 0;
    // WHICH way a bump goes (round 5).  A bump of at most sparse_from rows (1024) is factorized densely as it stands.  A larger one
    // is eliminated SPARSELY in rounds (2c) as long as that pays, and only the rest -- a matrix that has become dense, typically a
    // third to a half of an IPM basis' bump -- goes to the dense code: on the bases of the IPM (random LPs, 16 000 rows, bump of
    // 10 700) that is nnz(L+U) 24.6 M and 60 ms where the dense bump as it stands gave 95 M and 330 ms; the sequential minimum-
    // Markowitz elimination of the same bases ends in 22.2 M, so there is no better order to be had.  The
    // rounds end at sparse_min columns; or, once at most rest_max columns are left (what the dense code takes), when the current
    // matrix holds more than dense_at x columns^2 entries, after two slow rounds, or when it has grown beyond fill_max x nnz(B) + 2^20
    // entries.  A bump of more than sparse_first_max rows (131 072: the chains of a 1M-row basis, where a round costs 1 ms and
    // frees a handful of pivots) is TORN first (2b) as until round 4, with the rounds as the fall-back; rounds that give up with
    // more than rest_max columns left start again with tearing.  Only a basis neither way can take is refused.
    //   IPXK_LU_SPARSE=t: tearing first for every bump beyond IPXK_LU_BUMP_MAX, rounds as the fall-back (the policy of round 4);
    //   =1: rounds instead of tearing for those bumps, round 4's end rules; =0: tearing only.
    const char* sparse_env = getenv("IPXK_LU_SPARSE");
    const bool sparse_allowed = !(sparse_env && sparse_env[0] == '0');
    const bool legacy_rounds = sparse_env && sparse_env[0] == '1';
    const bool legacy = sparse_env && (sparse_env[0] == '0' || sparse_env[0] == '1' || sparse_env[0] == 't');
    // The limit decides WHETHER a bump is torn; the spikes themselves may fill the largest dense block the panel kernels take
    // (16 rows per thread: 16384 rows) before tearing gives up -- on the IPM bases of random LPs of 12 000 ... 24 000 rows tearing
    // ends with 8000 ... 11 000 spikes.  (A small limit set for tests binds the spikes too.)
    const int spike_max = kb_max > 4 * kPanelThreads ? std::max(kb_max, kDenseHardMax) : kb_max;
    const int rest_max = legacy ? kb_max : spike_max;
    int sparse_min = 512;
    if (const char* e = getenv("IPXK_LU_SPARSE_MIN")) sparse_min = std::max(0, atoi(e));
    int slow_den = legacy ? 256 : 2048;   // ... or it fits the dense code and two rounds in a row each eliminate fewer than 1 / 256 (2048) of the columns
    if (const char* e = getenv("IPXK_LU_SPARSE_SLOW_DEN")) slow_den = std::max(0, atoi(e));
    int fill_max = 8;               // ... or the bump has filled in beyond 8 x nnz(B) + 2^20 entries
    if (const char* e = getenv("IPXK_LU_SPARSE_FILL_MAX")) fill_max = std::max(0, atoi(e));
    int sparse_from = legacy ? kb_max : std::min(kb_max, 1024);
    if (const char* e = getenv("IPXK_LU_SPARSE_FROM")) sparse_from = std::max(0, atoi(e));
    int sparse_first_max = 131072;
    if (const char* e = getenv("IPXK_LU_SPARSE_FIRST_MAX")) sparse_first_max = std::max(0, atoi(e));
    double dense_at = legacy ? 0.0 : 0.2;
    if (const char* e = getenv("IPXK_LU_SPARSE_DENSITY")) dense_at = atof(e);
    SparseOut sp;
    bool sparse_done = false;
    while (dim > 0) {
        IPXK_HIP(hipMemsetAsync(counters.get() + 8, 0, batch * sizeof(int), s));
        for (int b = 0; b < batch; b++) {
            const int tag = 2 * (rounds + b);
            hipLaunchKernelGGL(lu_col_find_kernel, dim3(g), dim3(kBlock), 0, s, R);
            hipLaunchKernelGGL(lu_col_commit_kernel, dim3(g), dim3(kBlock), 0, s, R, tag, 8 + b);
            hipLaunchKernelGGL(lu_row_find_kernel, dim3(g), dim3(kBlock), 0, s, R);
            hipLaunchKernelGGL(lu_row_pick_kernel, dim3(g), dim3(kBlock), 0, s, R);
            hipLaunchKernelGGL(lu_row_commit_kernel, dim3(g), dim3(kBlock), 0, s, R, tag + 1, 8 + b);
        }
        IPXK_HIP(hipMemcpyAsync(h, counters.get(), 16 * sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        if (h[7]) throw Error(IPXK_E_ARGUMENT, "row index of B out of range");
        int last_busy = -1;
        for (int b = 0; b < batch; b++) if (h[8 + b] > 0) last_busy = b;
        rounds += last_busy + 1 < batch ? last_busy + 2 : batch;      // the iteration that found nothing counts
        if (last_busy == batch - 1) continue;
        // the rounds stall: done, or (a bump beyond the dense limit) tear spikes off and go on
        IPXK_HIP(hipMemsetAsync(counters.get() + 1, 0, 2 * sizeof(int), s));
        hipLaunchKernelGGL(lu_count_kinds_kernel, dim3(g), dim3(kBlock), 0, s, dim, ckind.get(), counters.get());
        IPXK_HIP(hipMemcpyAsync(h, counters.get(), 8 * sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        const int npiv = h[1] + h[2], nact = dim - npiv - ntorn;
        if (nact == 0) break;

        const bool rounds_now = !tearing && sparse_allowed && nact > sparse_from &&
                                (after_failed_tear == 1 || legacy_rounds || (!legacy && after_failed_tear != 2 && nact <= sparse_first_max));
        if (rounds_now) {                                               // 2c. elimination rounds down to sparse_min rows
            SparseGlobal G{rstage.get(), cstage.get(), pivrow.get(), pivot.get(), ckind.get()};
            try {
                sp = sparse_rounds(s, W, dim, nb, Bi, colof.get(), Bx, G, nact, std::min(sparse_min, kb_max), rest_max, slow_den, fill_max, dense_at, !legacy,
                                   &rounds, abstol, pivottol, h);
            } catch (const Error& e) {
                if (legacy || after_failed_tear != 0 || e.code != IPXK_E_UNSUPPORTED) throw;
                if (getenv("IPXK_VERBOSE")) fprintf(stderr, "ipxk: LU dim %d: %s: starting again with tearing\n", dim, e.what());
                lu_factorize_device(c, S, dim, nb_in, Bp, Bi, Bx, pivottol, strict, info, 2);
                return;
            }
            sparse_done = true;
            break;
        }
        if (!tearing) {
            if (nact <= (legacy ? kb_max : std::max(kb_max, sparse_from))) break;      // small enough: dense as it stands
            tearing = true;
        } else {
            tear_width = npiv - npiv_at_tear < 64 ? std::min(2 * tear_width, 1024) : 1;
        }
        const int take = std::min(tear_width, nact);
        W.tkey.ensure(d1); W.tkey2.ensure(d1);
        hipLaunchKernelGGL(lu_tear_keys_kernel, dim3(g), dim3(kBlock), 0, s, dim, cstage.get(), cc.get(), W.tkey.get());
        {
            size_t bytes = 0;
            IPXK_HIP(rocprim::radix_sort_keys(nullptr, bytes, W.tkey.get(), W.tkey2.get(), (size_t)dim, 0u, 64u, s));
            IPXK_HIP(rocprim::radix_sort_keys(T.need(bytes), bytes, W.tkey.get(), W.tkey2.get(), (size_t)dim, 0u, 64u, s));
        }
        hipLaunchKernelGGL(lu_tear_apply_kernel, dim3(grid_for(take)), dim3(kBlock), 0, s, R, W.tkey2.get(), take);
        ntorn += take;
        npiv_at_tear = npiv;
        if (ntorn > spike_max && sparse_allowed && after_failed_tear != 2) {
            if (getenv("IPXK_VERBOSE"))
                fprintf(stderr, "ipxk: LU dim %d: %d spikes torn off and %d columns still active: starting again with elimination rounds\n", dim, ntorn,
                        nact - take);
            lu_factorize_device(c, S, dim, nb_in, Bp, Bi, Bx, pivottol, strict, info, 1);
            return;
        }
        if (ntorn > spike_max) {
            char msg[200];
            snprintf(msg, sizeof msg, "LU: %d spikes torn off the bump and %d columns still active: the dense block would exceed %d rows "
                     "(IPXK_LU_BUMP_MAX)", ntorn, nact - take, spike_max);
            throw Error(IPXK_E_UNSUPPORTED, msg);
        }
    }
    if (dim > 0) {
        IPXK_HIP(hipMemsetAsync(counters.get() + 1, 0, 2 * sizeof(int), s));
        hipLaunchKernelGGL(lu_count_kinds_kernel, dim3(g), dim3(kBlock), 0, s, dim, ckind.get(), counters.get());
        IPXK_HIP(hipMemcpyAsync(h, counters.get(), 8 * sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
    }
    I.col_singletons = dim > 0 ? h[1] : 0;
    I.row_singletons = dim > 0 ? h[2] : 0;
    I.sparse_pivots = sp.pivots;
    I.sparse_rounds = sp.rounds;
    const int npiv_sing = (int)(I.col_singletons + I.row_singletons) + sp.pivots;       // all pivots before the dense block
    I.rounds = rounds;
    if (npiv_sing > 0) {           // dense stages: rounds in order, inside a round by index
        DevBuf<u64>& k1 = W.skey; DevBuf<u64>& k2 = W.skey2;
        k1.ensure(d1); k2.ensure(d1);
        hipLaunchKernelGGL(lu_stage_keys_kernel, dim3(g), dim3(kBlock), 0, s, dim, cstage.get(), pivrow.get(), k1.get(), cand.get());
        size_t bytes = 0;
        IPXK_HIP(rocprim::radix_sort_pairs(nullptr, bytes, k1.get(), k2.get(), cand.get(), claim.get(), (size_t)dim, 0u, 64u, s));
        IPXK_HIP(rocprim::radix_sort_pairs(T.need(bytes), bytes, k1.get(), k2.get(), cand.get(), claim.get(), (size_t)dim, 0u, 64u, s));
        hipLaunchKernelGGL(lu_stage_assign_kernel, dim3(grid_for(npiv_sing)), dim3(kBlock), 0, s, npiv_sing, claim.get(), pivrow.get(),
                           cstage.get(), rstage.get());
    }
    const double t1 = now_s();
    // ---- 2. bump
    DevBuf<int> &rloc = W.rloc, &cloc = W.cloc, &brow = W.brow, &bcol = W.bcol, &brstep = W.brstep, &bcstep = W.bcstep,
                &bstep = W.bstep, &prow = W.prow, &pcol = W.pcol;
    DevBuf<double>& D = W.D;
    rloc.ensure(d1); cloc.ensure(d1); bstep.ensure(8); prow.ensure(2 * kPanel); pcol.ensure(2 * kPanel);
    int kb = 0;
    if (dim > 0) {
        hipLaunchKernelGGL(lu_active_flag_kernel, dim3(g), dim3(kBlock), 0, s, dim, rstage.get(), flag.get());
        scan_exclusive(T, flag.get(), rank.get(), (size_t)dim, s);
        kb = dim - npiv_sing;
        brow.ensure((size_t)std::max(kb, 1)); bcol.ensure((size_t)std::max(kb, 1));
        hipLaunchKernelGGL(lu_compact_kernel, dim3(g), dim3(kBlock), 0, s, dim, flag.get(), rank.get(), rloc.get(), brow.get());
        hipLaunchKernelGGL(lu_active_flag_kernel, dim3(g), dim3(kBlock), 0, s, dim, cstage.get(), flag.get());
        scan_exclusive(T, flag.get(), rank.get(), (size_t)dim, s);
        hipLaunchKernelGGL(lu_compact_kernel, dim3(g), dim3(kBlock), 0, s, dim, flag.get(), rank.get(), cloc.get(), bcol.get());
    }
    I.bump = kb;
    I.spikes = tearing ? ntorn : 0;
    if (kb > (tearing ? spike_max : sparse_done ? rest_max : kb_max)) {
        char msg[160];
        snprintf(msg, sizeof msg, "LU: after the singletons a bump of %d rows remains (limit %d, IPXK_LU_BUMP_MAX)", kb, kb_max);
        throw Error(IPXK_E_UNSUPPORTED, msg);
    }
    int bpiv = 0;
    int64_t nspk = 0;                  // tearing: entries of the spikes in pivoted rows (future entries of U)
    brstep.ensure((size_t)std::max(kb, 1)); bcstep.ensure((size_t)std::max(kb, 1));
    if (kb > 0) {
        D.ensure((size_t)kb * kb);
        IPXK_HIP(hipMemsetAsync(D.get(), 0, (size_t)kb * kb * sizeof(double), s));
        IPXK_HIP(hipMemsetAsync(bstep.get(), 0, 8 * sizeof(int), s));
        const int gk = grid_for(kb);
        hipLaunchKernelGGL(lu_fill_int_kernel, dim3(gk), dim3(kBlock), 0, s, (int64_t)kb, -1, brstep.get());
        hipLaunchKernelGGL(lu_fill_int_kernel, dim3(gk), dim3(kBlock), 0, s, (int64_t)kb, -1, bcstep.get());
        if (sparse_done) {
            // the block's columns in ascending order of their number of entries (ties: index): fewer nonzeros in its factors
            LuWork::Sp& P = W.sp;
            IPXK_REQUIRE(sp.kb == kb, "LU: the elimination rounds and the stages disagree about what is left");
            P.okey.ensure((size_t)kb); P.okey2.ensure((size_t)kb); P.cposl.ensure((size_t)kb);
            hipLaunchKernelGGL(sp_colorder_keys_kernel, dim3(gk), dim3(kBlock), 0, s, kb, P.cc.get(), P.okey.get());
            size_t bytes = 0;
            IPXK_HIP(rocprim::radix_sort_keys(nullptr, bytes, P.okey.get(), P.okey2.get(), (size_t)kb, 0u, 64u, s));
            IPXK_HIP(rocprim::radix_sort_keys(T.need(bytes), bytes, P.okey.get(), P.okey2.get(), (size_t)kb, 0u, 64u, s));
            hipLaunchKernelGGL(sp_colorder_apply_kernel, dim3(gk), dim3(kBlock), 0, s, kb, P.okey2.get(), P.gcol[sp.cur].get(), bcol.get(), cloc.get(),
                               P.cposl.get());
            if (sp.nnz > 0)
                hipLaunchKernelGGL(sp_dense_fill_kernel, dim3(grid_for(sp.nnz)), dim3(kBlock), 0, s, sp.nnz, kb, P.Bi[sp.cur].get(), P.colof[sp.cur].get(),
                                   P.Bx[sp.cur].get(), P.cposl.get(), D.get());
        } else if (!tearing) {
            hipLaunchKernelGGL(lu_dense_fill_kernel, dim3((kb + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, s, kb, bcol.get(), Bp, Bi, Bx,
                               rloc.get(), D.get());
        } else {
            // ---- 2b. the spikes through the row singleton pivots (forward substitution, 64 spikes at a time)
            DevBuf<u64> &lkey = W.lkey, &lkey2 = W.lkey2;
            DevBuf<double> &lval = W.lval, &lval2 = W.lval2;
            for (DevBuf<u64>* b : {&lkey, &lkey2}) b->ensure(nz1);
            for (DevBuf<double>* b : {&lval, &lval2}) b->ensure(nz1);
            hipLaunchKernelGGL(lu_lentry_keys_kernel, dim3(grid_for(nb)), dim3(kBlock), 0, s, nb, colof.get(), Bi, Bx, ckind.get(),
                               pivrow.get(), rstage.get(), cstage.get(), rloc.get(), pivot.get(), npiv_sing, lkey.get(), lval.get());
            sort_keys(T, lkey.get(), lkey2.get(), lval.get(), lval2.get(), (size_t)nb, 64, s);
            W.lrp.ensure(d1 + 1);
            hipLaunchKernelGGL(lu_colptr_kernel, dim3(grid_for(dim + 1)), dim3(kBlock), 0, s, dim, nb, lkey2.get(), W.lrp.get());
            // first stage of every half-round (the stage sort left the tags in W.skey2), and which of them have work
            const int ntags = npiv_sing > 0 ? 2 * rounds : 0;
            W.tagptr.ensure((size_t)ntags + 1); W.tagwork.ensure((size_t)ntags);
            hipLaunchKernelGGL(lu_colptr_kernel, dim3(grid_for(ntags + 1)), dim3(kBlock), 0, s, ntags, (int64_t)dim, W.skey2.get(), W.tagptr.get());
            hipLaunchKernelGGL(lu_tagwork_kernel, dim3(grid_for(ntags)), dim3(kBlock), 0, s, ntags, W.tagptr.get(), W.lrp.get(), W.tagwork.get());
            std::vector<ipxint> tagptr((size_t)ntags + 1);
            std::vector<int> tagwork((size_t)ntags);
            IPXK_HIP(hipMemcpyAsync(tagptr.data(), W.tagptr.get(), tagptr.size() * sizeof(ipxint), hipMemcpyDeviceToHost, s));
            IPXK_HIP(hipMemcpyAsync(tagwork.data(), W.tagwork.get(), tagwork.size() * sizeof(int), hipMemcpyDeviceToHost, s));
            IPXK_HIP(hipStreamSynchronize(s));
            // the batches of 64 spikes are independent: a GROUP of them travels together (one launch per half-round / run for the
            // whole group), as many as IPXK_LU_SPIKE_MEM_MB (4096) of dim x 64 blocks allow
            const int nbatches = (kb + kSpikeBatch - 1) / kSpikeBatch;
            const size_t xs = (size_t)dim * kSpikeBatch;
            size_t mem_mb = 4096;
            if (const char* e = getenv("IPXK_LU_SPIKE_MEM_MB")) mem_mb = (size_t)std::max(1, atoi(e));
            const int gmax = (int)std::max<size_t>(1, std::min<size_t>(256, (mem_mb << 20) / (xs * sizeof(double))));
            const int gsize = std::min(nbatches, gmax);
            W.X.ensure(xs * gsize);
            nspk = 0;
            for (int b0 = 0; b0 < nbatches; b0 += gsize) {
                const int G = std::min(gsize, nbatches - b0), c0 = b0 * kSpikeBatch;
                IPXK_HIP(hipMemsetAsync(W.X.get(), 0, xs * G * sizeof(double), s));
                hipLaunchKernelGGL(lu_spike_scatter_kernel, dim3(kSpikeBatch, G), dim3(kBlock), 0, s, kb, c0, xs, bcol.get(), Bp, Bi, Bx, rstage.get(),
                                   rloc.get(), npiv_sing, W.X.get());
                auto rows = [&](int64_t s0, int64_t s1) {
                    if (s1 <= s0) return;
                    const int64_t wgs = std::min<int64_t>(2048, (s1 - s0 + kBlock / 64 - 1) / (kBlock / 64));
                    hipLaunchKernelGGL(lu_spike_round_kernel, dim3((unsigned)wgs, G), dim3(kBlock), 0, s, (int)s0, (int)s1, xs, W.lrp.get(),
                                       lkey2.get(), lval2.get(), W.X.get());
                };
                // half-rounds of many rows: a launch over the chip each; runs of small ones: one workgroup per batch and run
                // (IPXK_LU_SPIKE_RUNS=0: a launch per half-round with work, as before)
                static const bool runs = !(getenv("IPXK_LU_SPIKE_RUNS") && getenv("IPXK_LU_SPIKE_RUNS")[0] == '0');
                constexpr int kSmallRows = 4 * (kSpikeRunThreads / 64);      // up to four rows per wavefront
                for (int t = 0; t < ntags;) {
                    const bool small = runs && tagptr[t + 1] - tagptr[t] <= kSmallRows;
                    if (!small) {
                        if (tagwork[t] > 0) rows(tagptr[t], tagptr[t + 1]);
                        t++;
                        continue;
                    }
                    int t1 = t, work = 0;
                    while (t1 < ntags && tagptr[t1 + 1] - tagptr[t1] <= kSmallRows) { work += tagwork[t1] > 0; t1++; }
                    if (work == 1) {                                     // (a lone half-round with work: the plain launch)
                        for (int q = t; q < t1; q++) if (tagwork[q] > 0) rows(tagptr[q], tagptr[q + 1]);
                    } else if (work > 1) {
                        hipLaunchKernelGGL(lu_spike_run_kernel, dim3(1, G), dim3(kSpikeRunThreads), 0, s, t, t1, xs, W.tagptr.get(), W.lrp.get(),
                                           lkey2.get(), lval2.get(), W.X.get());
                    }
                    t = t1;
                }
                rows(npiv_sing, dim);                                   // the rows that were never pivoted
                hipLaunchKernelGGL(lu_spike_dense_kernel, dim3(grid_for((int64_t)kb * kSpikeBatch), G), dim3(kBlock), 0, s, kb, c0, xs, npiv_sing,
                                   W.X.get(), D.get());
                IPXK_HIP(hipMemsetAsync(counters.get() + 3, 0, sizeof(int), s));
                hipLaunchKernelGGL(lu_spike_count_kernel, dim3(grid_for((int64_t)npiv_sing * kSpikeBatch), G), dim3(kBlock), 0, s, kb, c0, xs, npiv_sing,
                                   W.X.get(), counters.get() + 3);
                IPXK_HIP(hipMemcpyAsync(h, counters.get(), 8 * sizeof(int), hipMemcpyDeviceToHost, s));
                IPXK_HIP(hipStreamSynchronize(s));
                const int add = h[3];
                if (add > 0) {
                    grow_keep(W.spk_c, (size_t)nspk, (size_t)nspk + add, s);
                    grow_keep(W.spk_s, (size_t)nspk, (size_t)nspk + add, s);
                    grow_keep(W.spk_v, (size_t)nspk, (size_t)nspk + add, s);
                    const int cur = (int)nspk;
                    IPXK_HIP(hipMemcpyAsync(counters.get() + 3, &cur, sizeof(int), hipMemcpyHostToDevice, s));
                    hipLaunchKernelGGL(lu_spike_append_kernel, dim3(grid_for((int64_t)npiv_sing * kSpikeBatch), G), dim3(kBlock), 0, s, kb, c0, xs, npiv_sing,
                                       W.X.get(), counters.get() + 3, W.spk_c.get(), W.spk_s.get(), W.spk_v.get());
                    IPXK_HIP(hipStreamSynchronize(s));                  // `cur` is a stack variable
                    nspk += add;
                }
            }
        }
        Dense A{kb, D.get(), brstep.get(), bcstep.get(), bstep.get(), prow.get(), pcol.get(), abstol};
        static const bool two_level = !(getenv("IPXK_LU_TWO_LEVEL") && getenv("IPXK_LU_TWO_LEVEL")[0] == '0');
        if (kb <= kPanelThreads || !two_level) {
            const int width = kb <= kPanelThreads ? kPanel : kb <= 2 * kPanelThreads ? kNarrowWide : kb <= 4 * kPanelThreads ? kNarrow : kPanel;
            for (int c0 = 0; c0 < kb; c0 += width) {
                const int c1 = std::min(kb, c0 + width);
                if (kb <= kPanelThreads) hipLaunchKernelGGL(lu_panel_small_kernel, dim3(1), dim3(kPanelThreads), 0, s, A, c0, c1);
                else if (kb <= 2 * kPanelThreads) hipLaunchKernelGGL((lu_panel_multi_kernel<2, kNarrowWide>), dim3(1), dim3(kPanelThreads), 0, s, A, c0, c1, 1);
                else if (kb <= 4 * kPanelThreads) hipLaunchKernelGGL((lu_panel_multi_kernel<4, kNarrow>), dim3(1), dim3(kPanelThreads), 0, s, A, c0, c1, 1);
                else hipLaunchKernelGGL(lu_panel_kernel, dim3(1), dim3(kPanelThreads), 0, s, A, c0, c1);
                if (c1 < kb) {
                    hipLaunchKernelGGL(lu_panel_rows_kernel, dim3(grid_for(kb - c1)), dim3(kBlock), 0, s, A, c1, kb, 0);
                    hipLaunchKernelGGL(lu_trailing_kernel, dim3((kb + 63) / 64, (kb - c1 + 63) / 64), dim3(kBlock), 0, s, A, c1, kb, 0);
                }
            }
        } else {
            // two-level panels: sub-panels in registers (R rows per thread), the trailing matrix once per kPanel columns
            int W = kb <= 2 * kPanelThreads ? kNarrowWide : kb <= 4 * kPanelThreads ? kNarrow : kb <= 8 * kPanelThreads ? kNarrowDeep :
                    kb <= 16 * kPanelThreads ? kNarrowHuge : kNarrowGiant;
            if (const char* e = getenv("IPXK_LU_PANEL_W")) {               // (tests: a narrower sub-panel than the bump needs -- more rows per thread)
                const int w = atoi(e);
                if ((w == 1 || w == 2 || w == 4 || w == 8 || w == 16) && w <= W) W = w;
            }
            // the matrix cores for the trailing update of large bumps (IPXK_LU_MFMA_MIN rows and more, default 1025; 0: never)
            const char* mfma_env = getenv("IPXK_LU_MFMA_MIN");                 // (read per factorization: the tests switch it)
            const int mfma_min = mfma_env ? atoi(mfma_env) : kPanelThreads + 1;
            const bool use_mfma = mfma_min > 0 && kb >= mfma_min;
            LuWork& W_ = S->work;
            if (use_mfma) W_.ubuf.ensure((size_t)kPanel * kb);
            // the sub-panel's rows of U and its update of the rest of the outer panel in one launch (IPXK_LU_FUSED_SUB=0: two)
            const bool fused_sub = !(getenv("IPXK_LU_FUSED_SUB") && getenv("IPXK_LU_FUSED_SUB")[0] == '0');
            W_.usub.ensure((size_t)kNarrowWideMax * kPanel);
            // LOOK-AHEAD (with the matrix cores; IPXK_LU_LOOKAHEAD=0: off): an outer panel's update of the NEXT outer panel's columns
            // runs first, on this stream; its update of everything beyond runs on a second stream while the next outer panel is
            // factorized here.  The outer panels use two sets of pivot lists / counters alternately (the late update still reads
            // its own), a row the next panel pivots meanwhile stays live for the late update (step >= that panel's count), and the
            // next panel's rows of U beyond its columns wait for the late update.  Every entry still receives each panel's update
            // exactly once, panels in order: the same factors bit for bit.
            // Measured (scripts/gpu_lu_fused_check.py): 107.7 -> 96.2 ms at 8000 rows with 32 compute units kept free for the panel
            // kernels (16: no gain; the same 32 spread over the mask's words: slower), nothing at 5000 rows -- so from 6144 rows on
            // (IPXK_LU_LOOKAHEAD=1: always with the matrix cores, =0: never).
            const char* look_env = getenv("IPXK_LU_LOOKAHEAD");
            const bool lookahead = use_mfma && (look_env ? look_env[0] != '0' : kb >= 6144);
            Dense Ap[2] = {A, A};
            Ap[1].bstep = bstep.get() + 4; Ap[1].prow = prow.get() + kPanel; Ap[1].pcol = pcol.get() + kPanel;
            if (lookahead && !W_.s2) {
                // the late update leaves some compute units to the panel kernels of the first stream (a one-workgroup kernel of 1024
                // threads does not get a slot on a chip that a 15 000-workgroup kernel keeps full): IPXK_LU_LOOKAHEAD_FREE_CUS
                int free_cus = 32;
                if (const char* e = getenv("IPXK_LU_LOOKAHEAD_FREE_CUS")) free_cus = std::max(0, std::min(128, atoi(e)));
                uint32_t mask[8];
                for (int w = 0; w < 8; w++) mask[w] = 0xffffffffu;
                const bool spread = getenv("IPXK_LU_LOOKAHEAD_SPREAD") != nullptr;       // (measurement: the free units taken from all eight words of the mask)
                // (bit b of the mask = unit b / 8 of XCC b % 8, scripts/bench_cumask.hip; an XCC whose bits are all clear keeps all its units: the
                // default frees 4 units of every XCD.  Measured and not used: 16 / 24 units of XCC 0 alone for a one-XCD panel of 16 / 24
                // workgroups with two rows per thread -- workgroups go to the XCCs in turn whatever the mask says, so the late update's share
                // on XCC 0 crawls on what is left of it: 8000 rows 62.7 -> 82 / 106 ms, 12 000 rows 149 -> 205 / 324 ms)
                int xcc0 = 0;
                if (const char* e = getenv("IPXK_LU_LOOKAHEAD_XCC0")) xcc0 = std::max(0, std::min(31, atoi(e)));    // (measurement: that many units of XCC 0 only)
                for (int b = 0; b < (xcc0 ? xcc0 : free_cus); b++) {
                    const int bit = xcc0 ? 8 * b : spread ? (b % 8) * 32 + b / 8 : b;
                    mask[bit / 32] &= ~(1u << (bit % 32));
                }
                if (free_cus > 0 && hipExtStreamCreateWithCUMask(&W_.s2, 8, mask) != hipSuccess) {
                    (void)hipGetLastError();               // (a device the mask does not fit: a plain stream -- correct, no overlap to speak of)
                    W_.s2 = nullptr;
                }
                if (!W_.s2) IPXK_HIP(hipStreamCreateWithFlags(&W_.s2, hipStreamNonBlocking));
                for (hipEvent_t* e : {&W_.ev_rows[0], &W_.ev_rows[1], &W_.ev_trail[0], &W_.ev_trail[1]})
                    IPXK_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
            }
            // the outer panel by ONE launch of <= 32 cooperating workgroups (IPXK_LU_COOP=0: the sub-panel launches below)
            const bool coop = !(getenv("IPXK_LU_COOP") && getenv("IPXK_LU_COOP")[0] == '0') && !getenv("IPXK_LU_PANEL_W");
            // 1 row per thread up to 8192 rows (G <= 32), 2 beyond (G <= 64).  Measured at 8000 rows: 1 and 2 rows per thread 68.9 ms
            // both (the exchange, not the width of the barrier, is what a column costs: ~5 us); 4 rows per thread spill to scratch
            // (146 ms) and are not used
            int coopR = kb <= 8 * kPanelThreads ? 1 : 2;
            if (const char* e = getenv("IPXK_LU_COOP_R")) {                   // (measurement)
                const int r = atoi(e);
                if ((r == 1 || r == 2) && r >= coopR) coopR = r;
            }
            const int coopG = (kb + coopR * kCoopThreads - 1) / (coopR * kCoopThreads);
            int coop_steps_done = 0;
            // the participants on ONE XCD (its L2 is coherent: messages by plain stores) where one workgroup per compute unit of that
            // XCD holds them all and nothing else competes for the XCD: at most 24 workgroups, no look-ahead (blocks of up to 6144
            // rows).  Measured: 2600 / 5000 rows 16.0 / 37.4 -> 14.4 / 33.0 ms; with the look-ahead's late update on the other stream
            // the participants wait for compute units of their XCD, 8000 rows 62.8 -> 76.3 ms, 12 000 rows 149 -> 193 ms, so those
            // blocks keep all XCDs and write-through messages.  (IPXK_LU_COOP_XCD=0: never, =1: wherever at most 32 workgroups take part)
            const char* xcd_env = getenv("IPXK_LU_COOP_XCD");
            const bool coop_xcd = coop && !(xcd_env && xcd_env[0] == '0') && ((xcd_env && xcd_env[0] == '1') ? coopG <= 32 : (!lookahead && coopG <= 24));
            if (coop) {
                W_.coop_slots.ensure((size_t)5 * kCoopMaxG * kCoopSlot); W_.coop_bar.ensure(2);
                if (W_.coop_xcc.size() < (size_t)kCoopMaxG) { W_.coop_xcc.ensure((size_t)kCoopMaxG); IPXK_HIP(hipMemsetAsync(W_.coop_xcc.get(), 0, kCoopMaxG * sizeof(unsigned long long), s)); }
                IPXK_HIP(hipMemsetAsync(W_.coop_bar.get(), 0, 2 * sizeof(unsigned), s));
                hipLaunchKernelGGL(lu_fill_u64_kernel, dim3(8), dim3(kBlock), 0, s, (int64_t)5 * kCoopMaxG * kCoopSlot, (u64)kCoopSentinel,
                                   reinterpret_cast<u64*>(W_.coop_slots.get()));
            }
            int k = 0, last_late = -1;                  // outer panel index; the last outer panel with a late update in flight
            for (int c0 = 0; c0 < kb; c0 += kPanel, k++) {
                const int c1o = std::min(kb, c0 + kPanel);
                const Dense& P = lookahead ? Ap[k & 1] : A;
                const int* step_src = (lookahead && k > 0) ? Ap[(k - 1) & 1].bstep : nullptr;
                if (coop) {
                    if (++W_.coop_epoch == 0) ++W_.coop_epoch;
                    const Coop C{W_.coop_slots.get(), coop_steps_done % 5, reinterpret_cast<int*>(W_.coop_bar.get() + 1), coop_xcd ? 1 : 0, W_.coop_epoch,
                                 W_.coop_xcc.get()};
                    const int grid = coop_xcd ? coopG * 8 : coopG;
                    if (coopR == 1) hipLaunchKernelGGL((lu_panel_coop_kernel<1>), dim3(grid), dim3(kCoopThreads), 0, s, P, C, c0, c1o, step_src);
                    else hipLaunchKernelGGL((lu_panel_coop_kernel<2>), dim3(grid), dim3(kCoopThreads), 0, s, P, C, c0, c1o, step_src);
                    coop_steps_done += c1o - c0;
                }
                // (measured and dropped: the whole outer panel in ONE launch, the sub-panels' updates of the rest of the outer
                // panel by that one workgroup too -- bit-identical, but one CU moves those kb x 28 columns at 50-100 GB/s:
                // 228 ms at 8000 rows against 130 with the three launches per sub-panel below)
                for (int ci = c0; ci < c1o && !coop; ci += W) {
                    const int ce = std::min(c1o, ci + W), first = ci == c0 ? 1 : 0;
                    const double* us = fused_sub ? W_.usub.get() : nullptr;
                    if (W == kNarrowWide) hipLaunchKernelGGL((lu_panel_multi_kernel<2, kNarrowWide>), dim3(1), dim3(kPanelThreads), 0, s, P, ci, ce, first, us, c1o, step_src);
                    else if (W == kNarrow) hipLaunchKernelGGL((lu_panel_multi_kernel<4, kNarrow>), dim3(1), dim3(kPanelThreads), 0, s, P, ci, ce, first, us, c1o, step_src);
                    else if (W == kNarrowDeep) hipLaunchKernelGGL((lu_panel_multi_kernel<8, kNarrowDeep>), dim3(1), dim3(kPanelThreads), 0, s, P, ci, ce, first, us, c1o, step_src);
                    else if (W == kNarrowHuge) hipLaunchKernelGGL((lu_panel_multi_kernel<16, kNarrowHuge>), dim3(1), dim3(kPanelThreads), 0, s, P, ci, ce, first, us, c1o, step_src);
                    else hipLaunchKernelGGL((lu_panel_multi_kernel<32, kNarrowGiant>), dim3(1), dim3(kPanelThreads), 0, s, P, ci, ce, first, us, c1o, step_src);
                    if (ce < c1o) {         // the rest of the outer panel
                        if (fused_sub) {
                            hipLaunchKernelGGL(lu_subpanel_update_kernel, dim3((kb + 63) / 64), dim3(kBlock), 0, s, P, ce, c1o, W_.usub.get());
                        } else {
                            hipLaunchKernelGGL(lu_panel_rows_kernel, dim3(1), dim3(kBlock), 0, s, P, ce, c1o, 1);
                            hipLaunchKernelGGL(lu_trailing_kernel, dim3((kb + 63) / 64, 1), dim3(kBlock), 0, s, P, ce, c1o, 1);
                        }
                    }
                }
                if (c1o < kb) {
                    if (lookahead) {
                        if (last_late >= 0) IPXK_HIP(hipStreamWaitEvent(s, W_.ev_trail[last_late & 1], 0));      // the columns beyond are up to date
                        hipLaunchKernelGGL(lu_panel_rows_kernel, dim3(grid_for(kb - c1o)), dim3(kBlock), 0, s, P, c1o, kb, 2, W_.ubuf.get(), kb);
                        IPXK_HIP(hipEventRecord(W_.ev_rows[k & 1], s));
                        const int cl = std::min(kb, c1o + kPanel);
                        hipLaunchKernelGGL(lu_trailing_mfma_kernel, dim3((kb + 63) / 64, 1), dim3(kBlock), 0, s, P, W_.ubuf.get(), kb, c1o, cl, c1o);
                        last_late = -1;
                        if (cl < kb) {
                            IPXK_HIP(hipStreamWaitEvent(W_.s2, W_.ev_rows[k & 1], 0));
                            hipLaunchKernelGGL(lu_trailing_mfma_kernel, dim3((kb + 63) / 64, (kb - cl + 63) / 64), dim3(kBlock), 0, W_.s2, P, W_.ubuf.get(), kb,
                                               cl, kb, c1o);
                            IPXK_HIP(hipEventRecord(W_.ev_trail[k & 1], W_.s2));
                            last_late = k;
                        }
                    } else if (use_mfma) {
                        hipLaunchKernelGGL(lu_panel_rows_kernel, dim3(grid_for(kb - c1o)), dim3(kBlock), 0, s, P, c1o, kb, 2, W_.ubuf.get(), kb);
                        hipLaunchKernelGGL(lu_trailing_mfma_kernel, dim3((kb + 63) / 64, (kb - c1o + 63) / 64), dim3(kBlock), 0, s, P, W_.ubuf.get(), kb,
                                           c1o, kb, c1o);
                    } else {
                        hipLaunchKernelGGL(lu_panel_rows_kernel, dim3(grid_for(kb - c1o)), dim3(kBlock), 0, s, P, c1o, kb, 2);
                        hipLaunchKernelGGL(lu_trailing_kernel, dim3((kb + 63) / 64, (kb - c1o + 63) / 64), dim3(kBlock), 0, s, P, c1o, kb, 2);
                    }
                }
            }
            if (last_late >= 0) IPXK_HIP(hipStreamWaitEvent(s, W_.ev_trail[last_late & 1], 0));
            if (lookahead && k > 0 && ((k - 1) & 1))        // the counters of the last outer panel to where they are read
                IPXK_HIP(hipMemcpyAsync(bstep.get(), bstep.get() + 4, 2 * sizeof(int), hipMemcpyDeviceToDevice, s));
        }
        IPXK_HIP(hipMemcpyAsync(h, bstep.get(), 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        h[2] = 0;
        if (S->work.coop_bar.size() >= 2) IPXK_HIP(hipMemcpyAsync(h + 2, S->work.coop_bar.get() + 1, sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        if (h[2]) throw Error(IPXK_E_HIP, "LU: the cooperative panel kernel gave up waiting for its workgroups (IPXK_LU_COOP=0 selects the one-workgroup panels)");
        bpiv = h[0];
    }
    const int ndep = kb - bpiv;
    I.num_dependent = ndep;
    S->ndep = ndep;
    S->bump_start = npiv_sing;
    S->bump_size = bpiv;
    S->dependent.ensure((size_t)std::max(ndep, 1));
    if (kb > 0) {
        const int gk = grid_for(kb);
        // pivots of the bump, then the dependent columns paired with the left-over rows, both ascending
        hipLaunchKernelGGL(lu_bump_stage_kernel, dim3(gk), dim3(kBlock), 0, s, kb, npiv_sing, bcstep.get(), bcol.get(),
                           cstage.get(), flag.get(), ckind.get());
        scan_exclusive(T, flag.get(), rank.get(), (size_t)kb, s);
        hipLaunchKernelGGL(lu_bump_rest_kernel, dim3(gk), dim3(kBlock), 0, s, kb, npiv_sing + bpiv, flag.get(), rank.get(),
                           bcol.get(), cstage.get(), ckind.get(), S->dependent.get());
        hipLaunchKernelGGL(lu_bump_stage_kernel, dim3(gk), dim3(kBlock), 0, s, kb, npiv_sing, brstep.get(), brow.get(),
                           rstage.get(), flag.get(), (unsigned char*)nullptr);
        scan_exclusive(T, flag.get(), rank.get(), (size_t)kb, s);
        hipLaunchKernelGGL(lu_bump_rest_kernel, dim3(gk), dim3(kBlock), 0, s, kb, npiv_sing + bpiv, flag.get(), rank.get(),
                           brow.get(), rstage.get(), (unsigned char*)nullptr, (ipxint*)nullptr);
    }
    const double t2 = now_s();
    // ---- 3. assembly
    S->rowperm.ensure(d1); S->colperm.ensure(d1);
    S->Lp.ensure(d1 + 1); S->Up.ensure(d1 + 1);
    const int64_t kbsq = (int64_t)kb * kb;
    // the entries outside the dense block: B itself, or (after elimination rounds) the list of the entries that left the
    // current matrix, with the values they had then
    const int64_t nbB = nb;
    if (sparse_done) nb = sp.ne;
    const int* asm_row = sparse_done ? W.sp.Erow.get() : Bi;
    const int* asm_col = sparse_done ? W.sp.Ecol.get() : colof.get();
    const double* asm_val = sparse_done ? W.sp.Eval.get() : Bx;
    const int64_t nl = nb + kbsq, nu = nb + kbsq + dim + nspk;
    int64_t lnz = 0, unz = 0;
    if (dim > 0) {
        DevBuf<u64> &lkey = W.lkey, &lkey2 = W.lkey2, &ukey = W.ukey, &ukey2 = W.ukey2;
        DevBuf<double> &lval = W.lval, &lval2 = W.lval2, &uval = W.uval, &uval2 = W.uval2;
        for (DevBuf<u64>* b : {&lkey, &lkey2}) b->ensure((size_t)std::max<int64_t>(nl, 1));
        for (DevBuf<u64>* b : {&ukey, &ukey2}) b->ensure((size_t)nu);
        for (DevBuf<double>* b : {&lval, &lval2}) b->ensure((size_t)std::max<int64_t>(nl, 1));
        for (DevBuf<double>* b : {&uval, &uval2}) b->ensure((size_t)nu);
        if (nl > 0) hipLaunchKernelGGL(lu_fill_u64_kernel, dim3(grid_for(nl)), dim3(kBlock), 0, s, nl, kNoKey, lkey.get());
        hipLaunchKernelGGL(lu_fill_u64_kernel, dim3(grid_for(nu)), dim3(kBlock), 0, s, nu, kNoKey, ukey.get());
        const int kshift = bits_for((int64_t)dim + 1);            // 2^kshift > dim
        Assemble A{dim, kb, Bp, asm_row, asm_col, asm_val, rstage.get(), cstage.get(), rloc.get(), cloc.get(), brow.get(), bcol.get(),
                   bcstep.get(), pivot.get(), D.get(), ckind.get(), lkey.get(), ukey.get(), lval.get(), uval.get(), tearing ? 1 : 0, kshift};
        if (nb > 0) hipLaunchKernelGGL(lu_keys_sparse_kernel, dim3(grid_for(nb)), dim3(kBlock), 0, s, A, nb);
        if (kb > 0) hipLaunchKernelGGL(lu_keys_dense_kernel, dim3(grid_for(kbsq)), dim3(kBlock), 0, s, A, nb);
        hipLaunchKernelGGL(lu_keys_unit_kernel, dim3(g), dim3(kBlock), 0, s, A, nb + kbsq);
        if (nspk > 0)
            hipLaunchKernelGGL(lu_keys_spike_kernel, dim3(grid_for(nspk)), dim3(kBlock), 0, s, A, nb + kbsq + dim, (int)nspk,
                               W.spk_c.get(), W.spk_s.get(), W.spk_v.get());
        // (2 * kshift key bits instead of 64: five radix passes instead of eight at 1M rows; the unused slots hold all ones and
        // sort behind every key)
        if (nl > 0) sort_keys(T, lkey.get(), lkey2.get(), lval.get(), lval2.get(), (size_t)nl, 2 * kshift, s);
        sort_keys(T, ukey.get(), ukey2.get(), uval.get(), uval2.get(), (size_t)nu, 2 * kshift, s);
        hipLaunchKernelGGL(lu_colptr_kernel, dim3(grid_for(dim + 1)), dim3(kBlock), 0, s, dim, nl, lkey2.get(), S->Lp.get(), kshift);
        hipLaunchKernelGGL(lu_colptr_kernel, dim3(grid_for(dim + 1)), dim3(kBlock), 0, s, dim, nu, ukey2.get(), S->Up.get(), kshift);
        ipxint ends[2] = {0, 0};
        IPXK_HIP(hipMemcpyAsync(&ends[0], S->Lp.get() + dim, sizeof(ipxint), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipMemcpyAsync(&ends[1], S->Up.get() + dim, sizeof(ipxint), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        lnz = ends[0]; unz = ends[1];
        S->Li.ensure((size_t)std::max<int64_t>(lnz, 1)); S->Lx.ensure((size_t)std::max<int64_t>(lnz, 1));
        S->Ui.ensure((size_t)std::max<int64_t>(unz, 1)); S->Ux.ensure((size_t)std::max<int64_t>(unz, 1));
        if (lnz > 0) {
            hipLaunchKernelGGL(lu_rowidx_kernel, dim3(grid_for(lnz)), dim3(kBlock), 0, s, lnz, lkey2.get(), S->Li.get(), kshift);
            IPXK_HIP(hipMemcpyAsync(S->Lx.get(), lval2.get(), (size_t)lnz * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL(lu_rowidx_kernel, dim3(grid_for(unz)), dim3(kBlock), 0, s, unz, ukey2.get(), S->Ui.get(), kshift);
        IPXK_HIP(hipMemcpyAsync(S->Ux.get(), uval2.get(), (size_t)unz * sizeof(double), hipMemcpyDeviceToDevice, s));
        IPXK_HIP(hipMemsetAsync(counters.get() + 6, 0, sizeof(int), s));
        hipLaunchKernelGGL(lu_perm_kernel, dim3(g), dim3(kBlock), 0, s, dim, rstage.get(), S->rowperm.get(), counters.get() + 6);
        hipLaunchKernelGGL(lu_perm_kernel, dim3(g), dim3(kBlock), 0, s, dim, cstage.get(), S->colperm.get(), counters.get() + 6);
        IPXK_HIP(hipMemcpyAsync(h, counters.get(), 8 * sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));               // the key buffers go out of scope
        if (h[6]) throw Error(IPXK_E_HIP, "LU: a pivot stage is missing");
    } else {
        IPXK_HIP(hipMemsetAsync(S->Lp.get(), 0, sizeof(ipxint), s));
        IPXK_HIP(hipMemsetAsync(S->Up.get(), 0, sizeof(ipxint), s));
        IPXK_HIP(hipStreamSynchronize(s));
    }
    S->lnz = lnz; S->unz = unz;
    I.lnz = lnz; I.unz = unz;
    I.seconds_singletons = t1 - t0;
    I.seconds_bump = t2 - t1;
    I.seconds_assemble = now_s() - t2;
    S->valid = true;
    S->pivottol_used = pivottol;
    S->strict_used = strict;
    S->last_info = I;
    S->generation++;
    if (info) *info = I;
    if (getenv("IPXK_VERBOSE"))
        fprintf(stderr, "ipxk: LU dim %d nnz %lld: %lld column + %lld row singletons in %d rounds (%.2f ms), bump %d (%.2f ms, %d dependent), "
                "assembly %.2f ms; nnz(L) %lld nnz(U) %lld; %d sparse pivots in %d elimination rounds, %d spikes\n", dim, (long long)nbB, (long long)I.col_singletons, (long long)I.row_singletons,
                rounds, I.seconds_singletons * 1e3, kb, I.seconds_bump * 1e3, ndep, I.seconds_assemble * 1e3, (long long)lnz, (long long)unz,
                (int)I.sparse_pivots, (int)I.sparse_rounds, (int)I.spikes);
}

namespace {
// column j of the packed matrix against column cand[j] of [A I] as the context holds it: same length, same rows in the same
// order, same values (bit patterns); every candidate at most once
__global__ void lu_reuse_compare_kernel(int m, int n, const int* __restrict__ cand, const int* __restrict__ Bp, const int* __restrict__ Bi,
                                        const double* __restrict__ Bx, const int* __restrict__ Ap, const int* __restrict__ Ai,
                                        const double* __restrict__ Ax, int* __restrict__ mark, int* __restrict__ flag) {
    IPXK_GRID_STRIDE(j, m) {
        const int col = cand[j];
        const int b = Bp[j], len = Bp[j + 1] - b;
        bool same;
        if (col >= n) {
            same = len == 1 && Bi[b] == col - n && Bx[b] == 1.0;
        } else {
            const int a = Ap[col];
            same = len == Ap[col + 1] - a;
            for (int t = 0; same && t < len; t++)
                same = Bi[b + t] == Ai[a + t] && __double_as_longlong(Bx[b + t]) == __double_as_longlong(Ax[a + t]);
        }
        if (atomicExch(&mark[col], (int)j + 1) != 0) same = false;
        if (!same) *flag = 1;
    }
}
// position k of the resident basis -> the caller's column that holds the same variable
__global__ void lu_reuse_sigma_kernel(int m, const ipxint* __restrict__ basis, const int* __restrict__ mark, int* __restrict__ sigma, int* __restrict__ flag) {
    IPXK_GRID_STRIDE(k, m) {
        const int q = mark[basis[k]];
        if (q == 0) *flag = 1;
        sigma[k] = q - 1;
    }
}
__global__ void lu_reuse_colperm_kernel(int m, const ipxint* __restrict__ colperm, const int* __restrict__ sigma, ipxint* __restrict__ out) {
    IPXK_GRID_STRIDE(t, m) out[t] = sigma[colperm[t]];
}
}  // namespace

// Basis::Load / Basis::Factorize (src/basis.cc:81-156) of the reference's Basis right after Maxvolume on the device ask for the
// factorization of the basis whose factors this context ALREADY holds (ipxk_lu_factorize_basis at the end of ipxk_maxvolume), its
// columns in another order.  If the matrix handed in is exactly that basis -- every column compared entry by entry on the device
// with the column of [A I] its offsets name, the set of columns equal to the resident basis -- and the tolerances are the ones the
// resident factors were computed with, nothing is computed: the factors stay, and ipxk_lu_get_factors returns the column
// permutation in the caller's numbering.  IPXK_LU_REUSE=0 switches this off.
static bool lu_reuse_resident(Context* c, LuState* S, int dim, const ipxint* Bbegin, const ipxint* Bend, const int* dBp, const int* dBi,
                              const double* dBx, double pivottol, bool strict) {
    static const bool off = getenv("IPXK_LU_REUSE") && getenv("IPXK_LU_REUSE")[0] == '0';
    const int64_t m = c->m, n = c->n;
    if (off || !S->valid || !S->from_basis || S->ndep != 0 || dim == 0 || dim != (int)m || S->dim != dim || !c->have_plain) return false;
    if (pivottol != S->pivottol_used || strict != S->strict_used) return false;
    if ((int64_t)c->h_Ap.size() != n + 1) return false;
    const std::vector<ipxint>& Ap = c->h_Ap;
    std::vector<int> cand((size_t)dim);
    for (int j = 0; j < dim; j++) {
        const ipxint b = Bbegin[j], e = Bend[j];
        if (b >= Ap[(size_t)n]) {                        // a slack column of [A I]: one entry, stored behind the structural ones
            const ipxint i = b - Ap[(size_t)n];
            if (i >= m || e != b + 1) return false;
            cand[(size_t)j] = (int)(n + i);
        } else {
            const int64_t col = (std::upper_bound(Ap.begin(), Ap.end(), b) - Ap.begin()) - 1;
            if (col < 0 || col >= n || Ap[(size_t)col] != b || Ap[(size_t)col + 1] != e) return false;
            cand[(size_t)j] = (int)col;
        }
    }
    hipStream_t s = c->stream;
    S->reuse_cand.upload(cand, s);
    S->reuse_mark.ensure((size_t)(n + m)); S->reuse_sigma.ensure((size_t)dim); S->reuse_flag.ensure(1); S->colperm_view.ensure((size_t)dim);
    IPXK_HIP(hipMemsetAsync(S->reuse_mark.get(), 0, (size_t)(n + m) * sizeof(int), s));
    IPXK_HIP(hipMemsetAsync(S->reuse_flag.get(), 0, sizeof(int), s));
    const int g = grid_for(dim);
    hipLaunchKernelGGL(lu_reuse_compare_kernel, dim3(g), dim3(kBlock), 0, s, dim, (int)n, S->reuse_cand.get(), dBp, dBi, dBx, c->pl_Ap.get(),
                       c->pl_Ai.get(), c->pl_Ax.get(), S->reuse_mark.get(), S->reuse_flag.get());
    hipLaunchKernelGGL(lu_reuse_sigma_kernel, dim3(g), dim3(kBlock), 0, s, dim, S->basis.get(), S->reuse_mark.get(), S->reuse_sigma.get(),
                       S->reuse_flag.get());
    int flag = 0;
    IPXK_HIP(hipMemcpyAsync(&flag, S->reuse_flag.get(), sizeof(int), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    if (flag) return false;
    hipLaunchKernelGGL(lu_reuse_colperm_kernel, dim3(g), dim3(kBlock), 0, s, dim, S->colperm.get(), S->reuse_sigma.get(), S->colperm_view.get());
    S->view = true;
    if (getenv("IPXK_VERBOSE")) fprintf(stderr, "ipxk: LU dim %d: the resident factors of this basis handed out again\n", dim);
    return true;
}

long lu_generation(const Context* c) { return c->lu ? c->lu->generation : 0; }

void lu_invalidate(Context* c) { if (c->lu) c->lu->valid = false; }

bool lu_view(const Context* c, LuView* out) {
    const LuState* S = c->lu;
    if (!S || !S->valid) return false;
    out->dim = S->dim;
    out->ndep = S->ndep;
    out->from_basis = S->from_basis;
    out->bump_start = S->bump_start;
    out->bump_size = S->bump_size;
    out->F = DeviceFactors{S->Lp.get(), S->Li.get(), S->Up.get(), S->Ui.get(), S->Lx.get(), S->Ux.get(), S->lnz, S->unz};
    out->rowperm = S->rowperm.get();
    out->colperm = S->colperm.get();
    out->basis = S->basis.get();
    return true;
}

// the plain CSC copy of the structural matrix that ipxk_lu_factorize_basis keeps on the device
void lu_plain_matrix(const Context* c, const int** Ap, const int** Ai, const double** Ax) {
    const LuState* S = c->lu;
    IPXK_REQUIRE(S && S->have_A, "no resident copy of the matrix (ipxk_lu_factorize_basis)");
    *Ap = c->pl_Ap.get(); *Ai = c->pl_Ai.get(); *Ax = c->pl_Ax.get();
}

void lu_factorize_host(Context* c, int64_t dim64, const ipxint* Bbegin, const ipxint* Bend, const ipxint* Bi,
                       const double* Bx, double pivottol, bool strict, ipxk_lu_info* info) {
    IPXK_REQUIRE(dim64 >= 0 && dim64 < (int64_t(1) << 30), "dimension out of range");
    IPXK_REQUIRE(pivottol > 0.0 && pivottol <= 1.0, "pivottol must lie in (0,1]");
    const int dim = (int)dim64;
    hipStream_t s = c->stream;
    LuState* S = lu_state(c);
    // pack the columns (they are ranges of a larger array on the caller's side, src/basis.cc:122-128)
    std::vector<int> bp((size_t)dim + 1, 0), bi;
    std::vector<double> bx;
    int64_t nb = 0;
    for (int j = 0; j < dim; j++) {
        IPXK_REQUIRE(Bend[j] >= Bbegin[j], "Bend < Bbegin");
        nb += Bend[j] - Bbegin[j];
    }
    IPXK_REQUIRE(nb < (int64_t(1) << 31), "nnz(B) exceeds 32 bits");
    bi.resize((size_t)nb); bx.resize((size_t)nb);
    int64_t q = 0;
    for (int j = 0; j < dim; j++) {
        for (ipxint p = Bbegin[j]; p < Bend[j]; p++, q++) {
            IPXK_REQUIRE(Bi[p] >= 0 && Bi[p] < dim, "row index of B out of range");
            bi[(size_t)q] = (int)Bi[p];
            bx[(size_t)q] = Bx[p];
        }
        bp[(size_t)j + 1] = (int)q;
    }
    DevBuf<int> dBp, dBi;
    DevBuf<double> dBx;
    dBp.upload(bp, s); dBi.upload(bi, s); dBx.upload(bx, s);
    dBi.ensure(1); dBx.ensure(1);
    IPXK_HIP(hipStreamSynchronize(s));
    if (lu_reuse_resident(c, S, dim, Bbegin, Bend, dBp.get(), dBi.get(), dBx.get(), pivottol, strict)) {
        if (info) {
            *info = S->last_info;
            info->seconds_singletons = info->seconds_bump = info->seconds_assemble = 0.0;
            info->reused = 1;
        }
        return;
    }
    S->from_basis = false;
    lu_factorize_device(c, S, dim, nb, dBp.get(), dBi.get(), dBx.get(), pivottol, strict, info);
}

void lu_factorize_basis(Context* c, const ipxint* basis, double pivottol, bool strict, ipxk_lu_info* info) {
    IPXK_REQUIRE(pivottol > 0.0 && pivottol <= 1.0, "pivottol must lie in (0,1]");
    IPXK_REQUIRE(c->nranks == 1, "the basis path does not shard: run it as independent replicas");
    const int m = (int)c->m, n = (int)c->n;
    hipStream_t s = c->stream;
    LuState* S = lu_state(c);
    IPXK_REQUIRE(c->have_plain, "no resident copy of the matrix");
    S->have_A = true;
    S->basis.upload(basis, (size_t)m, s);
    S->basis.ensure(1);
    const size_t m1 = (size_t)std::max(m, 1);
    LuWork& W = S->work;
    DevBuf<int> &cnt = W.cnt, &dBp = W.Bp, &dBi = W.Bi;
    DevBuf<double>& dBx = W.Bx;
    DevBuf<int> bad(1);
    cnt.ensure(m1); dBp.ensure(m1 + 1);
    Tmp& T = W.T;
    IPXK_HIP(hipMemsetAsync(bad.get(), 0, sizeof(int), s));
    int64_t nb = 0;
    if (m > 0) {
        hipLaunchKernelGGL(lu_basis_count_kernel, dim3(grid_for(m)), dim3(kBlock), 0, s, m, n, S->basis.get(), c->pl_Ap.get(), cnt.get(), bad.get());
        scan_exclusive(T, cnt.get(), dBp.get(), (size_t)m, s);
        int last[2] = {0, 0}, hbad = 0;
        IPXK_HIP(hipMemcpyAsync(&last[0], dBp.get() + m - 1, sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipMemcpyAsync(&last[1], cnt.get() + m - 1, sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipMemcpyAsync(&hbad, bad.get(), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        if (hbad) throw Error(IPXK_E_ARGUMENT, "basis entry out of range");
        nb = (int64_t)last[0] + last[1];
        const int nb32 = (int)nb;
        IPXK_HIP(hipMemcpyAsync(dBp.get() + m, &nb32, sizeof(int), hipMemcpyHostToDevice, s));
        dBi.ensure((size_t)std::max<int64_t>(nb, 1)); dBx.ensure((size_t)std::max<int64_t>(nb, 1));
        hipLaunchKernelGGL(lu_basis_fill_kernel, dim3(grid_for(m)), dim3(kBlock), 0, s, m, n, S->basis.get(), c->pl_Ap.get(), c->pl_Ai.get(),
                           c->pl_Ax.get(), dBp.get(), dBi.get(), dBx.get());
        IPXK_HIP(hipStreamSynchronize(s));               // nb32
    } else {
        IPXK_HIP(hipMemsetAsync(dBp.get(), 0, sizeof(int), s));
        dBi.ensure(1); dBx.ensure(1);
    }
    lu_factorize_device(c, S, m, nb, dBp.get(), dBi.get(), dBx.get(), pivottol, strict, info);
    S->from_basis = true;
}

void lu_get_factors(Context* c, ipxint* Lp, ipxint* Li, double* Lx, ipxint* Up, ipxint* Ui, double* Ux,
                    ipxint* rowperm, ipxint* colperm, ipxint* dependent) {
    LuState* S = c->lu;
    IPXK_REQUIRE(S && S->valid, "no LU factorization in this context");
    hipStream_t s = c->stream;
    const size_t dim = (size_t)S->dim;
    if (Lp) S->Lp.download(Lp, dim + 1, s);
    if (Li) S->Li.download(Li, (size_t)S->lnz, s);
    if (Lx) S->Lx.download(Lx, (size_t)S->lnz, s);
    if (Up) S->Up.download(Up, dim + 1, s);
    if (Ui) S->Ui.download(Ui, (size_t)S->unz, s);
    if (Ux) S->Ux.download(Ux, (size_t)S->unz, s);
    if (rowperm) S->rowperm.download(rowperm, dim, s);
    if (colperm) (S->view ? S->colperm_view : S->colperm).download(colperm, dim, s);
    if (dependent) S->dependent.download(dependent, (size_t)S->ndep, s);
    IPXK_HIP(hipStreamSynchronize(s));
}

}  // namespace ipxk
