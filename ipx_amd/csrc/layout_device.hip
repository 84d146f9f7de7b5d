// Model upload and gather-matrix layouts built ON THE DEVICE (round 4).
//
// The reference's NormalMatrix stores a reference to the model and copies nothing (src/normal_matrix.h:20-27), so
// constructing a KKT solver is free; rounds 1-3 built every layout of the model matrix with single-threaded host
// loops (transpose, bucketing, std::sort per tile: 3.2 s at 1M x 2M, per solver object).  Here the matrix is uploaded
// once as it is (CSC, 64-bit indices), narrowed and validated, transposed, and cut into the XCD-sliced tile layout
// and the sorted sub-tile layout (internal.hpp) by radix sorts -- the same scheme nmatrix.hip uses for N:
//   * a STABLE sort of the entries, enumerated in storage order, by (tile, row in tile) IS the sliced layout:
//     tile pointers by binary search in the sorted keys, per-row byte counts from the runs of equal keys, indices
//     and values by a gather through the sorted positions;
//   * the sorted sub-tiles need two sorts: by (sub-tile, row) -- which numbers the slots -- and then, stably, by
//     (sub-tile, offset in the slice): ties keep the slot order, exactly the host builder's comparator.
// Every array equals the host builder's bit for bit (tests/test_gpu_layout.py compares them all); the host builder
// (spmv.hip) stays as that test's reference and as the path of matrices the device path does not cover
// (long rows, gathered vectors that fit an XCD's L2, gathers with locality: phased / fused / sorted-fused layouts).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <chrono>
#include <memory>

#include "context.hpp"

namespace ipxk {

namespace {

using u64 = unsigned long long;
#define IPXK_GS(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

int gridn(int64_t n) { return (int)std::min<int64_t>(8192, std::max<int64_t>(1, (n + kBlock - 1) / kBlock)); }
int bits_for(u64 maxval) { int b = 1; while (b < 64 && (maxval >> b) != 0) b++; return b; }

__global__ void narrow_kernel(int64_t nz, const ipxint* __restrict__ in, int* __restrict__ out, int limit, int* bad) {
    IPXK_GS(p, nz) {
        const ipxint v = in[p];
        if (v < 0 || v >= limit) *bad = 1;
        out[p] = (int)v;
    }
}
// column (row of the gather matrix) and position of every entry, enumerated row by row
__global__ void rowof_kernel(int nrows, const int* __restrict__ ptr, int* __restrict__ rowof, unsigned* __restrict__ pos) {
    IPXK_GS(r, nrows)
        for (int p = ptr[r]; p < ptr[r + 1]; p++) { rowof[p] = (int)r; if (pos) pos[p] = (unsigned)p; }
}
__global__ void gather_transposed_kernel(int64_t nz, const unsigned* __restrict__ perm, const int* __restrict__ colof,
                                         const double* __restrict__ Ax, int* __restrict__ Ti, double* __restrict__ Tx) {
    IPXK_GS(t, nz) {
        const unsigned p = perm[t];
        Ti[t] = colof[p];
        Tx[t] = Ax[p];
    }
}
template <class K>
__global__ void lower_bounds_kernel(int64_t count, int64_t nz, const K* __restrict__ sorted, u64 stride, int shift, unsigned* __restrict__ out) {
    // out[t] = first position whose key is >= t * stride (shift: keys are compared after >> shift)
    IPXK_GS(t, count) {
        const u64 want = (u64)t * stride;
        int64_t lo = 0, hi = nz;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (((u64)sorted[mid] >> shift) < want) lo = mid + 1; else hi = mid; }
        out[t] = (unsigned)lo;
    }
}
__global__ void row_pointers_kernel(int64_t m, int64_t nz, const unsigned* __restrict__ sorted_rows, int* __restrict__ Tp) {
    IPXK_GS(i, m + 1) {
        int64_t lo = 0, hi = nz;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted_rows[mid] < (unsigned)i) lo = mid + 1; else hi = mid; }
        Tp[i] = (int)lo;
    }
}
__global__ void max_len_kernel(int nrows, const int* __restrict__ ptr, int* out) {
    int best = 0;
    IPXK_GS(r, nrows) best = max(best, ptr[r + 1] - ptr[r]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d, 64));
    if ((threadIdx.x & 63) == 0 && best > 0) atomicMax(out, best);
}
__global__ void row_extent_kernel(int n, const int* __restrict__ rows, const int* __restrict__ ptr, int* __restrict__ ext) {
    IPXK_GS(l, n) { ext[2 * l] = ptr[rows[l]]; ext[2 * l + 1] = ptr[rows[l] + 1]; }
}
// key of the sliced layout: (tile, row in tile), tile = row block * ns + slice of the gathered index
__global__ void sliced_keys_kernel(int nrows, const int* __restrict__ ptr, const int* __restrict__ idx, int R, int ns, int slice,
                                   unsigned* __restrict__ key, unsigned* __restrict__ pos) {
    IPXK_GS(r, nrows) {
        const unsigned base = (unsigned)(r / R) * (unsigned)ns, rr = (unsigned)(r % R);
        for (int p = ptr[r]; p < ptr[r + 1]; p++) {
            key[p] = (base + (unsigned)(idx[p] / slice)) * (unsigned)R + rr;
            pos[p] = (unsigned)p;
        }
    }
}
// key of the sorted layout's first sort: (sub-tile, row in row block)
__global__ void sorted_keys1_kernel(int nrows, const int* __restrict__ ptr, const int* __restrict__ idx, int RB, int ns, int nsub, int slice,
                                    int half, unsigned* __restrict__ key, unsigned* __restrict__ pos) {
    IPXK_GS(r, nrows) {
        const unsigned tile0 = (unsigned)(r / RB) * (unsigned)ns, rr = (unsigned)(r % RB);
        for (int p = ptr[r]; p < ptr[r + 1]; p++) {
            const int sl = idx[p] / slice, off = idx[p] - sl * slice;
            const unsigned sub = (tile0 + (unsigned)sl) * (unsigned)nsub + (unsigned)min(off / half, nsub - 1);
            key[p] = sub * (unsigned)RB + rr;
            pos[p] = (unsigned)p;
        }
    }
}
// second sort: (sub-tile, offset in the slice), enumerated in slot order
__global__ void sorted_keys2_kernel(int64_t nz, const unsigned* __restrict__ key1s, const unsigned* __restrict__ perm1, const int* __restrict__ idx,
                                    int RB, int ns, int nsub, int slice, u64* __restrict__ key2, unsigned* __restrict__ pos) {
    IPXK_GS(e, nz) {
        const unsigned sub = key1s[e] / (unsigned)RB;
        const int sl = (int)((sub / (unsigned)nsub) % (unsigned)ns);
        const int off = idx[perm1[e]] - sl * slice;
        key2[e] = ((u64)sub << kSortedOffBits) | (u64)(unsigned)off;
        pos[e] = (unsigned)e;
    }
}
__global__ void sorted_fill_kernel(int64_t nz, const u64* __restrict__ key2s, const unsigned* __restrict__ perm2, const unsigned* __restrict__ perm1,
                                   const unsigned* __restrict__ sub_ptr, const double* __restrict__ val, unsigned* __restrict__ pack,
                                   double* __restrict__ out_val) {
    IPXK_GS(f, nz) {
        const u64 k = key2s[f];
        const unsigned e = perm2[f], sub = (unsigned)(k >> kSortedOffBits), off = (unsigned)(k & ((1u << kSortedOffBits) - 1u));
        pack[f] = ((e - sub_ptr[sub]) << kSortedOffBits) | off;
        out_val[f] = val[perm1[e]];
    }
}
// byte counts per key from the runs of equal keys (cnt is zero on entry); a run of more than 255 raises *overflow
__global__ void run_counts_kernel(int64_t nz, const unsigned* __restrict__ sorted, unsigned char* __restrict__ cnt, int* overflow) {
    IPXK_GS(e, nz) {
        const unsigned k = sorted[e];
        if (e > 0 && sorted[e - 1] == k) continue;
        int len = 1;
        while (e + len < nz && len <= 256 && sorted[e + len] == k) len++;
        if (len > 255) *overflow = 1;
        cnt[k] = (unsigned char)len;
    }
}
__global__ void gather_entries_kernel(int64_t nz, const unsigned* __restrict__ perm, const int* __restrict__ idx, const double* __restrict__ val,
                                      int* __restrict__ out_idx, double* __restrict__ out_val) {
    IPXK_GS(e, nz) {
        const unsigned p = perm[e];
        out_idx[e] = idx[p];
        out_val[e] = val[p];
    }
}
// [0] = largest tile, [2..3] = (64 bits) sum over the row blocks of their fullest slice's entries
__global__ void tile_stats_kernel(int nrb, int ns, const unsigned* __restrict__ tile_ptr, int* out) {
    int best_tile = 0;
    u64 dom = 0;
    IPXK_GS(rb, nrb) {
        unsigned best = 0;
        for (int sl = 0; sl < ns; sl++) best = max(best, tile_ptr[(size_t)rb * ns + sl + 1] - tile_ptr[(size_t)rb * ns + sl]);
        best_tile = max(best_tile, (int)best);
        dom += best;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { best_tile = max(best_tile, __shfl_xor(best_tile, d, 64)); dom += __shfl_xor(dom, d, 64); }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(out, best_tile);
        if (dom) atomicAdd(reinterpret_cast<u64*>(out + 2), dom);
    }
}
__global__ void max_range_kernel(int64_t count, const unsigned* __restrict__ ptr, int* out) {
    int best = 0;
    IPXK_GS(t, count) best = max(best, (int)(ptr[t + 1] - ptr[t]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d, 64));
    if ((threadIdx.x & 63) == 0 && best > 0) atomicMax(out, best);
}
// entries of a list of columns, one after the other (dense columns: precond.hip)
__global__ void gather_columns_kernel(int k, const int* __restrict__ cols, const int* __restrict__ off, const int* __restrict__ Ap,
                                      const int* __restrict__ Ai, const double* __restrict__ Ax, ipxint* __restrict__ out_i, double* __restrict__ out_x) {
    const int kk = blockIdx.x;
    if (kk >= k) return;
    const int j = cols[kk], p0 = Ap[j], len = Ap[j + 1] - p0, o = off[kk];
    for (int t = threadIdx.x; t < len; t += blockDim.x) { out_i[o + t] = Ai[p0 + t]; out_x[o + t] = Ax[p0 + t]; }
}
__global__ void widen_kernel(int64_t nz, const int* __restrict__ in, ipxint* __restrict__ out) { IPXK_GS(p, nz) out[p] = in[p]; }

// ---- accumulated tiles -----------------------------------------------------------------------
// key = (tile << 18 | offset in the slice), enumerated in storage order (a stable sort keeps that order among ties)
__global__ void acc_keys_kernel(int nrows, const int* __restrict__ ptr, const int* __restrict__ idx, int RB, int ns, int slice,
                                u64* __restrict__ key, unsigned* __restrict__ pos, int* __restrict__ rowof) {
    IPXK_GS(r, nrows) {
        const u64 tile0 = (u64)(r / RB) * (u64)ns;
        for (int p = ptr[r]; p < ptr[r + 1]; p++) {
            const int sl = idx[p] / slice, off = idx[p] - sl * slice;
            key[p] = ((tile0 + (u64)sl) << kSortedOffBits) | (u64)(unsigned)off;
            pos[p] = (unsigned)p;
            rowof[p] = (int)r;
        }
    }
}
// the entry words in sorted order: row in block << 18 | offset
__global__ void acc_words_kernel(int64_t nz, const u64* __restrict__ keys, const unsigned* __restrict__ perm, const int* __restrict__ rowof, int RB,
                                 unsigned* __restrict__ word) {
    IPXK_GS(e, nz) word[e] = ((unsigned)(rowof[perm[e]] % RB) << kSortedOffBits) | (unsigned)(keys[e] & ((1u << kSortedOffBits) - 1u));
}
// The batches of one tile, by ONE wavefront: the tile's entries are walked in address order, 64 candidates at a time
// (the entries that waited from the previous batch first, then the stream); a candidate is taken unless its row
// already has an entry in the current batch (stamp) or an earlier candidate of the same group has the same row
// (claim: the lowest lane wins) or the batch is full; whoever is not taken waits for the next batch, in order.
// Sequential by nature (a greedy list schedule), but only ~ne/64 steps per tile and all tiles in parallel.
__global__ __launch_bounds__(64) void acc_batch_kernel(int RB, const unsigned* __restrict__ tile_ptr, const unsigned* __restrict__ word,
                                                        unsigned* __restrict__ dst, unsigned* pendA, unsigned* pendB, unsigned* __restrict__ bstart,
                                                        unsigned* __restrict__ nbatch, u64* deferred_total) {
    extern __shared__ unsigned ab_lds[];
    unsigned* stamp = ab_lds;
    unsigned* claim = ab_lds + RB;
    const int tile = blockIdx.x, lane = threadIdx.x;
    const unsigned e0 = tile_ptr[tile];
    const int ne = (int)(tile_ptr[tile + 1] - e0);
    for (int r = lane; r < RB; r += 64) { stamp[r] = 0u; claim[r] = 0xffffffffu; }
    __syncthreads();
    unsigned cur = 1;
    int fill = 0, out = 0, nb = 0, cursor = 0, na = 0, ia = 0, nbp = 0;
    u64 ndef = 0;
    unsigned *pa = pendA + e0, *pb = pendB + e0;
    if (ne > 0) { if (lane == 0) bstart[e0] = e0; nb = 1; }
    const u64 lt = (1ull << lane) - 1ull;
    while (out < ne) {
        int n = 0, c = 0;
        if (ia < na) {
            n = min(64, na - ia);
            if (lane < n) c = (int)__hip_atomic_load(pa + ia + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // written by this wave: read it past L1
            ia += n;
        } else if (fill < kAccBatch && cursor < ne) {
            n = min(64, ne - cursor);
            c = cursor + lane;
            cursor += n;
        } else {                                   // close the batch
            if (lane == 0) bstart[e0 + nb] = e0 + (unsigned)out;
            nb++; cur++; fill = 0;
            unsigned* t = pa; pa = pb; pb = t;
            na = nbp; ia = 0; nbp = 0;
            __syncthreads();                       // the waiting list written by this wave is read back below
            continue;
        }
        const bool live = lane < n;
        const unsigned row = live ? (word[e0 + c] >> kSortedOffBits) : 0u;
        const bool ok = live && stamp[row] != cur;
        if (ok) atomicMin(&claim[row], (unsigned)lane);
        __syncthreads();
        const bool win = ok && claim[row] == (unsigned)lane;
        __syncthreads();
        if (win) { claim[row] = 0xffffffffu; stamp[row] = cur; }
        const u64 wmask = __ballot(win);
        const int prefix = __popcll(wmask & lt);
        const bool take = win && prefix < kAccBatch - fill;
        if (take) dst[e0 + c] = e0 + (unsigned)(out + prefix);
        const int nt = __popcll(__ballot(take));
        const bool def = live && !take;
        const u64 dmask = __ballot(def);
        if (def) pb[nbp + __popcll(dmask & lt)] = (unsigned)c;
        const int nd = __popcll(dmask);
        nbp += nd; ndef += (u64)nd; out += nt; fill += nt;
        __syncthreads();
    }
    if (lane == 0) {
        nbatch[tile] = (unsigned)nb;
        if (ndef) atomicAdd(deferred_total, ndef);
    }
}
__global__ void acc_scatter_kernel(int64_t nz, const unsigned* __restrict__ dst, const unsigned* __restrict__ word, const unsigned* __restrict__ perm,
                                   const double* __restrict__ val, unsigned* __restrict__ pack, double* __restrict__ out_val) {
    IPXK_GS(e, nz) {
        const unsigned d = dst[e];
        pack[d] = word[e];
        out_val[d] = val[perm[e]];
    }
}
// exclusive scan of the tiles' batch counts by one workgroup (a few thousand tiles)
__global__ __launch_bounds__(1024) void scan_u32_kernel(int n, const unsigned* __restrict__ in, unsigned* __restrict__ out) {
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const unsigned v = i < n ? in[i] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned before = carry;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (i < n) out[i] = before + incl - v;
        __syncthreads();
        if (tid == 1023) carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}
__global__ void acc_bptr_kernel(int ntiles, const unsigned* __restrict__ tile_ptr, const unsigned* __restrict__ tile_batch,
                                const unsigned* __restrict__ bstart, unsigned nz, unsigned* __restrict__ bptr) {
    const int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const unsigned e0 = tile_ptr[tile], b0 = tile_batch[tile], nb = tile_batch[tile + 1] - b0;
    for (unsigned q = threadIdx.x; q < nb; q += blockDim.x) bptr[b0 + q] = bstart[e0 + q];
    if (tile == ntiles - 1 && threadIdx.x == 0) bptr[tile_batch[ntiles]] = nz;
}

// ---- one-slice (FUSED) forms: a tile is a row block, offsets are relative to the tile's smallest gathered index ----
// smallest / largest gathered index per tile, the rows' count bytes (optional), a flag for rows with descending / repeated indices
__global__ void tile_window_kernel(int nrows, const int* __restrict__ ptr, const int* __restrict__ idx, int RB, int* __restrict__ lo,
                                   int* __restrict__ hi, unsigned char* __restrict__ cnt, int* flags) {
    IPXK_GS(r, nrows) {
        const int p0 = ptr[r], p1 = ptr[r + 1];
        if (cnt) { if (p1 - p0 > 255) flags[0] = 1; cnt[r] = (unsigned char)(p1 - p0); }
        if (p1 == p0) continue;
        int a = idx[p0], b = a;
        for (int p = p0 + 1; p < p1; p++) {
            const int v = idx[p];
            if (v <= idx[p - 1]) flags[1] = 1;                   // not ascending
            a = min(a, v); b = max(b, v);
        }
        atomicMin(lo + r / RB, a);
        atomicMax(hi + r / RB, b);
    }
}
// [0] widest window (hi - lo) of a tile, [1] most entries of a tile; empty tiles get lo = 0
__global__ void tile_window_stats_kernel(int nrb, int nrows, int RB, const int* __restrict__ ptr, int* __restrict__ lo, const int* __restrict__ hi,
                                         int* out) {
    IPXK_GS(t, nrb) {
        const int r0 = (int)t * RB, r1 = min(nrows, r0 + RB);
        const int ne = ptr[r1] - ptr[r0];
        if (ne == 0) { lo[t] = 0; continue; }
        atomicMax(out + 0, hi[t] - lo[t]);
        atomicMax(out + 1, ne);
    }
}
// key = tile << 18 | (index - the tile's smallest index), enumerated in storage order
__global__ void fused_keys_kernel(int nrows, const int* __restrict__ ptr, const int* __restrict__ idx, int RB, const int* __restrict__ lo,
                                  u64* __restrict__ key, unsigned* __restrict__ pos, int* __restrict__ rowof) {
    IPXK_GS(r, nrows) {
        const u64 t = (u64)(r / RB);
        const int base = lo[r / RB];
        for (int p = ptr[r]; p < ptr[r + 1]; p++) {
            key[p] = (t << kSortedOffBits) | (u64)(unsigned)(idx[p] - base);
            pos[p] = (unsigned)p;
            if (rowof) rowof[p] = (int)r;
        }
    }
}
// sorted fused tiles: slot = place of the entry in the tile's row-major (= storage) order
__global__ void sorted_fused_fill_kernel(int64_t nz, const u64* __restrict__ keys, const unsigned* __restrict__ perm, const int* __restrict__ ptr, int RB,
                                         const double* __restrict__ val, unsigned* __restrict__ pack, double* __restrict__ out_val) {
    IPXK_GS(f, nz) {
        const u64 k = keys[f];
        const unsigned p = perm[f], t = (unsigned)(k >> kSortedOffBits);
        pack[f] = ((p - (unsigned)ptr[(size_t)t * RB]) << kSortedOffBits) | (unsigned)(k & ((1u << kSortedOffBits) - 1u));
        out_val[f] = val[p];
    }
}
__global__ void tile_ptr_from_rows_kernel(int nrb, int nrows, int RB, const int* __restrict__ ptr, unsigned* __restrict__ out) {
    IPXK_GS(t, (int64_t)nrb + 1) out[t] = (unsigned)ptr[min((int64_t)nrows, t * RB)];
}

struct Tmp {
    DevBuf<unsigned char> bytes;
    void* need(size_t n) { if (bytes.size() < n) bytes.resize(n); return bytes.get(); }
};
template <class K>
void sort_pairs(Tmp& T, const K* kin, K* kout, const unsigned* vin, unsigned* vout, size_t n, int bits, hipStream_t s) {
    size_t bytes = 0;
    IPXK_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, (unsigned)bits, s));
    IPXK_HIP(rocprim::radix_sort_pairs(T.need(bytes), bytes, kin, kout, vin, vout, n, 0u, (unsigned)bits, s));
}

}  // namespace

// scratch of the layout builders: kept for the two gather matrices of a model, released when the model is built
struct LayoutScratch {
    DevBuf<unsigned> k1, k2, v1, v2, v3, v4;
    DevBuf<u64> q1, q2;
    DevBuf<int> stats;
    Tmp T;
};
LayoutScratch* new_layout_scratch() { return new LayoutScratch; }
void free_layout_scratch(LayoutScratch* S) { delete S; }

// ---------------------------------------------------------------------------
// the model on the device: CSC and the row-wise copy, 32-bit indices
// ---------------------------------------------------------------------------
void upload_plain_model(Context* c, const ipxint* Ap, const ipxint* Ai, const double* Ax) {
    const int64_t m = c->m, n = c->n;
    hipStream_t s = c->stream;
    IPXK_REQUIRE(m < (int64_t(1) << 31) - 1 && n < (int64_t(1) << 31) - 1, "dimension exceeds 32-bit device indices");
    IPXK_REQUIRE(Ap[0] == 0, "colptr[0] must be 0");
    const int64_t nz = Ap[n];
    IPXK_REQUIRE(nz >= 0 && nz < (int64_t(1) << 31) - kLongSeg, "nnz exceeds 32-bit device indices");
    c->h_Ap.assign(Ap, Ap + n + 1);
    std::vector<int> ap32((size_t)n + 1);
    for (int64_t j = 0; j < n; j++) {
        IPXK_REQUIRE(Ap[j] <= Ap[j + 1], "colptr not monotone");
        ap32[(size_t)j] = (int)Ap[j];
    }
    ap32[(size_t)n] = (int)nz;
    c->nnz = nz;
    const size_t nz1 = (size_t)std::max<int64_t>(nz, 1);
    c->pl_Ap.upload(ap32, s);
    c->pl_Ai.ensure(nz1); c->pl_Ax.ensure(nz1); c->pl_Tp.ensure((size_t)m + 1); c->pl_Ti.ensure(nz1); c->pl_Tx.ensure(nz1);
    DevBuf<int> bad(1);
    IPXK_HIP(hipMemsetAsync(bad.get(), 0, sizeof(int), s));
    if (nz > 0) {
        DevBuf<ipxint> ai64(nz1);
        ai64.upload(Ai, (size_t)nz, s);
        c->pl_Ax.upload(Ax, (size_t)nz, s);
        hipLaunchKernelGGL(narrow_kernel, dim3(gridn(nz)), dim3(kBlock), 0, s, nz, ai64.get(), c->pl_Ai.get(), (int)m, bad.get());
        int flag = 0;
        IPXK_HIP(hipMemcpyAsync(&flag, bad.get(), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));           // ai64 goes out of scope
        IPXK_REQUIRE(flag == 0, "row index out of range");
        // Transpose (src/sparse_matrix.cc:120-151): a stable sort by row of the entries enumerated column by column
        // leaves every row in ascending source-column order, like the reference's counting sort
        DevBuf<int> colof(nz1);
        DevBuf<unsigned> pos(nz1), rows2(nz1), perm(nz1);
        Tmp T;
        hipLaunchKernelGGL(rowof_kernel, dim3(gridn(n)), dim3(kBlock), 0, s, (int)n, c->pl_Ap.get(), colof.get(), pos.get());
        sort_pairs<unsigned>(T, reinterpret_cast<const unsigned*>(c->pl_Ai.get()), rows2.get(), pos.get(), perm.get(), (size_t)nz,
                             bits_for((u64)std::max<int64_t>(m, 2) - 1), s);
        hipLaunchKernelGGL(gather_transposed_kernel, dim3(gridn(nz)), dim3(kBlock), 0, s, nz, perm.get(), colof.get(), c->pl_Ax.get(),
                           c->pl_Ti.get(), c->pl_Tx.get());
        hipLaunchKernelGGL(row_pointers_kernel, dim3(gridn(m + 1)), dim3(kBlock), 0, s, m, nz, rows2.get(), c->pl_Tp.get());
        IPXK_HIP(hipStreamSynchronize(s));           // temporaries go out of scope
    } else {
        IPXK_HIP(hipMemsetAsync(c->pl_Tp.get(), 0, ((size_t)m + 1) * sizeof(int), s));
        IPXK_HIP(hipStreamSynchronize(s));
    }
    IPXK_HIP(hipGetLastError());
    c->have_plain = true;
}

// host copies of the entries (64-bit indices) for the paths that still build on the host
void ensure_host_model(Context* c, bool rowwise) {
    hipStream_t s = c->stream;
    const size_t nz = (size_t)c->nnz;
    auto fetch = [&](const DevBuf<int>& ptr, size_t nptr, const DevBuf<int>& idx, const DevBuf<double>& val, std::vector<ipxint>& hp,
                     std::vector<ipxint>& hi, std::vector<double>& hx) {
        DevBuf<ipxint> wide(std::max(std::max(nz, nptr), (size_t)1));
        hp.resize(nptr); hi.resize(nz); hx.resize(nz);
        hipLaunchKernelGGL(widen_kernel, dim3(gridn((int64_t)nptr)), dim3(kBlock), 0, s, (int64_t)nptr, ptr.get(), wide.get());
        wide.download(hp.data(), nptr, s);
        if (nz) {
            hipLaunchKernelGGL(widen_kernel, dim3(gridn((int64_t)nz)), dim3(kBlock), 0, s, (int64_t)nz, idx.get(), wide.get());
            wide.download(hi.data(), nz, s);
            val.download(hx.data(), nz, s);
        }
        IPXK_HIP(hipStreamSynchronize(s));
    };
    IPXK_REQUIRE(c->have_plain, "model not uploaded");
    if (!rowwise && c->h_Ai.size() != nz) {
        std::vector<ipxint> hp;
        fetch(c->pl_Ap, (size_t)c->n + 1, c->pl_Ai, c->pl_Ax, hp, c->h_Ai, c->h_Ax);
    }
    if (rowwise && (c->h_ATp.size() != (size_t)c->m + 1 || c->h_ATi.size() != nz))
        fetch(c->pl_Tp, (size_t)c->m + 1, c->pl_Ti, c->pl_Tx, c->h_ATp, c->h_ATi, c->h_ATx);
}

// the entries of `cols` (structural columns), one column after the other, on the host
void fetch_columns(Context* c, const std::vector<ipxint>& cols, std::vector<ipxint>& Cp, std::vector<ipxint>& Ci, std::vector<double>& Cx) {
    const int k = (int)cols.size();
    hipStream_t s = c->stream;
    Cp.assign((size_t)k + 1, 0);
    std::vector<int> c32((size_t)k), off((size_t)k);
    for (int kk = 0; kk < k; kk++) {
        const ipxint j = cols[(size_t)kk];
        c32[(size_t)kk] = (int)j;
        off[(size_t)kk] = (int)Cp[(size_t)kk];
        Cp[(size_t)kk + 1] = Cp[(size_t)kk] + (c->h_Ap[(size_t)j + 1] - c->h_Ap[(size_t)j]);
    }
    const size_t tot = (size_t)Cp[(size_t)k];
    Ci.resize(tot); Cx.resize(tot);
    if (k == 0 || tot == 0) return;
    DevBuf<int> dc, doff;
    DevBuf<ipxint> di(tot);
    DevBuf<double> dx(tot);
    dc.upload(c32, s); doff.upload(off, s);
    hipLaunchKernelGGL(gather_columns_kernel, dim3(k), dim3(kBlock), 0, s, k, dc.get(), doff.get(), c->pl_Ap.get(), c->pl_Ai.get(), c->pl_Ax.get(),
                       di.get(), dx.get());
    di.download(Ci.data(), tot, s);
    dx.download(Cx.data(), tot, s);
    IPXK_HIP(hipStreamSynchronize(s));
    IPXK_HIP(hipGetLastError());
}

int device_max_row_length(LayoutScratch& S, int nrows, const int* dptr, hipStream_t s) {
    S.stats.ensure(8);
    IPXK_HIP(hipMemsetAsync(S.stats.get(), 0, 8 * sizeof(int), s));
    hipLaunchKernelGGL(max_len_kernel, dim3(gridn(nrows)), dim3(kBlock), 0, s, nrows, dptr, S.stats.get());
    int h = 0;
    IPXK_HIP(hipMemcpyAsync(&h, S.stats.get(), sizeof(int), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    return h;
}

// ---------------------------------------------------------------------------
// Long rows (more than kMaxRowLen entries: dense columns of A in the row-wise copy, dense rows in the column-wise one) taken out of a
// row-wise matrix on the device: the flags, the segment arrays of the long-row kernels -- equal to GatherMatrix::build()'s -- and the
// matrix with those rows left empty, from which the tile layouts are then built as for a matrix without long rows.
// (Until round 5 a single long row sent the whole model to the host builders: 1.1 s per gather matrix at 15 M entries.)
// ---------------------------------------------------------------------------
namespace {
constexpr int kMaxLongRowsDevice = 1 << 16;
__global__ void long_flag_kernel(int nrows, const int* __restrict__ ptr, unsigned char* __restrict__ flag, int* __restrict__ slen, int* counters,
                                 int* __restrict__ list, int cap) {
    IPXK_GS(r, (int64_t)nrows + 1) {
        if (r == nrows) { slen[r] = 0; continue; }
        const int len = ptr[r + 1] - ptr[r];
        const bool lg = len > kMaxRowLen;
        flag[r] = lg ? 1 : 0;
        slen[r] = lg ? 0 : len;
        if (lg) {
            const int k = atomicAdd(counters, 1);
            if (k < cap) list[k] = (int)r;
        }
    }
}
__global__ void strip_copy_kernel(int nrows, const int* __restrict__ ptr, const int* __restrict__ sptr, const unsigned char* __restrict__ flag,
                                  const int* __restrict__ idx, const double* __restrict__ val, int* __restrict__ oidx, double* __restrict__ oval) {
    // 8 lanes per row: a wavefront copies 8 rows, consecutive entries by consecutive lanes
    const int g = threadIdx.x & 7;
    for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; r < nrows; r += ((int64_t)gridDim.x * blockDim.x) >> 3) {
        if (flag[r]) continue;
        const int p0 = ptr[r], len = ptr[r + 1] - p0, q0 = sptr[r];
        for (int t = g; t < len; t += 8) { oidx[q0 + t] = idx[p0 + t]; oval[q0 + t] = val[p0 + t]; }
    }
}
__global__ void long_copy_kernel(const int* __restrict__ lrow, const int* __restrict__ loff, const int* __restrict__ ptr, const int* __restrict__ idx,
                                 const double* __restrict__ val, int* __restrict__ lidx, double* __restrict__ lval) {
    const int l = blockIdx.x, r = lrow[l], p0 = ptr[r], len = ptr[r + 1] - p0, q0 = loff[l];
    for (int t = threadIdx.x; t < len; t += blockDim.x) { lidx[q0 + t] = idx[p0 + t]; lval[q0 + t] = val[p0 + t]; }
}
}  // namespace

bool device_strip_long_rows(LayoutScratch& S, GatherMatrix& G, int nrows, const int* dptr, const int* didx, const double* dval,
                            DevBuf<int>& sptr, DevBuf<int>& sidx, DevBuf<double>& sval, int64_t* nnz_short, hipStream_t s) {
    S.stats.ensure(8);
    IPXK_HIP(hipMemsetAsync(S.stats.get(), 0, 8 * sizeof(int), s));
    DevBuf<int> slen((size_t)nrows + 1), list((size_t)kMaxLongRowsDevice);
    G.row_long.ensure((size_t)nrows);
    hipLaunchKernelGGL(long_flag_kernel, dim3(gridn((int64_t)nrows + 1)), dim3(kBlock), 0, s, nrows, dptr, G.row_long.get(), slen.get(), S.stats.get(),
                       list.get(), kMaxLongRowsDevice);
    sptr.ensure((size_t)nrows + 1);
    {
        size_t bytes = 0;
        IPXK_HIP(rocprim::exclusive_scan(nullptr, bytes, slen.get(), sptr.get(), 0, (size_t)nrows + 1, rocprim::plus<int>(), s));
        IPXK_HIP(rocprim::exclusive_scan(S.T.need(bytes), bytes, slen.get(), sptr.get(), 0, (size_t)nrows + 1, rocprim::plus<int>(), s));
    }
    int nl = 0, ns_total = 0;
    IPXK_HIP(hipMemcpyAsync(&nl, S.stats.get(), sizeof(int), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipMemcpyAsync(&ns_total, sptr.get() + nrows, sizeof(int), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    if (nl <= 0 || nl > kMaxLongRowsDevice) return false;
    std::vector<int> lrow((size_t)nl);
    IPXK_HIP(hipMemcpyAsync(lrow.data(), list.get(), (size_t)nl * sizeof(int), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    std::sort(lrow.begin(), lrow.end());                       // (the atomic appended them in no particular order)
    // their extents: two words per long row
    DevBuf<int> dl((size_t)nl), ext((size_t)2 * nl);
    dl.upload(lrow, s);
    hipLaunchKernelGGL(row_extent_kernel, dim3(gridn(nl)), dim3(kBlock), 0, s, nl, dl.get(), dptr, ext.get());
    std::vector<int> he((size_t)2 * nl);
    ext.download(he.data(), he.size(), s);
    IPXK_HIP(hipStreamSynchronize(s));
    std::vector<int> sp0, sp1, lslot, loff((size_t)nl);
    int64_t nlong_entries = 0;
    for (int l = 0; l < nl; l++) {
        const int len = he[2 * l + 1] - he[2 * l];
        loff[l] = (int)nlong_entries;
        lslot.push_back((int)sp0.size());
        for (int q0 = 0; q0 < len; q0 += kLongSeg) {
            sp0.push_back((int)(nlong_entries + q0));
            sp1.push_back((int)(nlong_entries + std::min(q0 + kLongSeg, len)));
        }
        nlong_entries += len;
    }
    lslot.push_back((int)sp0.size());
    G.nlong = nl;
    G.nseg = (int)sp0.size();
    G.seg_p0.upload(sp0, s); G.seg_p1.upload(sp1, s); G.long_row.upload(lrow, s); G.long_slot.upload(lslot, s);
    G.lidx.ensure((size_t)nlong_entries); G.lval.ensure((size_t)nlong_entries);
    DevBuf<int> doff((size_t)nl);
    doff.upload(loff, s);
    hipLaunchKernelGGL(long_copy_kernel, dim3(nl), dim3(kBlock), 0, s, dl.get(), doff.get(), dptr, didx, dval, G.lidx.get(), G.lval.get());
    G.long_partials.resize((size_t)std::max(G.nseg, 1));
    G.h_row_long.assign((size_t)nrows, 0);
    for (int r : lrow) G.h_row_long[(size_t)r] = 1;
    sidx.ensure((size_t)std::max(ns_total, 1)); sval.ensure((size_t)std::max(ns_total, 1));
    hipLaunchKernelGGL(strip_copy_kernel, dim3(gridn((int64_t)nrows * 8)), dim3(kBlock), 0, s, nrows, dptr, sptr.get(), G.row_long.get(), didx, dval,
                       sidx.get(), sval.get());
    IPXK_HIP(hipStreamSynchronize(s));                          // the host vectors and temporaries go out of scope
    IPXK_HIP(hipGetLastError());
    *nnz_short = ns_total;
    return true;
}

// ---------------------------------------------------------------------------
// XCD-sliced tiles (the arrays of GatherMatrix::build_sliced with ns_request = 0, bit for bit)
// ---------------------------------------------------------------------------
// returns false when the layout does not apply (x fits an XCD's L2, a tile does not fit LDS, > 255 entries of a row in
// one slice): the caller then takes the host path
bool device_build_sliced(LayoutScratch& S, SlicedMatrix& out, int nrows, int ncols, int64_t nnz, const int* dptr, const int* didx,
                         const double* dval, hipStream_t s, int ns_request) {
    out = SlicedMatrix();
    if (nrows == 0 || nnz == 0 || ncols == 0) return false;
    int ns = 1;                                   // ns_request == 1: the fused tiles (one slice)
    if (ns_request != 1) {
        const int64_t x_bytes = (int64_t)ncols * 8;
        int64_t slice_bytes = int64_t(2) << 20;
        if (const char* e = getenv("IPXK_SLICE_TEST_KB"))
            if (atoi(e) > 0) slice_bytes = (int64_t)atoi(e) << 10;
        if (x_bytes <= 2 * slice_bytes && !(getenv("IPXK_SLICE_FORCE2") && x_bytes > slice_bytes)) return false;
        ns = 2;
        while (ns < 8 && x_bytes > (int64_t)ns * slice_bytes) ns *= 2;
    }
    const int64_t slice = (((int64_t)ncols + ns - 1) / ns + 15) / 16 * 16;
    int R = kSlicedRows;
    while (R > kBlock && ((int64_t)nrows + R - 1) / R * (int64_t)ns < 2048) R /= 2;
    const size_t nz = (size_t)nnz;
    S.k1.ensure(nz); S.k2.ensure(nz); S.v1.ensure(nz); S.v2.ensure(nz); S.stats.ensure(8);
    int nrb = 0, h[4] = {0, 0, 0, 0};
    int64_t ntiles = 0;
    for (;; R /= 2) {
        if (R < kBlock) return false;
        nrb = (nrows + R - 1) / R;
        ntiles = (int64_t)nrb * ns;
        if ((u64)ntiles * (u64)R >= (u64(1) << 32)) return false;
        hipLaunchKernelGGL(sliced_keys_kernel, dim3(gridn(nrows)), dim3(kBlock), 0, s, nrows, dptr, didx, R, ns, (int)slice, S.k1.get(), S.v1.get());
        sort_pairs<unsigned>(S.T, S.k1.get(), S.k2.get(), S.v1.get(), S.v2.get(), nz, bits_for((u64)ntiles * (u64)R - 1), s);
        out.tile_ptr.ensure((size_t)ntiles + 1);
        hipLaunchKernelGGL(lower_bounds_kernel<unsigned>, dim3(gridn(ntiles + 1)), dim3(kBlock), 0, s, ntiles + 1, nnz, S.k2.get(), (u64)R, 0,
                           out.tile_ptr.get());
        IPXK_HIP(hipMemsetAsync(S.stats.get(), 0, 8 * sizeof(int), s));
        hipLaunchKernelGGL(tile_stats_kernel, dim3(gridn(nrb)), dim3(kBlock), 0, s, nrb, ns, out.tile_ptr.get(), S.stats.get());
        IPXK_HIP(hipMemcpyAsync(h, S.stats.get(), sizeof h, hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        if (h[0] <= kSlicedMaxTile) break;
    }
    u64 dom = 0;
    memcpy(&dom, h + 2, sizeof dom);
    out.dominant_fraction = (double)(int64_t)dom / (double)nnz;
    const size_t nslots = (size_t)ntiles * R;
    out.cnt.ensure(nslots); out.idx.ensure(nz); out.val.ensure(nz);
    IPXK_HIP(hipMemsetAsync(out.cnt.get(), 0, nslots, s));
    IPXK_HIP(hipMemsetAsync(S.stats.get(), 0, 8 * sizeof(int), s));
    hipLaunchKernelGGL(run_counts_kernel, dim3(gridn(nnz)), dim3(kBlock), 0, s, nnz, S.k2.get(), out.cnt.get(), S.stats.get());
    hipLaunchKernelGGL(gather_entries_kernel, dim3(gridn(nnz)), dim3(kBlock), 0, s, nnz, S.v2.get(), didx, dval, out.idx.get(), out.val.get());
    int over = 0;
    IPXK_HIP(hipMemcpyAsync(&over, S.stats.get(), sizeof(int), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    IPXK_HIP(hipGetLastError());
    if (over) { out = SlicedMatrix(); return false; }
    out.R = R; out.nslices = ns; out.nrb = nrb; out.nrows_pad = nrb * R; out.max_tile = h[0];
    out.partial.resize(ns > 1 ? (size_t)ns * out.nrows_pad : 1);
    out.built = true;
    return true;
}

// ---------------------------------------------------------------------------
// sorted sub-tiles (the arrays of GatherMatrix::build_sorted, bit for bit); needs the sliced layout's slices
// ---------------------------------------------------------------------------
bool device_build_sorted(LayoutScratch& S, SortedMatrix& out, const SlicedMatrix& sliced, int nrows, int ncols, int64_t nnz, const int* dptr,
                         const int* didx, const double* dval, hipStream_t s) {
    out = SortedMatrix();
    if (!sliced.built || sliced.nslices < 2 || nnz == 0) return false;
    const int ns = sliced.nslices;
    const int64_t slice = (((int64_t)ncols + ns - 1) / ns + 15) / 16 * 16;
    if (slice > (int64_t(1) << kSortedOffBits)) return false;
    static const int nsub_env = [] { const char* e = getenv("IPXK_SORTED_NSUB"); return e && atoi(e) > 0 ? std::min(atoi(e), 16) : 2; }();
    const int nsub = nsub_env;
    const int64_t half = (slice / nsub + 15) / 16 * 16;
    const size_t nz = (size_t)nnz;
    S.k1.ensure(nz); S.k2.ensure(nz); S.v1.ensure(nz); S.v2.ensure(nz); S.v3.ensure(nz); S.v4.ensure(nz); S.q1.ensure(nz); S.q2.ensure(nz);
    S.stats.ensure(8);
    int RB = 32 * kSortedThreads, nrb = 0, max_sub = 0;
    int64_t nsubs = 0;
    for (;; RB /= 2) {
        if (RB < 4 * kSortedThreads) return false;
        nrb = (nrows + RB - 1) / RB;
        nsubs = (int64_t)nrb * ns * nsub;
        if ((u64)nsubs * (u64)RB >= (u64(1) << 32)) return false;
        hipLaunchKernelGGL(sorted_keys1_kernel, dim3(gridn(nrows)), dim3(kBlock), 0, s, nrows, dptr, didx, RB, ns, nsub, (int)slice, (int)half,
                           S.k1.get(), S.v1.get());
        sort_pairs<unsigned>(S.T, S.k1.get(), S.k2.get(), S.v1.get(), S.v2.get(), nz, bits_for((u64)nsubs * (u64)RB - 1), s);
        out.sub_ptr.ensure((size_t)nsubs + 1);
        hipLaunchKernelGGL(lower_bounds_kernel<unsigned>, dim3(gridn(nsubs + 1)), dim3(kBlock), 0, s, nsubs + 1, nnz, S.k2.get(), (u64)RB, 0,
                           out.sub_ptr.get());
        IPXK_HIP(hipMemsetAsync(S.stats.get(), 0, 8 * sizeof(int), s));
        hipLaunchKernelGGL(max_range_kernel, dim3(gridn(nsubs)), dim3(kBlock), 0, s, nsubs, out.sub_ptr.get(), S.stats.get());
        IPXK_HIP(hipMemcpyAsync(&max_sub, S.stats.get(), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        if (max_sub <= kSortedMaxSub) break;
    }
    const size_t nslots = (size_t)nsubs * RB;
    out.cnt.ensure(nslots); out.pack.ensure(nz); out.val.ensure(nz);
    IPXK_HIP(hipMemsetAsync(out.cnt.get(), 0, nslots, s));
    IPXK_HIP(hipMemsetAsync(S.stats.get(), 0, 8 * sizeof(int), s));
    hipLaunchKernelGGL(run_counts_kernel, dim3(gridn(nnz)), dim3(kBlock), 0, s, nnz, S.k2.get(), out.cnt.get(), S.stats.get());
    // second sort: by (sub-tile, offset), stable on the slot order
    hipLaunchKernelGGL(sorted_keys2_kernel, dim3(gridn(nnz)), dim3(kBlock), 0, s, nnz, S.k2.get(), S.v2.get(), didx, RB, ns, nsub, (int)slice,
                       S.q1.get(), S.v3.get());
    sort_pairs<u64>(S.T, S.q1.get(), S.q2.get(), S.v3.get(), S.v4.get(), nz, kSortedOffBits + bits_for((u64)nsubs - 1), s);
    hipLaunchKernelGGL(sorted_fill_kernel, dim3(gridn(nnz)), dim3(kBlock), 0, s, nnz, S.q2.get(), S.v4.get(), S.v2.get(), out.sub_ptr.get(), dval,
                       out.pack.get(), out.val.get());
    int over = 0;
    IPXK_HIP(hipMemcpyAsync(&over, S.stats.get(), sizeof(int), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    IPXK_HIP(hipGetLastError());
    if (over) { out = SortedMatrix(); return false; }
    out.nslices = ns; out.nsub = nsub; out.nrb = nrb; out.RB = RB; out.nrows_pad = nrb * RB;
    out.max_sub = max_sub; out.slice_elems = (int)slice;
    out.partial.resize((size_t)ns * out.nrows_pad);
    out.built = true;
    return true;
}

// ---------------------------------------------------------------------------
// accumulated tiles (the arrays of GatherMatrix::build_acc, bit for bit); needs the sliced layout's slices
// ---------------------------------------------------------------------------
int acc_rows_per_block(int nrows, int ns) {
    static const int cap = [] { const char* e = getenv("IPXK_ACC_ROWS"); return e && atoi(e) >= 1024 ? std::min(atoi(e), kAccMaxRows) : kAccMaxRows; }();
    int RB = cap;
    // (at least one tile per CU; halving the row block halves the entries per line of the slice and doubles the waiting entries:
    // 1M x 2M with 8192 rows: 107 us per pass, with 16384: 94)
    while (RB > 1024 && ((int64_t)nrows + RB - 1) / RB * (int64_t)ns < 256) RB /= 2;
    return RB;
}

bool device_build_acc(LayoutScratch& S, AccMatrix& out, const SlicedMatrix& sliced, int nrows, int ncols, int64_t nnz, const int* dptr,
                      const int* didx, const double* dval, hipStream_t s) {
    out = AccMatrix();
    if (!sliced.built || sliced.nslices < 2 || nnz == 0) return false;
    const int ns = sliced.nslices;
    const int64_t slice = (((int64_t)ncols + ns - 1) / ns + 15) / 16 * 16;
    if (slice > (int64_t(1) << kSortedOffBits)) return false;
    const int RB = acc_rows_per_block(nrows, ns);
    const int nrb = (nrows + RB - 1) / RB;
    const int64_t ntiles = (int64_t)nrb * ns;
    const size_t nz = (size_t)nnz;
    S.q1.ensure(nz); S.q2.ensure(nz); S.v1.ensure(nz); S.v2.ensure(nz); S.v3.ensure(nz); S.v4.ensure(nz); S.k1.ensure(nz); S.k2.ensure(nz);
    S.stats.ensure(8);
    DevBuf<int> rowof(nz);
    DevBuf<unsigned> tile_ptr((size_t)ntiles + 1), nbatch((size_t)ntiles + 1), bstart(nz);
    hipLaunchKernelGGL(acc_keys_kernel, dim3(gridn(nrows)), dim3(kBlock), 0, s, nrows, dptr, didx, RB, ns, (int)slice, S.q1.get(), S.v1.get(), rowof.get());
    sort_pairs<u64>(S.T, S.q1.get(), S.q2.get(), S.v1.get(), S.v2.get(), nz, kSortedOffBits + bits_for((u64)std::max<int64_t>(ntiles, 2) - 1), s);
    hipLaunchKernelGGL(lower_bounds_kernel<u64>, dim3(gridn(ntiles + 1)), dim3(kBlock), 0, s, ntiles + 1, nnz, S.q2.get(), (u64)1, kSortedOffBits,
                       tile_ptr.get());
    unsigned* word = S.k1.get();
    hipLaunchKernelGGL(acc_words_kernel, dim3(gridn(nnz)), dim3(kBlock), 0, s, nnz, S.q2.get(), S.v2.get(), rowof.get(), RB, word);
    IPXK_HIP(hipMemsetAsync(S.stats.get(), 0, 8 * sizeof(int), s));
    const size_t lds = (size_t)RB * 2 * sizeof(unsigned);
    IPXK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(acc_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(kAccMaxRows * 2 * sizeof(unsigned))));
    hipLaunchKernelGGL(acc_batch_kernel, dim3((unsigned)ntiles), dim3(64), lds, s, RB, tile_ptr.get(), word, S.k2.get(), S.v3.get(), S.v4.get(),
                       bstart.get(), nbatch.get(), reinterpret_cast<u64*>(S.stats.get()));
    out.tile_batch.ensure((size_t)ntiles + 1);
    hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(1024), 0, s, (int)ntiles, nbatch.get(), out.tile_batch.get());
    unsigned nb_total = 0;
    u64 ndef = 0;
    IPXK_HIP(hipMemcpyAsync(&nb_total, out.tile_batch.get() + ntiles, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipMemcpyAsync(&ndef, S.stats.get(), sizeof(u64), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    out.bptr.ensure((size_t)nb_total + 1); out.pack.ensure(nz); out.val.ensure(nz);
    hipLaunchKernelGGL(acc_bptr_kernel, dim3((unsigned)ntiles), dim3(64), 0, s, (int)ntiles, tile_ptr.get(), out.tile_batch.get(), bstart.get(),
                       (unsigned)nnz, out.bptr.get());
    hipLaunchKernelGGL(acc_scatter_kernel, dim3(gridn(nnz)), dim3(kBlock), 0, s, nnz, S.k2.get(), word, S.v2.get(), dval, out.pack.get(),
                       out.val.get());
    IPXK_HIP(hipStreamSynchronize(s));       // the temporaries go out of scope
    IPXK_HIP(hipGetLastError());
    out.nslices = ns; out.nrb = nrb; out.RB = RB; out.nrows_pad = nrb * RB; out.slice_elems = (int)slice;
    out.nbatches = nb_total; out.deferred = (int64_t)ndef;
    out.partial.resize((size_t)ns * out.nrows_pad);
    out.built = true;
    return true;
}

// ---------------------------------------------------------------------------
// FUSED sorted tiles (the arrays of GatherMatrix::build_sorted_fused, bit for bit)
// ---------------------------------------------------------------------------
bool device_build_sorted_fused(LayoutScratch& S, SortedMatrix& out, int nrows, int ncols, int64_t nnz, const int* dptr, const int* didx,
                               const double* dval, hipStream_t s) {
    out = SortedMatrix();
    if (nrows == 0 || nnz == 0 || ncols == 0) return false;
    static const int cap = [] { const char* e = getenv("IPXK_SF_MAXSUB"); return e && atoi(e) >= 256 ? std::min(atoi(e), kSortedMaxSub) : kSortedMaxSub; }();
    const size_t nz = (size_t)nnz;
    S.q1.ensure(nz); S.q2.ensure(nz); S.v1.ensure(nz); S.v2.ensure(nz); S.stats.ensure(8);
    DevBuf<int> lo, hi;
    int RB = 32 * kSortedThreads, nrb = 0, h[4] = {0, 0, 0, 0};
    for (;; RB /= 2) {
        if (RB < kSortedThreads) return false;
        nrb = (nrows + RB - 1) / RB;
        if (RB > kSortedThreads && nrb < 1024) continue;          // enough tiles to fill the chip
        lo.ensure((size_t)nrb); hi.ensure((size_t)nrb);
        out.cnt.ensure((size_t)nrb * RB);
        IPXK_HIP(hipMemsetAsync(lo.get(), 0x7f, (size_t)nrb * sizeof(int), s));
        IPXK_HIP(hipMemsetAsync(hi.get(), 0xff, (size_t)nrb * sizeof(int), s));
        IPXK_HIP(hipMemsetAsync(out.cnt.get(), 0, (size_t)nrb * RB, s));
        IPXK_HIP(hipMemsetAsync(S.stats.get(), 0, 8 * sizeof(int), s));
        hipLaunchKernelGGL(tile_window_kernel, dim3(gridn(nrows)), dim3(kBlock), 0, s, nrows, dptr, didx, RB, lo.get(), hi.get(), out.cnt.get(), S.stats.get() + 2);
        hipLaunchKernelGGL(tile_window_stats_kernel, dim3(gridn(nrb)), dim3(kBlock), 0, s, nrb, nrows, RB, dptr, lo.get(), hi.get(), S.stats.get());
        IPXK_HIP(hipMemcpyAsync(h, S.stats.get(), sizeof h, hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        if (h[2]) { out = SortedMatrix(); return false; }          // a row of more than 255 entries
        if (h[1] <= cap) break;
    }
    if (h[0] >= (1 << kSortedOffBits)) { out = SortedMatrix(); return false; }     // a tile's window of x is too wide: no locality to use
    hipLaunchKernelGGL(fused_keys_kernel, dim3(gridn(nrows)), dim3(kBlock), 0, s, nrows, dptr, didx, RB, lo.get(), S.q1.get(), S.v1.get(), (int*)nullptr);
    sort_pairs<u64>(S.T, S.q1.get(), S.q2.get(), S.v1.get(), S.v2.get(), nz, kSortedOffBits + bits_for((u64)std::max(nrb, 2) - 1), s);
    out.sub_ptr.ensure((size_t)nrb + 1); out.pack.ensure(nz); out.val.ensure(nz); out.xmin.ensure((size_t)nrb);
    hipLaunchKernelGGL(tile_ptr_from_rows_kernel, dim3(gridn(nrb + 1)), dim3(kBlock), 0, s, nrb, nrows, RB, dptr, out.sub_ptr.get());
    hipLaunchKernelGGL(sorted_fused_fill_kernel, dim3(gridn(nnz)), dim3(kBlock), 0, s, nnz, S.q2.get(), S.v2.get(), dptr, RB, dval, out.pack.get(), out.val.get());
    IPXK_HIP(hipMemcpyAsync(out.xmin.get(), lo.get(), (size_t)nrb * sizeof(int), hipMemcpyDeviceToDevice, s));
    IPXK_HIP(hipStreamSynchronize(s));
    IPXK_HIP(hipGetLastError());
    out.nslices = 1; out.nsub = 1; out.nrb = nrb; out.RB = RB; out.nrows_pad = nrb * RB; out.max_sub = h[1]; out.slice_elems = 0; out.fused = true;
    out.built = true;
    return true;
}

// ---------------------------------------------------------------------------
// FUSED accumulated tiles (the arrays of GatherMatrix::build_acc_fused, bit for bit)
// ---------------------------------------------------------------------------
bool device_build_acc_fused(LayoutScratch& S, AccMatrix& out, int nrows, int ncols, int64_t nnz, const int* dptr, const int* didx,
                            const double* dval, hipStream_t s) {
    out = AccMatrix();
    if (nrows == 0 || nnz == 0 || ncols == 0) return false;
    int RB = kAccBatch;
    while (((int64_t)nrows + RB - 1) / RB > kMaxPartials) RB *= 2;
    if (RB > kAccMaxRows) return false;
    const int nrb = (nrows + RB - 1) / RB;
    const size_t nz = (size_t)nnz;
    S.q1.ensure(nz); S.q2.ensure(nz); S.v1.ensure(nz); S.v2.ensure(nz); S.v3.ensure(nz); S.v4.ensure(nz); S.k1.ensure(nz); S.k2.ensure(nz);
    S.stats.ensure(8);
    DevBuf<int> lo((size_t)nrb), hi((size_t)nrb), rowof(nz);
    DevBuf<unsigned> tile_ptr((size_t)nrb + 1), nbatch((size_t)nrb + 1), bstart(nz);
    IPXK_HIP(hipMemsetAsync(lo.get(), 0x7f, (size_t)nrb * sizeof(int), s));
    IPXK_HIP(hipMemsetAsync(hi.get(), 0xff, (size_t)nrb * sizeof(int), s));
    IPXK_HIP(hipMemsetAsync(S.stats.get(), 0, 8 * sizeof(int), s));
    hipLaunchKernelGGL(tile_window_kernel, dim3(gridn(nrows)), dim3(kBlock), 0, s, nrows, dptr, didx, RB, lo.get(), hi.get(), (unsigned char*)nullptr,
                       S.stats.get() + 2);
    hipLaunchKernelGGL(tile_window_stats_kernel, dim3(gridn(nrb)), dim3(kBlock), 0, s, nrb, nrows, RB, dptr, lo.get(), hi.get(), S.stats.get());
    int h[4] = {0, 0, 0, 0};
    IPXK_HIP(hipMemcpyAsync(h, S.stats.get(), sizeof h, hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    if (h[3]) return false;                                        // a row with descending indices: the sum would not be in storage order
    if (h[0] >= (1 << kSortedOffBits)) return false;               // a tile's window of x is too wide
    hipLaunchKernelGGL(fused_keys_kernel, dim3(gridn(nrows)), dim3(kBlock), 0, s, nrows, dptr, didx, RB, lo.get(), S.q1.get(), S.v1.get(), rowof.get());
    sort_pairs<u64>(S.T, S.q1.get(), S.q2.get(), S.v1.get(), S.v2.get(), nz, kSortedOffBits + bits_for((u64)std::max(nrb, 2) - 1), s);
    hipLaunchKernelGGL(tile_ptr_from_rows_kernel, dim3(gridn(nrb + 1)), dim3(kBlock), 0, s, nrb, nrows, RB, dptr, tile_ptr.get());
    unsigned* word = S.k1.get();
    hipLaunchKernelGGL(acc_words_kernel, dim3(gridn(nnz)), dim3(kBlock), 0, s, nnz, S.q2.get(), S.v2.get(), rowof.get(), RB, word);
    IPXK_HIP(hipMemsetAsync(S.stats.get(), 0, 8 * sizeof(int), s));
    IPXK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(acc_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(kAccMaxRows * 2 * sizeof(unsigned))));
    hipLaunchKernelGGL(acc_batch_kernel, dim3((unsigned)nrb), dim3(64), (size_t)RB * 2 * sizeof(unsigned), s, RB, tile_ptr.get(), word, S.k2.get(),
                       S.v3.get(), S.v4.get(), bstart.get(), nbatch.get(), reinterpret_cast<u64*>(S.stats.get()));
    out.tile_batch.ensure((size_t)nrb + 1);
    hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(1024), 0, s, nrb, nbatch.get(), out.tile_batch.get());
    unsigned nb_total = 0;
    u64 ndef = 0;
    IPXK_HIP(hipMemcpyAsync(&nb_total, out.tile_batch.get() + nrb, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipMemcpyAsync(&ndef, S.stats.get(), sizeof(u64), hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    out.bptr.ensure((size_t)nb_total + 1); out.pack.ensure(nz); out.val.ensure(nz); out.xmin.ensure((size_t)nrb);
    hipLaunchKernelGGL(acc_bptr_kernel, dim3((unsigned)nrb), dim3(64), 0, s, nrb, tile_ptr.get(), out.tile_batch.get(), bstart.get(), (unsigned)nnz,
                       out.bptr.get());
    hipLaunchKernelGGL(acc_scatter_kernel, dim3(gridn(nnz)), dim3(kBlock), 0, s, nnz, S.k2.get(), word, S.v2.get(), dval, out.pack.get(), out.val.get());
    IPXK_HIP(hipMemcpyAsync(out.xmin.get(), lo.get(), (size_t)nrb * sizeof(int), hipMemcpyDeviceToDevice, s));
    IPXK_HIP(hipStreamSynchronize(s));
    IPXK_HIP(hipGetLastError());
    out.nslices = 1; out.nrb = nrb; out.RB = RB; out.nrows_pad = nrb * RB; out.slice_elems = 0; out.fused = true;
    out.nbatches = nb_total; out.deferred = (int64_t)ndef;
    out.built = true;
    return true;
}


// ---------------------------------------------------------------------------------------------------------------------------
// A renumbering of rows and columns that recovers locality (round 5).
// The SpMV of a matrix whose gathers have locality runs at 0.53 of the HBM roof, that of a uniformly random one at 0.31 (the x gathers
// miss the XCD's L2).  LPs that HAVE structure often arrive with it hidden -- rows and columns in the order a modelling tool emitted
// them.  A pure permutation brings it back: breadth-first levels of the bipartite graph rows <-> columns from a pseudo-peripheral row
// (Cuthill-McKee without the degree sort: two passes, the second from the smallest row of the first one's last level; further components
// after the first, up to kMaxComponents), rows and columns numbered by (level, old index) with one radix sort each.  The levels of a
// breadth-first search do not depend on which thread wins a race, so the numbering is deterministic.  A second copy of the model in
// the new numbering gets its own gather layouts (the same builders), both copies' products are timed, and the copy is used -- by the
// CR loop of the diag path, kkt_diag.hip -- only if it is at least 10 % faster.  A matrix without structure is recognised early (half
// of the rows reached within 8 levels: an expander) and costs a millisecond.  IPXK_REORDER=0: never; =1: keep the copy whatever the
// timing says (tests).
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxComponents = 64;
__global__ void bfs_expand_kernel(int nf, const int* __restrict__ frontier, const int* __restrict__ ptr, const int* __restrict__ idx,
                                  int* level_of, int level, int* __restrict__ next, int* next_count) {
    IPXK_GS(t, nf) {
        const int v = frontier[t];
        for (int p = ptr[v]; p < ptr[v + 1]; p++) {
            const int w = idx[p];
            if (level_of[w] < 0 && atomicCAS(&level_of[w], -1, level) == -1) next[atomicAdd(next_count, 1)] = w;
        }
    }
}
__global__ void first_unvisited_kernel(int n, const int* __restrict__ level_of, const int* __restrict__ ptr, int* out) {
    IPXK_GS(i, n) if (level_of[i] < 0 && ptr[i + 1] > ptr[i]) atomicMin(out, (int)i);
}
__global__ void min_of_list_kernel(int nf, const int* __restrict__ list, int* out) {
    IPXK_GS(t, nf) atomicMin(out, list[t]);
}
__global__ void level_keys_kernel(int n, const int* __restrict__ level_of, u64* __restrict__ keys, unsigned* __restrict__ vals) {
    IPXK_GS(i, n) {
        const unsigned lv = level_of[i] < 0 ? 0x7fffffffu : (unsigned)level_of[i];       // never reached (empty rows / columns): last
        keys[i] = ((u64)lv << 32) | (u64)i;
        vals[i] = (unsigned)i;
    }
}
__global__ void invert_perm_kernel(int n, const unsigned* __restrict__ perm, int* __restrict__ perm_out, int* __restrict__ inv) {
    IPXK_GS(i, n) { perm_out[i] = (int)perm[i]; inv[perm[i]] = (int)i; }
}
__global__ void permuted_keys_kernel(int64_t nz, const int* __restrict__ colof, const int* __restrict__ Ai, const int* __restrict__ colinv,
                                     const int* __restrict__ rowinv, u64* __restrict__ keys, unsigned* __restrict__ pos) {
    IPXK_GS(e, nz) {
        keys[e] = ((u64)(unsigned)colinv[colof[e]] << 32) | (u64)(unsigned)rowinv[Ai[e]];
        pos[e] = (unsigned)e;
    }
}
__global__ void permuted_fill_kernel(int64_t nz, const u64* __restrict__ keys, const unsigned* __restrict__ pos, const double* __restrict__ Ax,
                                     int* __restrict__ Ai_new, double* __restrict__ Ax_new, unsigned* __restrict__ col_new) {
    IPXK_GS(e, nz) {
        Ai_new[e] = (int)(keys[e] & 0xffffffffu);
        col_new[e] = (unsigned)(keys[e] >> 32);
        Ax_new[e] = Ax[pos[e]];
    }
}
__global__ void gather_rows_kernel(int n, const int* __restrict__ perm, const double* __restrict__ in, double* __restrict__ out) {
    IPXK_GS(i, n) out[i] = in[perm[i]];
}
__global__ void scatter_rows_kernel(int n, const int* __restrict__ perm, const double* __restrict__ in, double* __restrict__ out) {
    IPXK_GS(i, n) out[perm[i]] = in[i];
}
template <class K>
void sort_pairs_u64(Tmp& T, const K* kin, K* kout, const unsigned* vin, unsigned* vout, size_t n, int bits, hipStream_t s) {
    sort_pairs<K>(T, kin, kout, vin, vout, n, bits, s);
}

// breadth-first levels of the bipartite graph from row `start`; rows and columns not yet reached only.  Returns the number of row
// levels added (level numbers continue from level0), the rows reached, and the smallest row of the last row frontier.
// No host round trip per level: the kernels take the size of their frontier from a ring of three device counters (a level's kernel
// zeroes the counters two levels ahead), the sizes go into a history array, and the host looks at the history every kBfsBatch levels --
// launches past the last level find an empty frontier and do nothing.
constexpr int kBfsBatch = 16;
constexpr int kBfsMaxLevels = 1 << 20;
__global__ void bfs_level_kernel(const int* __restrict__ frontier, const int* __restrict__ ptr, const int* __restrict__ idx, int* level_of, int level,
                                 int* __restrict__ next, const int* count_in, int* count_out, int* zero_a, int* zero_b, int* hist_slot) {
    const int nf = *count_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (zero_a) *zero_a = 0;
        if (zero_b) *zero_b = 0;
        if (hist_slot) *hist_slot = nf;
    }
    IPXK_GS(t, nf) {
        const int v = frontier[t];
        for (int p = ptr[v]; p < ptr[v + 1]; p++) {
            const int w = idx[p];
            if (level_of[w] < 0 && atomicCAS(&level_of[w], -1, level) == -1) next[atomicAdd(count_out, 1)] = w;
        }
    }
}
struct BfsOut { int levels = 0; int64_t rows = 0; int last_min = -1; int64_t rows_by_8 = 0; };
BfsOut bfs_levels(Context* c, int start, int level0, int* row_level, int* col_level, int* fr, int* fc, int* ring, int* hist, std::vector<int>& hh) {
    hipStream_t s = c->stream;
    BfsOut out;
    // ring: cntR[3] at ring[0..2], cntC[3] at ring[3..5]
    const int init[6] = {1, 0, 0, 0, 0, 0};
    IPXK_HIP(hipMemcpyAsync(ring, init, sizeof(init), hipMemcpyHostToDevice, s));
    IPXK_HIP(hipMemcpyAsync(row_level + start, &level0, sizeof(int), hipMemcpyHostToDevice, s));
    IPXK_HIP(hipMemcpyAsync(fr, &start, sizeof(int), hipMemcpyHostToDevice, s));
    IPXK_HIP(hipStreamSynchronize(s));                    // (init, start, level0 are stack variables)
    const int grid = 256;                                 // (a frontier of a matrix with structure holds a few thousand rows; larger ones stride)
    int L = 0, found = -1;
    hh.clear();
    while (found < 0 && L < kBfsMaxLevels) {
        for (int b = 0; b < kBfsBatch; b++, L++) {
            int* cntR = ring + L % 3;
            int* cntC = ring + 3 + L % 3;
            // rows of level L -> their columns (level L); zeroes cntR[L+2] and cntC[L+1]; history of the row frontier sizes
            hipLaunchKernelGGL(bfs_level_kernel, dim3(grid), dim3(kBlock), 0, s, fr, c->pl_Tp.get(), c->pl_Ti.get(), col_level, level0 + L, fc, cntR, cntC,
                               ring + (L + 2) % 3, ring + 3 + (L + 1) % 3, hist + L);
            // those columns -> rows of level L + 1
            hipLaunchKernelGGL(bfs_level_kernel, dim3(grid), dim3(kBlock), 0, s, fc, c->pl_Ap.get(), c->pl_Ai.get(), row_level, level0 + L + 1, fr, cntC,
                               ring + (L + 1) % 3, (int*)nullptr, (int*)nullptr, (int*)nullptr);
        }
        hh.resize((size_t)L);
        IPXK_HIP(hipMemcpyAsync(hh.data() + (L - kBfsBatch), hist + (L - kBfsBatch), kBfsBatch * sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        for (int l = L - kBfsBatch; l < L; l++) if (hh[(size_t)l] == 0) { found = l; break; }
    }
    if (found < 0) found = L;
    out.levels = found;                                   // row levels level0 .. level0 + found - 1
    for (int l = 0; l < found; l++) { out.rows += hh[(size_t)l]; if (l <= 8) out.rows_by_8 = out.rows; }
    // the last non-empty row frontier is still in fr (later launches wrote nothing)
    const int last_nf = found > 0 ? hh[(size_t)found - 1] : 0;
    if (last_nf > 0) {
        const int big = 0x7fffffff;
        int got = big;
        IPXK_HIP(hipMemcpyAsync(ring, &big, sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(min_of_list_kernel, dim3(gridn(last_nf)), dim3(kBlock), 0, s, last_nf, fr, ring);
        IPXK_HIP(hipMemcpyAsync(&got, ring, sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        out.last_min = got == big ? start : got;
    } else {
        out.last_min = start;
    }
    return out;
}

}  // namespace

void reorder_model(Context* c) {
    Reordered& R = c->reord;
    R = Reordered();
    const char* env = getenv("IPXK_REORDER");
    if (env && env[0] == '0') return;
    const bool force = env && env[0] == '1';
    const int64_t m = c->m, n = c->n, nz = c->nnz;
    if (!c->have_plain || m < 2 || n < 1 || nz < 1 || c->nranks > 1) return;
    if (!force && nz < (int64_t(1) << 20)) return;          // small models: every gathered vector is cache resident anyway
    hipStream_t s = c->stream;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since0 = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    DevBuf<int> row_level((size_t)m), col_level((size_t)n), fr((size_t)m), fc((size_t)n), counters(8), hist((size_t)kBfsMaxLevels);
    std::vector<int> hh;
    int* h = nullptr;
    IPXK_HIP(hipHostMalloc(reinterpret_cast<void**>(&h), 4 * sizeof(int)));
    struct Free { int* p; ~Free() { (void)hipHostFree(p); } } free_h{h};
    auto clear_levels = [&]() {
        IPXK_HIP(hipMemsetAsync(row_level.get(), 0xff, (size_t)m * sizeof(int), s));
        IPXK_HIP(hipMemsetAsync(col_level.get(), 0xff, (size_t)n * sizeof(int), s));
    };
    auto first_unvisited = [&]() {
        const int big = 0x7fffffff;
        IPXK_HIP(hipMemcpyAsync(counters.get() + 7, &big, sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(first_unvisited_kernel, dim3(gridn(m)), dim3(kBlock), 0, s, (int)m, row_level.get(), c->pl_Tp.get(), counters.get() + 7);
        IPXK_HIP(hipMemcpyAsync(h + 2, counters.get() + 7, sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        return h[2] == big ? -1 : h[2];
    };
    // pass 1: from the first nonempty row, to find a row at the far end
    clear_levels();
    int start = first_unvisited();
    if (start < 0) return;
    BfsOut b1 = bfs_levels(c, start, 0, row_level.get(), col_level.get(), fr.get(), fc.get(), counters.get(), hist.get(), hh);
    if (!force && b1.rows_by_8 * 2 >= m) {                  // an expander: no numbering helps
        R.levels = b1.levels;
        R.ms = ms_since0();
        if (getenv("IPXK_VERBOSE")) fprintf(stderr, "ipxk: reordering: %lld of %lld rows within 8 levels of row %d -- no locality to recover (%.1f ms)\n",
                                            (long long)b1.rows_by_8, (long long)m, start, R.ms);
        return;
    }
    const double t_pass1 = ms_since0();
    // pass 2: the levels that count, from the far end; then the other components
    clear_levels();
    int level0 = 0, comps = 0;
    int64_t reached = 0;
    start = b1.last_min;
    while (start >= 0 && comps < kMaxComponents) {
        const BfsOut b = bfs_levels(c, start, level0, row_level.get(), col_level.get(), fr.get(), fc.get(), counters.get(), hist.get(), hh);
        level0 += b.levels;
        reached += b.rows;
        comps++;
        if (reached >= m) break;
        start = first_unvisited();
    }
    R.levels = level0;
    R.components = comps;
    const double t_pass2 = ms_since0();
    // numbering: (level, old index)
    Tmp T;
    const size_t big = (size_t)std::max(m, n);
    DevBuf<u64> k1(big), k2(big);
    DevBuf<unsigned> v1(big), v2(big);
    R.rowperm.ensure((size_t)m); R.rowinv.ensure((size_t)m); R.colperm.ensure((size_t)n); R.colinv.ensure((size_t)n);
    hipLaunchKernelGGL(level_keys_kernel, dim3(gridn(m)), dim3(kBlock), 0, s, (int)m, row_level.get(), k1.get(), v1.get());
    sort_pairs<u64>(T, k1.get(), k2.get(), v1.get(), v2.get(), (size_t)m, 64, s);
    hipLaunchKernelGGL(invert_perm_kernel, dim3(gridn(m)), dim3(kBlock), 0, s, (int)m, v2.get(), R.rowperm.get(), R.rowinv.get());
    hipLaunchKernelGGL(level_keys_kernel, dim3(gridn(n)), dim3(kBlock), 0, s, (int)n, col_level.get(), k1.get(), v1.get());
    sort_pairs<u64>(T, k1.get(), k2.get(), v1.get(), v2.get(), (size_t)n, 64, s);
    hipLaunchKernelGGL(invert_perm_kernel, dim3(gridn(n)), dim3(kBlock), 0, s, (int)n, v2.get(), R.colperm.get(), R.colinv.get());
    IPXK_HIP(hipStreamSynchronize(s));
    const double t_perm = ms_since0();
    // the matrix in the new numbering: entries keyed (new column, new row), sorted; then its row-wise copy as upload_plain_model builds it
    const size_t nz1 = (size_t)nz;
    R.Ap.ensure((size_t)n + 1); R.Ai.ensure(nz1); R.Ax.ensure(nz1); R.Tp.ensure((size_t)m + 1); R.Ti.ensure(nz1); R.Tx.ensure(nz1);
    {
        DevBuf<int> colof(nz1);
        DevBuf<u64> q1(nz1), q2(nz1);
        DevBuf<unsigned> p1(nz1), p2(nz1), cols_new(nz1);
        hipLaunchKernelGGL(rowof_kernel, dim3(gridn(n)), dim3(kBlock), 0, s, (int)n, c->pl_Ap.get(), colof.get(), (unsigned*)nullptr);
        hipLaunchKernelGGL(permuted_keys_kernel, dim3(gridn(nz)), dim3(kBlock), 0, s, nz, colof.get(), c->pl_Ai.get(), R.colinv.get(), R.rowinv.get(),
                           q1.get(), p1.get());
        sort_pairs<u64>(T, q1.get(), q2.get(), p1.get(), p2.get(), nz1, 32 + bits_for((u64)std::max<int64_t>(n, 2) - 1), s);
        hipLaunchKernelGGL(permuted_fill_kernel, dim3(gridn(nz)), dim3(kBlock), 0, s, nz, q2.get(), p2.get(), c->pl_Ax.get(), R.Ai.get(), R.Ax.get(),
                           cols_new.get());
        hipLaunchKernelGGL(row_pointers_kernel, dim3(gridn(n + 1)), dim3(kBlock), 0, s, n, nz, cols_new.get(), R.Ap.get());
        // Transpose (as in upload_plain_model)
        DevBuf<unsigned> pos(nz1), rows2(nz1), perm(nz1);
        hipLaunchKernelGGL(rowof_kernel, dim3(gridn(n)), dim3(kBlock), 0, s, (int)n, R.Ap.get(), colof.get(), pos.get());
        sort_pairs<unsigned>(T, reinterpret_cast<const unsigned*>(R.Ai.get()), rows2.get(), pos.get(), perm.get(), nz1, bits_for((u64)std::max<int64_t>(m, 2) - 1), s);
        hipLaunchKernelGGL(gather_transposed_kernel, dim3(gridn(nz)), dim3(kBlock), 0, s, nz, perm.get(), colof.get(), R.Ax.get(), R.Ti.get(), R.Tx.get());
        hipLaunchKernelGGL(row_pointers_kernel, dim3(gridn(m + 1)), dim3(kBlock), 0, s, m, nz, rows2.get(), R.Tp.get());
        IPXK_HIP(hipStreamSynchronize(s));
    }
    const double t_matrix = ms_since0();
    // its gather layouts, by the builders of the original
    {
        std::unique_ptr<LayoutScratch, void (*)(LayoutScratch*)> S(new_layout_scratch(), free_layout_scratch);
        R.Acols.csr_ptr = R.Ap.get(); R.Acols.csr_idx = R.Ai.get(); R.Acols.csr_val = R.Ax.get();
        R.Arows.csr_ptr = R.Tp.get(); R.Arows.csr_idx = R.Ti.get(); R.Arows.csr_val = R.Tx.get();
        const bool ok = R.Acols.build_device(*S, n, m, nz, R.Ap.get(), R.Ai.get(), R.Ax.get(), s) &&
                        R.Arows.build_device(*S, m, n, nz, R.Tp.get(), R.Ti.get(), R.Tx.get(), s);
        if (!ok) { R = Reordered(); return; }
    }
    const double t_layouts = ms_since0();
    R.us_original = time_normal_pair(c, c->Acols, c->Arows);
    R.us_reordered = time_normal_pair(c, R.Acols, R.Arows);
    R.active = force || R.us_reordered < 0.9f * R.us_original;
    R.ms = ms_since0();
    if (getenv("IPXK_VERBOSE"))
        fprintf(stderr, "ipxk: reordering: %d levels in %d component(s); the two products %.1f us on the model as given, %.1f us renumbered -> %s (%.1f ms)\n",
                R.levels, R.components, R.us_original, R.us_reordered, R.active ? "renumbered copy in use" : "not used", R.ms);
    if (getenv("IPXK_VERBOSE"))
        fprintf(stderr, "ipxk:   first pass %.1f ms, second pass + components %.1f, numbering %.1f, renumbered matrix %.1f, its layouts %.1f, timing both %.1f\n",
                t_pass1, t_pass2 - t_pass1, t_perm - t_pass2, t_matrix - t_perm, t_layouts - t_matrix, R.ms - t_layouts);
    if (!R.active) {                                        // keep the numbering (ipxk_reorder_info), drop the copy
        R.Acols = GatherMatrix(); R.Arows = GatherMatrix();
        R.Ap = DevBuf<int>(); R.Ai = DevBuf<int>(); R.Tp = DevBuf<int>(); R.Ti = DevBuf<int>(); R.Ax = DevBuf<double>(); R.Tx = DevBuf<double>();
        return;
    }
    R.W.ensure((size_t)(n + m)); R.diagonal.ensure((size_t)m); R.resscale.ensure((size_t)m); R.rhs.ensure((size_t)m); R.y.ensure((size_t)m);
    R.tcols.ensure((size_t)n);
}

void reorder_permute_rows(Context* c, const double* in_old, double* out_new) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gridn(c->m)), dim3(kBlock), 0, c->stream, (int)c->m, c->reord.rowperm.get(), in_old, out_new);
}
void reorder_unpermute_rows(Context* c, const double* in_new, double* out_old) {
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(gridn(c->m)), dim3(kBlock), 0, c->stream, (int)c->m, c->reord.rowperm.get(), in_new, out_old);
}
void reorder_permute_weights(Context* c, const double* W_old, double* W_new) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gridn(c->n)), dim3(kBlock), 0, c->stream, (int)c->n, c->reord.colperm.get(), W_old, W_new);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gridn(c->m)), dim3(kBlock), 0, c->stream, (int)c->m, c->reord.rowperm.get(), W_old + c->n, W_new + c->n);
}

}  // namespace ipxk
