// N as a matrix of its own on the device             reference src/splitted_normal_matrix.cc:42-55
//
// SplittedNormalMatrix::Prepare copies the nonbasic columns of AI into a matrix N (CopyColumns + ScaleColumn) and
// _Apply forms N N' with it (AddNormalProduct, src/sparse_matrix.cc:211-222).  Rounds 1-2 formed N N' on the MODEL
// matrix with zero weights on the other columns -- nothing to build, but every application streams all of A (and,
// after the round-3 tile compaction, still writes and combines partial sums for every column of A and gathers
// from a vector of n entries).  Here N is built for real, on the device, at Prepare:
//   * the structural columns with nonzero weight (NONBASIC, not fixed) are numbered 0 .. nN-1 in ascending order;
//   * two gather matrices in the XCD-sliced tile layout (internal.hpp) are built from the plain copies of A by ONE
//     radix sort of 64-bit keys each -- key = (tile | row in tile | position in storage order), so the sorted keys
//     ARE the layout: tile pointers by binary search, per-row counts by a histogram, indices and values by a gather;
//       P1: rows = the nN columns of N, gathering from u (m entries)      t = W_N .* (N' u)
//       P2: rows = the m rows of A, gathering from t (nN entries)          lhs = W_I .* u + N t
//     a row's entries keep their storage order (column order of the row-wise copy = the reference's Transpose), the
//     slices are those of the sliced layout, so the arithmetic is that of the sliced layout on the masked matrix:
//     same products, the same association per slice of the gathered vector -- but P2's slices are slices of the
//     COMPACT vector t (half as many at C3), so its partial sums differ in association from the masked form
//     (rounding level, covered by the 1e-12 operator gate);
//   * the scaling (the weights W_N) is refreshed with every rescale; the structure is rebuilt only when the set of
//     columns changes.
// Only for models whose gathered vectors need slicing (x beyond an XCD's L2) and without long rows; otherwise the
// masked / compacted model matrix of spmv.hip serves.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "context.hpp"
#include "spmv_kernels.hpp"
#include "trisolve.hpp"

namespace ipxk {

namespace {

using u64 = unsigned long long;
constexpr int kTileShift = 42, kRowShift = 32;      // key = tile << 42 | row in tile << 32 | storage position
#define IPXK_GS(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

int gridn(int64_t n) { return (int)std::min<int64_t>(4096, std::max<int64_t>(1, (n + kBlock - 1) / kBlock)); }

__global__ void nm_keep_kernel(int n, const double* __restrict__ W, const int* __restrict__ Ap, int* __restrict__ keep, int* __restrict__ len) {
    IPXK_GS(j, n) {
        const int k = W[j] != 0.0 ? 1 : 0;
        keep[j] = k;
        len[j] = k ? Ap[j + 1] - Ap[j] : 0;
    }
}
__global__ void nm_colof_kernel(int n, const int* __restrict__ keep, const int* __restrict__ newidx, int* __restrict__ colof,
                                int* __restrict__ newidx_or_minus) {
    IPXK_GS(j, n) {
        if (keep[j]) colof[newidx[j]] = (int)j;
        newidx_or_minus[j] = keep[j] ? newidx[j] : -1;
    }
}
// # columns whose kept flag differs from the previous build's (the exact test behind the fingerprint)
__global__ void nm_diff_kernel(int n, const int* __restrict__ keep, const int* __restrict__ keep_prev, int* out) {
    int d = 0;
    IPXK_GS(j, n) d += keep[j] != keep_prev[j] ? 1 : 0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) d += __shfl_xor(d, s, 64);
    if ((threadIdx.x & 63) == 0 && d) atomicAdd(out, d);
}
// a 64-bit fingerprint of the kept set (order-independent sum of a hash of the kept indices)
__global__ void nm_hash_kernel(int n, const int* __restrict__ keep, u64* out) {
    u64 h = 0;
    IPXK_GS(j, n) if (keep[j]) { u64 x = (u64)j * 0x9E3779B97F4A7C15ull; x ^= x >> 29; h += x * 0xBF58476D1CE4E5B9ull; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) h += __shfl_xor(h, d, 64);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(out, h);
}
// keys of P1: one per entry of a kept column, in any order
__global__ void nm_keys_p1_kernel(int n, const int* __restrict__ Ap, const int* __restrict__ Ai, const int* __restrict__ newidx,
                                  const int* __restrict__ off, int R, int ns, int slice, u64* __restrict__ keys) {
    IPXK_GS(j, n) {
        const int c = newidx[j];
        if (c < 0) continue;
        const u64 rb = (u64)(c / R), r = (u64)(c % R);
        int q = off[j];
        for (int p = Ap[j]; p < Ap[j + 1]; p++, q++) {
            const u64 tile = rb * ns + (u64)(Ai[p] / slice);
            keys[q] = (tile << kTileShift) | (r << kRowShift) | (u64)(unsigned)p;
        }
    }
}
// keys of P2: entries of the row-wise copy whose column is kept; the others get the largest key (sorted to the end)
__global__ void nm_keys_p2_kernel(int m, const int* __restrict__ Tp, const int* __restrict__ Ti, const int* __restrict__ newidx,
                                  int R, int ns, int slice, u64* __restrict__ keys) {
    IPXK_GS(i, m) {
        const u64 rb = (u64)(i / R), r = (u64)(i % R);
        for (int q = Tp[i]; q < Tp[i + 1]; q++) {
            const int c = newidx[Ti[q]];
            keys[q] = c < 0 ? ~0ull : (((rb * ns + (u64)(c / slice)) << kTileShift) | (r << kRowShift) | (u64)(unsigned)q);
        }
    }
}
__global__ void nm_tileptr_kernel(int ntiles, int64_t nz, const u64* __restrict__ keys, unsigned* __restrict__ ptr) {
    IPXK_GS(t, (int64_t)ntiles + 1) {
        const u64 want = (u64)t << kTileShift;
        int64_t lo = 0, hi = nz;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
        ptr[t] = (unsigned)lo;
    }
}
// entries per (tile, row), indices, values; P2 gathers the NEW column index
__global__ void nm_fill_kernel(int64_t nz, const u64* __restrict__ keys, int R, const int* __restrict__ src_idx,
                               const double* __restrict__ src_val, const int* __restrict__ newidx, int* __restrict__ cnt32,
                               int* __restrict__ idx, double* __restrict__ val) {
    IPXK_GS(e, nz) {
        const u64 k = keys[e];
        const unsigned p = (unsigned)(k & 0xffffffffull);
        const int64_t slot = (int64_t)(k >> kTileShift) * R + (int64_t)((k >> kRowShift) & 1023ull);
        atomicAdd(cnt32 + slot, 1);
        const int g = src_idx[p];
        idx[e] = newidx ? newidx[g] : g;
        val[e] = src_val[p];
    }
}
__global__ void nm_pack_counts_kernel(int64_t nslots, const int* __restrict__ cnt32, unsigned char* __restrict__ cnt, int* overflow) {
    IPXK_GS(x, nslots) {
        const int v = cnt32[x];
        if (v > 255) *overflow = 1;
        cnt[x] = (unsigned char)v;
    }
}
__global__ void nm_max_tile_kernel(int ntiles, const unsigned* __restrict__ ptr, int* out) {
    int best = 0;
    IPXK_GS(t, ntiles) best = max(best, (int)(ptr[t + 1] - ptr[t]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, best);
}
__global__ void nm_weights_kernel(int nN, const int* __restrict__ colof, const double* __restrict__ W, double* __restrict__ wN) {
    IPXK_GS(c, nN) wN[c] = W[colof[c]];
}

// ---- N in plain compact form (for the accumulated-tiles layout, layout_device.hip) ----
// N' by row (= kept column c of A): ptr1[c] = off[colof[c]], entries copied in storage order
__global__ void nm_csr1_kernel(int nN, int64_t nzN, const int* __restrict__ colof, const int* __restrict__ off, const int* __restrict__ Ap,
                               const int* __restrict__ Ai, const double* __restrict__ Ax, int* __restrict__ ptr, int* __restrict__ idx,
                               double* __restrict__ val) {
    IPXK_GS(c, nN) {
        const int j = colof[c], o = off[j];
        ptr[c] = o;
        for (int p = Ap[j], q = o; p < Ap[j + 1]; p++, q++) { idx[q] = Ai[p]; val[q] = Ax[p]; }
        if (c == nN - 1) ptr[nN] = (int)nzN;
    }
}
// N by row of A: the entries whose column is kept, compact column index
__global__ void nm_csr2_count_kernel(int m, const int* __restrict__ Tp, const int* __restrict__ Ti, const int* __restrict__ newidx, int* __restrict__ cnt) {
    IPXK_GS(i, m) {
        int k = 0;
        for (int q = Tp[i]; q < Tp[i + 1]; q++) k += newidx[Ti[q]] >= 0 ? 1 : 0;
        cnt[i] = k;
    }
}
__global__ void nm_csr2_fill_kernel(int m, int64_t nzN, const int* __restrict__ Tp, const int* __restrict__ Ti, const double* __restrict__ Tx,
                                    const int* __restrict__ newidx, int* __restrict__ ptr, int* __restrict__ idx, double* __restrict__ val) {
    IPXK_GS(i, m) {
        int o = ptr[i];
        for (int q = Tp[i]; q < Tp[i + 1]; q++) {
            const int c = newidx[Ti[q]];
            if (c >= 0) { idx[o] = c; val[o] = Tx[q]; o++; }
        }
        if (i == m - 1) ptr[m] = (int)nzN;
    }
}

struct Tmp {
    DevBuf<unsigned char> bytes;
    void* need(size_t n) { if (bytes.size() < n) bytes.resize(n); return bytes.get(); }
};

int slices_for(int64_t ncols, int64_t* slice_out) {
    int64_t slice_bytes = int64_t(2) << 20;
    if (const char* e = getenv("IPXK_SLICE_TEST_KB"))        // tests: slices for small matrices (as in spmv.hip)
        if (atoi(e) > 0) slice_bytes = (int64_t)atoi(e) << 10;
    const int64_t x_bytes = ncols * 8;
    int ns = 1;
    if (x_bytes > 2 * slice_bytes) { ns = 2; while (ns < 8 && x_bytes > (int64_t)ns * slice_bytes) ns *= 2; }
    *slice_out = ((ncols + ns - 1) / ns + 15) / 16 * 16;
    return ns;
}

}  // namespace

struct NMatrix {
    bool valid = false;
    int nN = 0;
    u64 kept_hash = 0;
    SlicedMatrix P1, P2;
    AccMatrix A1, A2;                  // the same two gather matrices as accumulated tiles (used when built)
    DevBuf<int> cptr, cidx;            // scratch: N in plain compact form
    DevBuf<double> cval;
    DevBuf<int> keep, keep_built, len, newidx, off, colof, cnt32, counters;     // keep_built: the kept flags N was built for
    DevBuf<u64> keys, keys2, hash;
    DevBuf<double> wN, tN;
    Tmp T;
};
void destroy_nmatrix(NMatrix* N) { delete N; }

static void scan_int(Tmp& T, const int* in, int* out, size_t n, hipStream_t s) {
    size_t bytes = 0;
    IPXK_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, n, rocprim::plus<int>(), s));
    IPXK_HIP(rocprim::exclusive_scan(T.need(bytes), bytes, in, out, 0, n, rocprim::plus<int>(), s));
}
static void sort_u64(Tmp& T, u64* a, u64* b, size_t n, hipStream_t s) {
    size_t bytes = 0;
    IPXK_HIP(rocprim::radix_sort_keys(nullptr, bytes, a, b, n, 0u, 64u, s));
    IPXK_HIP(rocprim::radix_sort_keys(T.need(bytes), bytes, a, b, n, 0u, 64u, s));
}

// one gather matrix from its sorted keys; false if a tile does not fit (the caller retries with fewer rows per tile)
static bool finish_layout(NMatrix& N, SlicedMatrix& P, int nrows, int R, int ns, int64_t nz, const u64* keys, const int* src_idx,
                          const double* src_val, const int* newidx, hipStream_t s) {
    const int nrb = (nrows + R - 1) / R;
    const int ntiles = nrb * ns;
    P.R = R; P.nslices = ns; P.nrb = nrb; P.nrows_pad = nrb * R;
    P.tile_ptr.ensure((size_t)ntiles + 1);
    hipLaunchKernelGGL(nm_tileptr_kernel, dim3(gridn(ntiles + 1)), dim3(kBlock), 0, s, ntiles, nz, keys, P.tile_ptr.get());
    IPXK_HIP(hipMemsetAsync(N.counters.get(), 0, 4 * sizeof(int), s));
    hipLaunchKernelGGL(nm_max_tile_kernel, dim3(gridn(ntiles)), dim3(kBlock), 0, s, ntiles, P.tile_ptr.get(), N.counters.get());
    int h[4] = {0, 0, 0, 0};
    IPXK_HIP(hipMemcpyAsync(h, N.counters.get(), sizeof h, hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    P.max_tile = h[0];
    if (P.max_tile > kSlicedMaxTile) return false;
    const int64_t nslots = (int64_t)ntiles * R;
    N.cnt32.ensure((size_t)nslots);
    IPXK_HIP(hipMemsetAsync(N.cnt32.get(), 0, (size_t)nslots * sizeof(int), s));
    P.idx.ensure((size_t)std::max<int64_t>(nz, 1)); P.val.ensure((size_t)std::max<int64_t>(nz, 1));
    P.cnt.ensure((size_t)nslots);
    hipLaunchKernelGGL(nm_fill_kernel, dim3(gridn(nz)), dim3(kBlock), 0, s, nz, keys, R, src_idx, src_val, newidx, N.cnt32.get(),
                       P.idx.get(), P.val.get());
    hipLaunchKernelGGL(nm_pack_counts_kernel, dim3(gridn(nslots)), dim3(kBlock), 0, s, nslots, N.cnt32.get(), P.cnt.get(), N.counters.get() + 1);
    IPXK_HIP(hipMemcpyAsync(h, N.counters.get(), sizeof h, hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    if (h[1]) return false;                  // > 255 entries of a row in one slice: the byte counts cannot hold it
    P.partial.ensure((size_t)ns * P.nrows_pad);
    P.built = true;
    return true;
}

// (Re)builds N for the weights W (n + m: zero on the columns that are not in N); returns false when N is not used
// for this model (then the caller's masked / compacted model matrix serves).
bool nmatrix_prepare(Context* c, const double* W) {
    if (getenv("IPXK_REAL_N") && getenv("IPXK_REAL_N")[0] == '0') return false;
    const int m = (int)c->m, n = (int)c->n;
    // only where the gathered vectors need slicing, and no long rows
    if (!(c->Arows.use_sliced && c->Arows.sliced.nslices > 1 && c->Acols.use_sliced && c->Acols.nlong == 0 && c->Arows.nlong == 0)) return false;
    if (c->nnz >= (int64_t(1) << 31) || n < 1 || m < 1) return false;
    hipStream_t s = c->stream;
    if (!c->nmat) c->nmat = new NMatrix;
    NMatrix& N = *c->nmat;
    IPXK_REQUIRE(c->have_plain, "no resident copy of the matrix");
    N.keep.ensure((size_t)n); N.len.ensure((size_t)n); N.newidx.ensure((size_t)n); N.off.ensure((size_t)n);
    N.counters.ensure(4); N.hash.ensure(1);
    hipLaunchKernelGGL(nm_keep_kernel, dim3(gridn(n)), dim3(kBlock), 0, s, n, W, c->pl_Ap.get(), N.keep.get(), N.len.get());
    IPXK_HIP(hipMemsetAsync(N.hash.get(), 0, sizeof(u64), s));
    hipLaunchKernelGGL(nm_hash_kernel, dim3(gridn(n)), dim3(kBlock), 0, s, n, N.keep.get(), N.hash.get());
    u64 hsh = 0;
    IPXK_HIP(hipMemcpyAsync(&hsh, N.hash.get(), sizeof hsh, hipMemcpyDeviceToHost, s));
    IPXK_HIP(hipStreamSynchronize(s));
    // the structure is reused only for EXACTLY the same set of columns: the fingerprint is a fast reject, the flags
    // of the build decide (a colliding fingerprint must not bring back the N of another basis)
    bool same = N.valid && hsh == N.kept_hash && N.keep_built.size() >= (size_t)n;
    if (same) {
        IPXK_HIP(hipMemsetAsync(N.counters.get(), 0, sizeof(int), s));
        hipLaunchKernelGGL(nm_diff_kernel, dim3(gridn(n)), dim3(kBlock), 0, s, n, N.keep.get(), N.keep_built.get(), N.counters.get());
        int diff = 1;
        IPXK_HIP(hipMemcpyAsync(&diff, N.counters.get(), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        same = diff == 0;
    }
    if (!same) {
        N.valid = false;
        // numbering of the kept columns and their entry offsets
        scan_int(N.T, N.keep.get(), N.newidx.get(), (size_t)n, s);
        scan_int(N.T, N.len.get(), N.off.get(), (size_t)n, s);
        int last[4];
        IPXK_HIP(hipMemcpyAsync(&last[0], N.newidx.get() + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipMemcpyAsync(&last[1], N.keep.get() + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipMemcpyAsync(&last[2], N.off.get() + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipMemcpyAsync(&last[3], N.len.get() + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        const int nN = last[0] + last[1];
        const int64_t nzN = (int64_t)last[2] + last[3];
        if (nN < 1 || nzN < 1) return false;
        N.nN = nN;
        N.colof.ensure((size_t)nN);
        // (newidx becomes -1 on the columns that are not kept)
        hipLaunchKernelGGL(nm_colof_kernel, dim3(gridn(n)), dim3(kBlock), 0, s, n, N.keep.get(), N.newidx.get(), N.colof.get(), N.newidx.get());
        N.keys.ensure((size_t)c->nnz); N.keys2.ensure((size_t)c->nnz);
        // P1: rows = kept columns, gathered index = row of A
        int64_t slice1 = 0, slice2 = 0;
        const int ns1 = slices_for(m, &slice1), ns2 = slices_for(nN, &slice2);
        N.P1 = SlicedMatrix(); N.P2 = SlicedMatrix();
        bool ok = false;
        for (int R = kSlicedRows; R >= kBlock && !ok; R /= 2) {
            hipLaunchKernelGGL(nm_keys_p1_kernel, dim3(gridn(n)), dim3(kBlock), 0, s, n, c->pl_Ap.get(), c->pl_Ai.get(), N.newidx.get(), N.off.get(), R,
                               ns1, (int)slice1, N.keys.get());
            sort_u64(N.T, N.keys.get(), N.keys2.get(), (size_t)nzN, s);
            ok = finish_layout(N, N.P1, nN, R, ns1, nzN, N.keys2.get(), c->pl_Ai.get(), c->pl_Ax.get(), nullptr, s);
        }
        if (!ok) return false;
        ok = false;
        for (int R = kSlicedRows; R >= kBlock && !ok; R /= 2) {
            hipLaunchKernelGGL(nm_keys_p2_kernel, dim3(gridn(m)), dim3(kBlock), 0, s, m, c->pl_Tp.get(), c->pl_Ti.get(), N.newidx.get(), R, ns2, (int)slice2,
                               N.keys.get());
            sort_u64(N.T, N.keys.get(), N.keys2.get(), (size_t)c->nnz, s);      // the entries of dropped columns sort to the end
            ok = finish_layout(N, N.P2, m, R, ns2, nzN, N.keys2.get(), c->pl_Ti.get(), c->pl_Tx.get(), N.newidx.get(), s);
        }
        if (!ok) return false;
        // the accumulated-tiles form of both (IPXK_SPMV_ACC=0: the sliced tiles stay)
        N.A1 = AccMatrix(); N.A2 = AccMatrix();
        if (!(getenv("IPXK_SPMV_ACC") && getenv("IPXK_SPMV_ACC")[0] == '0') && ns1 > 1 && ns2 > 1) {
            std::unique_ptr<LayoutScratch, void (*)(LayoutScratch*)> LS(new_layout_scratch(), free_layout_scratch);
            const size_t nzn = (size_t)nzN;
            N.cptr.ensure((size_t)std::max(nN, m) + 1); N.cidx.ensure(nzn); N.cval.ensure(nzn);
            hipLaunchKernelGGL(nm_csr1_kernel, dim3(gridn(nN)), dim3(kBlock), 0, s, nN, nzN, N.colof.get(), N.off.get(), c->pl_Ap.get(), c->pl_Ai.get(),
                               c->pl_Ax.get(), N.cptr.get(), N.cidx.get(), N.cval.get());
            AccMatrix a1, a2;
            const bool ok1 = device_build_acc(*LS, a1, N.P1, nN, m, nzN, N.cptr.get(), N.cidx.get(), N.cval.get(), s);
            N.cnt32.ensure((size_t)m + 1);
            hipLaunchKernelGGL(nm_csr2_count_kernel, dim3(gridn(m)), dim3(kBlock), 0, s, m, c->pl_Tp.get(), c->pl_Ti.get(), N.newidx.get(), N.cnt32.get());
            scan_int(N.T, N.cnt32.get(), N.cptr.get(), (size_t)m, s);
            hipLaunchKernelGGL(nm_csr2_fill_kernel, dim3(gridn(m)), dim3(kBlock), 0, s, m, nzN, c->pl_Tp.get(), c->pl_Ti.get(), c->pl_Tx.get(),
                               N.newidx.get(), N.cptr.get(), N.cidx.get(), N.cval.get());
            const bool ok2 = device_build_acc(*LS, a2, N.P2, m, nN, nzN, N.cptr.get(), N.cidx.get(), N.cval.get(), s);
            if (ok1 && ok2) { N.A1 = std::move(a1); N.A2 = std::move(a2); }
        }
        N.wN.ensure((size_t)nN); N.tN.ensure((size_t)nN);
        N.kept_hash = hsh;
        N.keep_built.ensure((size_t)n);
        IPXK_HIP(hipMemcpyAsync(N.keep_built.get(), N.keep.get(), (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, s));
        N.valid = true;
        if (getenv("IPXK_VERBOSE"))
            fprintf(stderr, "ipxk: N built on the device: %d of %d structural columns, %lld entries; N'u tiles %d x %d (rows %d), N t tiles %d x %d (rows %d)\n",
                    nN, n, (long long)nzN, N.P1.nrb, N.P1.nslices, N.P1.R, N.P2.nrb, N.P2.nslices, N.P2.R);
    }
    hipLaunchKernelGGL(nm_weights_kernel, dim3(gridn(N.nN)), dim3(kBlock), 0, s, N.nN, N.colof.get(), W, N.wN.get());
    IPXK_HIP(hipGetLastError());
    return true;
}

void nmatrix_invalidate(Context* c) { if (c->nmat) c->nmat->valid = false; }

static SlicedView view_of(const SlicedMatrix& P, int nrows) {
    SlicedView V;
    V.nrows = nrows; V.nrows_pad = P.nrows_pad; V.nslices = P.nslices; V.nrb = P.nrb; V.R = P.R;
    V.tile_ptr = P.tile_ptr.get(); V.cnt = P.cnt.get(); V.idx = P.idx.get(); V.val = P.val.get(); V.partial = P.partial.get();
    V.row_long = nullptr; V.masked = 0;
    return V;
}
template <class Epi>
static void launch_tiles(const SlicedMatrix& P, int nrows, const double* x, const Epi& epi, const int* done, hipStream_t s) {
    const SlicedView V = view_of(P, nrows);
    const size_t lds = (size_t)(P.max_tile + P.max_tile / 32 + 1) * sizeof(double);
    const dim3 grid(V.nrb * V.nslices), block(kBlock);
    if (V.R == kBlock * 4) hipLaunchKernelGGL((spmv_sliced_tile_kernel<Epi, 4, false, false>), grid, block, lds, s, V, x, epi, (double*)nullptr, done);
    else if (V.R == kBlock * 2) hipLaunchKernelGGL((spmv_sliced_tile_kernel<Epi, 2, false, false>), grid, block, lds, s, V, x, epi, (double*)nullptr, done);
    else hipLaunchKernelGGL((spmv_sliced_tile_kernel<Epi, 1, false, false>), grid, block, lds, s, V, x, epi, (double*)nullptr, done);
    const int cg = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ((int64_t)nrows + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(spmv_sliced_combine_kernel<Epi>, dim3(cg), dim3(kBlock), 0, s, V, epi, (double*)nullptr, done);
}

template <class Epi>
static void launch_acc(const AccMatrix& A, const SlicedMatrix& P, int nrows, const double* x, const Epi& epi, const int* done, hipStream_t s) {
    AccView W;
    W.nrows = nrows; W.nrows_pad = A.nrows_pad; W.nslices = A.nslices; W.nrb = A.nrb; W.RB = A.RB; W.slice_elems = A.slice_elems;
    W.tile_batch = A.tile_batch.get(); W.bptr = A.bptr.get(); W.pack = A.pack.get(); W.val = A.val.get(); W.partial = A.partial.get();
    static bool lds_attr_set = false;
    if (!lds_attr_set) {
        IPXK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spmv_acc_tile_kernel<Epi>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(kAccMaxRows * sizeof(double))));
        lds_attr_set = true;
    }
    hipLaunchKernelGGL((spmv_acc_tile_kernel<Epi>), dim3(W.nrb * W.nslices), dim3(kAccThreads), (size_t)W.RB * sizeof(double), s, W, x, done);
    SlicedView V = view_of(P, nrows);
    V.nrows_pad = A.nrows_pad; V.partial = A.partial.get();
    const int cg = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ((int64_t)nrows + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(spmv_sliced_combine_kernel<Epi>, dim3(cg), dim3(kBlock), 0, s, V, epi, (double*)nullptr, done);
}

// work = W_I .* u + N (W_N .* (N' u))        (AddNormalProduct on N, plus the slack columns of N: the identity part)
void nmatrix_apply(Context* c, const double* WI, const double* u, double* work, const int* done) {
    NMatrix& N = *c->nmat;
    hipStream_t s = c->stream;
    EpiScale e1{{}, N.wN.get(), N.tN.get()};
    EpiNormalRows e2{{}, WI, u, work};
    if (N.A1.built && N.A2.built) {
        launch_acc(N.A1, N.P1, N.nN, u, e1, done, s);
        launch_acc(N.A2, N.P2, (int)c->m, N.tN.get(), e2, done, s);
        return;
    }
    launch_tiles(N.P1, N.nN, u, e1, done, s);
    launch_tiles(N.P2, (int)c->m, N.tN.get(), e2, done, s);
}

}  // namespace ipxk
