// Device-side analysis of the four triangular sweeps of SplittedNormalMatrix::Prepare
// (reference src/splitted_normal_matrix.cc:18-66 hands over L, U; the sweeps are those of
// TriangularSolve, src/sparse_matrix.cc:224-301).
//
// The factors change every IPM iteration, so the level schedule has to be rebuilt every time.  On
// the host that is ~100 ms of pointer chasing at 1M rows; here L and U are uploaded as given and
// everything else runs on the GPU:
//   1. row lists in natural order: U' and L' read the columns directly; the forward sweeps need
//      the row-wise forms -- a stable radix sort of the entries by row index (rocPRIM via hipCUB;
//      entries enumerated in ascending column order for L and in DESCENDING column order for U, the
//      order in which the reference's column loops update a row);
//   2. dependency levels by relaxation level[i] = max(level[dep] + 1) until nothing changes
//      (as many sweeps over the entries as the DAG is deep);
//   3. a stable sort of the unknowns (in processing order) by level, levels padded to 64 positions;
//   4. row extents by a prefix sum, rows gathered into level order.
// The host only sees O(#levels) numbers (level sizes, long-row flags, entry offsets) from which it
// derives the launch plan (plan_sweep).  The result is identical to analyse_sweep's host arrays.
#include <hipcub/hipcub.hpp>

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
                               const double* __restrict__ Ux, const double* __restrict__ uscale,
                               int* __restrict__ rp, int* __restrict__ ri, double* __restrict__ rx,
                               double* __restrict__ rxS, double* __restrict__ dgn, double* __restrict__ dgnS) {
    IPXK_GRID_STRIDE(k, m) {
        const ipxint p0 = Up[k], p1 = Up[k + 1] - 1;      // diagonal last
        const int base = (int)(p0 - k);
        rp[k] = base;
        const double sc = uscale[k];
        for (ipxint p = p0; p < p1; p++) {
            const int q = base + (int)(p - p0);
            ri[q] = (int)Ui[p];
            rx[q] = Ux[p];
            rxS[q] = Ux[p] * sc;
        }
        dgn[k] = Ux[p1];
        dgnS[k] = Ux[p1] * sc;
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

__global__ void rows_from_perm_kernel(int64_t nz, const int* __restrict__ perm, const int* __restrict__ colof,
                                      const double* __restrict__ X, const double* __restrict__ uscale,
                                      int* __restrict__ ri, double* __restrict__ rx, double* __restrict__ rxS) {
    IPXK_GRID_STRIDE(t, nz) {
        const int p = perm[t];
        const int col = colof[p];
        ri[t] = col;
        rx[t] = X[p];
        if (rxS) rxS[t] = X[p] * uscale[col];
    }
}

__global__ void u_diag_kernel(int m, const ipxint* __restrict__ Up, const double* __restrict__ Ux,
                              const double* __restrict__ uscale, double* __restrict__ dgn,
                              double* __restrict__ dgnS) {
    IPXK_GRID_STRIDE(k, m) {
        const double d = Ux[Up[k + 1] - 1];
        dgn[k] = d;
        dgnS[k] = d * uscale[k];
    }
}

// ---- step 2: levels -----------------------------------------------------------------------
__global__ void relax_levels_kernel(int dim, const int* __restrict__ rp, const int* __restrict__ ri,
                                    int* level, int* changed) {
    IPXK_GRID_STRIDE(i, dim) {
        int lv = 0;
        for (int p = rp[i]; p < rp[i + 1]; p++) {
            const int l = __hip_atomic_load(level + ri[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            lv = l > lv ? l : lv;
        }
        if (lv > level[i]) {
            __hip_atomic_store(level + i, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *changed = 1;
        }
    }
}

// ---- step 3: order ------------------------------------------------------------------------
__global__ void level_keys_kernel(int dim, int ascending, const int* __restrict__ level, int* __restrict__ keys,
                                  int* __restrict__ vals) {
    IPXK_GRID_STRIDE(t, dim) {
        const int i = ascending ? (int)t : dim - 1 - (int)t;
        keys[t] = level[i];
        vals[t] = i;
    }
}

__global__ void place_kernel(int dim, const int* __restrict__ sorted_level, const int* __restrict__ sorted_unknown,
                             const int* __restrict__ lstart, const int* __restrict__ lptr,
                             const int* __restrict__ rp, int* __restrict__ order, int* __restrict__ posof,
                             unsigned char* __restrict__ level_long) {
    IPXK_GRID_STRIDE(t, dim) {
        const int l = sorted_level[t], i = sorted_unknown[t];
        const int pos = lptr[l] + ((int)t - lstart[l]);
        order[pos] = i;
        posof[i] = pos;
        if (rp[i + 1] - rp[i] > kShortRow) level_long[l] = 1;
    }
}

// ---- step 4: rows into level order -----------------------------------------------------------
__global__ void row_length_kernel(int npos, const int* __restrict__ order, const int* __restrict__ rp,
                                  int* __restrict__ len) {
    IPXK_GRID_STRIDE(k, (int64_t)npos + 1) {
        const int i = k < npos ? order[k] : -1;
        len[k] = i >= 0 ? rp[i + 1] - rp[i] : 0;
    }
}

__global__ void gather_rows_kernel(int npos, const int* __restrict__ order, const int* __restrict__ ptr,
                                   const int* __restrict__ rp, const int* __restrict__ ri,
                                   const double* __restrict__ rx, const double* __restrict__ rxS,
                                   const double* __restrict__ dgn, const double* __restrict__ dgnS,
                                   int* __restrict__ idx, double* __restrict__ val, double* __restrict__ valS,
                                   double* __restrict__ dg, double* __restrict__ dgS) {
    IPXK_GRID_STRIDE(k, npos) {
        const int i = order[k];
        if (i < 0) {
            dg[k] = 1.0;
            if (dgS) dgS[k] = 1.0;
            continue;
        }
        int put = ptr[k];
        for (int p = rp[i]; p < rp[i + 1]; p++, put++) {
            idx[put] = ri[p];
            val[put] = rx[p];
            if (valS) valS[put] = rxS[p];
        }
        dg[k] = dgn[i];
        if (dgS) dgS[k] = dgnS[i];
    }
}

__global__ void gather_int_kernel(int n, const int* __restrict__ src, const int* __restrict__ at, int* __restrict__ out) {
    IPXK_GRID_STRIDE(i, n) out[i] = src[at[i]];
}

// dependency-slot table of one tail run (tail_lds_kernel)
__global__ void tslot_kernel(int ne, int e0, int k0, int k1, const int* __restrict__ idx,
                             const int* __restrict__ posof, short* __restrict__ tslot) {
    IPXK_GRID_STRIDE(e, ne) {
        const int pj = posof[idx[e0 + e]];
        tslot[e] = pj >= k0 && pj < k1 ? (short)(pj - k0) : (short)-1;
    }
}

struct Scratch {   // reused by the four sweeps of one Prepare
    DevBuf<int> keys, vals, keys2, vals2, colof, level, posof, len, lstart, lptr, at, tmpi;
    DevBuf<unsigned char> level_long, cub;
    DevBuf<int> rp, ri;
    DevBuf<double> rx, rxS, dgn, dgnS;
    int* h_flag = nullptr;   // pinned
    ~Scratch() { if (h_flag) (void)hipHostFree(h_flag); }
};

void sort_pairs(Scratch& W, int64_t n, int end_bit, hipStream_t s) {
    size_t bytes = 0;
    IPXK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, W.keys.get(), W.keys2.get(), W.vals.get(),
                                                W.vals2.get(), (int)n, 0, end_bit, s));
    if (W.cub.size() < bytes) W.cub.resize(bytes);
    IPXK_HIP(hipcub::DeviceRadixSort::SortPairs(W.cub.get(), bytes, W.keys.get(), W.keys2.get(), W.vals.get(),
                                                W.vals2.get(), (int)n, 0, end_bit, s));
}

int bits_for(int64_t n) {   // bits needed for values in [0, n)
    int b = 1;
    while ((int64_t(1) << b) < n) b++;
    return b;
}

// Deep dependency graphs: the relaxation needs as many rounds as there are levels, each a pass over
// all entries.  After kMaxRelaxRounds rounds the levels are computed by one sequential O(nnz) scan on
// the host instead (the unknowns' processing order is a topological order) and uploaded.
constexpr int kMaxRelaxRounds = 24;    // x 8 launches: a few milliseconds at 1M rows

// levels, order, level-ordered rows and launch plan of one sweep from its natural-order row list
template <class HostLevels>
void finish_sweep(Context* c, Scratch& W, Sweep& S, int dim, int64_t nz, bool ascending, bool running, bool scaled,
                  HostLevels&& host_levels) {
    hipStream_t s = c->stream;
    S.dim = dim;
    S.running = running;
    S.has_scaled = scaled;
    // 2. levels
    W.level.resize(std::max(dim, 1));
    IPXK_HIP(hipMemsetAsync(W.level.get(), 0, sizeof(int) * std::max(dim, 1), s));
    DevBuf<int> changed(1);
    if (!W.h_flag) IPXK_HIP(hipHostMalloc(reinterpret_cast<void**>(&W.h_flag), sizeof(int)));
    for (int round = 0; dim > 0; round++) {
        if (round >= kMaxRelaxRounds) {
            std::vector<int> lv((size_t)dim, 0);
            host_levels(lv);
            W.level.upload(lv, s);
            break;
        }
        IPXK_HIP(hipMemsetAsync(changed.get(), 0, sizeof(int), s));
        for (int r = 0; r < 8; r++)
            hipLaunchKernelGGL(relax_levels_kernel, dim3(grid_for(dim)), dim3(kBlock), 0, s, dim, W.rp.get(),
                               W.ri.get(), W.level.get(), changed.get());
        IPXK_HIP(hipMemcpyAsync(W.h_flag, changed.get(), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        if (*W.h_flag == 0) break;
    }
    // 3. stable sort of the unknowns in processing order by level
    W.keys.resize(std::max<int64_t>(std::max<int64_t>(dim, nz), 1));
    W.vals.resize(W.keys.size()); W.keys2.resize(W.keys.size()); W.vals2.resize(W.keys.size());
    int nlev = 0;
    std::vector<int> lstart, lptr;
    if (dim > 0) {
        hipLaunchKernelGGL(level_keys_kernel, dim3(grid_for(dim)), dim3(kBlock), 0, s, dim, ascending ? 1 : 0,
                           W.level.get(), W.keys.get(), W.vals.get());
        sort_pairs(W, dim, 31, s);
        // the last sorted key is the deepest level
        IPXK_HIP(hipMemcpyAsync(W.h_flag, W.keys2.get() + (dim - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        IPXK_HIP(hipStreamSynchronize(s));
        nlev = *W.h_flag + 1;
        W.lstart.resize(nlev + 1);
        hipLaunchKernelGGL(lower_bound_kernel, dim3(grid_for(nlev + 1)), dim3(kBlock), 0, s, nlev, (int64_t)dim,
                           W.keys2.get(), W.lstart.get());
        lstart.resize(nlev + 1);
        W.lstart.download(lstart.data(), (size_t)nlev + 1, s);
        IPXK_HIP(hipStreamSynchronize(s));
    }
    lptr.assign(nlev + 1, 0);
    for (int l = 0; l < nlev; l++) lptr[l + 1] = lptr[l] + (lstart[l + 1] - lstart[l] + 63) / 64 * 64;
    const int npos = lptr[nlev];
    S.nlevels = nlev;
    S.npos = npos;
    S.level_ptr = lptr;
    S.level_ptr_dev.upload(lptr, s);
    S.order.resize(std::max(npos, 1));
    IPXK_HIP(hipMemsetAsync(S.order.get(), 0xff, sizeof(int) * std::max(npos, 1), s));
    W.posof.resize(std::max(dim, 1));
    W.level_long.resize(std::max(nlev, 1));
    IPXK_HIP(hipMemsetAsync(W.level_long.get(), 0, std::max(nlev, 1), s));
    if (dim > 0)
        hipLaunchKernelGGL(place_kernel, dim3(grid_for(dim)), dim3(kBlock), 0, s, dim, W.keys2.get(), W.vals2.get(),
                           W.lstart.get(), S.level_ptr_dev.get(), W.rp.get(), S.order.get(), W.posof.get(),
                           W.level_long.get());
    // 4. row extents and rows in level order
    W.len.resize((size_t)npos + 1);
    S.ptr.resize((size_t)npos + 1);
    hipLaunchKernelGGL(row_length_kernel, dim3(grid_for(npos + 1)), dim3(kBlock), 0, s, npos, S.order.get(),
                       W.rp.get(), W.len.get());
    {
        size_t bytes = 0;
        IPXK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, W.len.get(), S.ptr.get(), npos + 1, s));
        if (W.cub.size() < bytes) W.cub.resize(bytes);
        IPXK_HIP(hipcub::DeviceScan::ExclusiveSum(W.cub.get(), bytes, W.len.get(), S.ptr.get(), npos + 1, s));
    }
    const size_t nzs = (size_t)std::max<int64_t>(nz, 1);
    S.idx.resize(nzs); S.val.resize(nzs); S.diag.resize(std::max(npos, 1));
    if (scaled) { S.valS.resize(nzs); S.diagS.resize(std::max(npos, 1)); }
    if (npos > 0)
        hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(npos)), dim3(kBlock), 0, s, npos, S.order.get(),
                           S.ptr.get(), W.rp.get(), W.ri.get(), W.rx.get(), scaled ? W.rxS.get() : nullptr,
                           W.dgn.get(), scaled ? W.dgnS.get() : nullptr, S.idx.get(), S.val.get(),
                           scaled ? S.valS.get() : nullptr, S.diag.get(), scaled ? S.diagS.get() : nullptr);
    // the O(#levels) numbers the host needs for the launch plan
    std::vector<unsigned char> level_long(nlev, 0);
    std::vector<int> lev_entry(nlev + 1, 0);
    W.tmpi.resize((size_t)nlev + 1);
    hipLaunchKernelGGL(gather_int_kernel, dim3(grid_for(nlev + 1)), dim3(kBlock), 0, s, nlev + 1, S.ptr.get(),
                       S.level_ptr_dev.get(), W.tmpi.get());
    W.tmpi.download(lev_entry.data(), (size_t)nlev + 1, s);
    if (nlev > 0) W.level_long.download(level_long.data(), (size_t)nlev, s);
    IPXK_HIP(hipStreamSynchronize(s));
    const int ntslot = plan_sweep(S, lptr, level_long, lev_entry);
    S.tslot.resize((size_t)std::max(ntslot, 1));
    for (const Sweep::Launch& L : S.plan)
        if (L.tail && L.ne > 0)
            hipLaunchKernelGGL(tslot_kernel, dim3(grid_for(L.ne)), dim3(kBlock), 0, s, L.ne, L.e0, lptr[L.l0],
                               lptr[L.l1], S.idx.get(), W.posof.get(), S.tslot.get() + L.tslot_off);
    S.chunk_long.upload(sweep_chunk_flags(lptr, level_long), s);
    IPXK_HIP(hipStreamSynchronize(s));
}

}  // namespace

void analyse_sweeps_device(Context* c, SplitOperator* S, const ipxint* Lp, const ipxint* Li, const double* Lx,
                           const ipxint* Up, const ipxint* Ui, const double* Ux,
                           const std::vector<double>& uscale) {
    hipStream_t s = c->stream;
    const int m = S->m;
    const int64_t nzL = Lp[m], nzU = Up[m], nzUo = nzU - m;
    const bool verbose = getenv("IPXK_VERBOSE") != nullptr;
    auto now = [&] {
        if (verbose) (void)hipStreamSynchronize(s);
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    const double t0 = now();
    // factors as given
    DevBuf<ipxint> dLp, dLi, dUp, dUi;
    DevBuf<double> dLx, dUx, dscale;
    dLp.upload(Lp, (size_t)m + 1, s); dUp.upload(Up, (size_t)m + 1, s);
    dLi.upload(Li, (size_t)nzL, s);   dLx.upload(Lx, (size_t)nzL, s);
    dUi.upload(Ui, (size_t)nzU, s);   dUx.upload(Ux, (size_t)nzU, s);
    dscale.upload(uscale, s);
    if (m > 0) {
        // column pointers were checked on the host (monotone, totals); indices are checked here
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
    Scratch W;
    const size_t maxnz = (size_t)std::max<int64_t>(std::max(nzL, nzU), 1);
    W.rp.resize((size_t)m + 1); W.ri.resize(maxnz); W.rx.resize(maxnz); W.rxS.resize(maxnz);
    W.dgn.resize(std::max(m, 1)); W.dgnS.resize(std::max(m, 1));
    W.keys.resize(std::max<size_t>(maxnz, (size_t)m + 1)); W.vals.resize(W.keys.size());
    W.keys2.resize(W.keys.size()); W.vals2.resize(W.keys.size()); W.colof.resize(maxnz);
    const int g = grid_for(m);
    const int rowbits = bits_for(std::max(m, 2));

    // --- U' sweep: unknown k gathers the rows above the diagonal of column k, ascending
    if (m > 0)
        hipLaunchKernelGGL(ut_rows_kernel, dim3(g), dim3(kBlock), 0, s, m, dUp.get(), dUi.get(), dUx.get(),
                           dscale.get(), W.rp.get(), W.ri.get(), W.rx.get(), W.rxS.get(), W.dgn.get(), W.dgnS.get());
    else IPXK_HIP(hipMemsetAsync(W.rp.get(), 0, sizeof(int), s));
    finish_sweep(c, W, S->Ut, m, nzUo, true, false, true, [&](std::vector<int>& lv) {
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
    finish_sweep(c, W, S->Lt, m, nzL, false, false, false, [&](std::vector<int>& lv) {
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
                           W.colof.get(), dLx.get(), (const double*)nullptr, W.ri.get(), W.rx.get(),
                           (double*)nullptr);
    finish_sweep(c, W, S->Lf, m, nzL, true, true, false, [&](std::vector<int>& lv) {
        for (int j = 0; j < m; j++)                         // column j is final when reached: push to rows i > j
            for (ipxint q = Lp[j]; q < Lp[j + 1]; q++) lv[Li[q]] = std::max(lv[Li[q]], lv[j] + 1);
    });

    // --- U sweep: row i of U without the diagonal, DESCENDING column order (sparse_matrix.cc:267-281)
    if (m > 0) {
        hipLaunchKernelGGL(uf_keys_kernel, dim3(g), dim3(kBlock), 0, s, m, dUp.get(), dUi.get(), W.keys.get(),
                           W.vals.get(), W.colof.get());
        if (nzUo > 0) sort_pairs(W, nzUo, rowbits, s);
        hipLaunchKernelGGL(u_diag_kernel, dim3(g), dim3(kBlock), 0, s, m, dUp.get(), dUx.get(), dscale.get(),
                           W.dgn.get(), W.dgnS.get());
    }
    hipLaunchKernelGGL(lower_bound_kernel, dim3(grid_for(m + 1)), dim3(kBlock), 0, s, m, nzUo, W.keys2.get(),
                       W.rp.get());
    if (nzUo > 0)
        hipLaunchKernelGGL(rows_from_perm_kernel, dim3(grid_for(nzUo)), dim3(kBlock), 0, s, nzUo, W.vals2.get(),
                           W.colof.get(), dUx.get(), dscale.get(), W.ri.get(), W.rx.get(), W.rxS.get());
    finish_sweep(c, W, S->Uf, m, nzUo, false, true, true, [&](std::vector<int>& lv) {
        for (int j = m - 1; j >= 0; j--)                    // descending: push to rows i < j
            for (ipxint q = Up[j]; q < Up[j + 1] - 1; q++) lv[Ui[q]] = std::max(lv[Ui[q]], lv[j] + 1);
    });
    IPXK_HIP(hipStreamSynchronize(s));
    if (verbose)
        fprintf(stderr, "ipxk: device analysis: upload of L, U %.1f ms (%.0f MB), four sweeps %.1f ms\n", (t1 - t0) * 1e3,
                ((double)(nzL + nzU) * 16 + (double)m * 24) / 1e6, (now() - t1) * 1e3);
    IPXK_HIP(hipGetLastError());
}

}  // namespace ipxk
