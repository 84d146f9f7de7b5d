// DiagonalPrecond on the device                reference src/diagonal_precond.cc
//   Factorize (:17-111): diagonal = W_I + sum_j W_j a_ij^2 over non-dense columns as a
//     row-gather SpMV; with dense columns the k x k Schur complement
//     S = inv(Wd) + Ad' inv(E) Ad is assembled column by column with the same SpMV
//     kernel and factorized by an in-library dense Cholesky (the GPU box has no LAPACK).
//   _Apply (:121-159): lhs = rhs ./ diagonal, or the Sherman-Morrison-Woodbury form.
#include "context.hpp"
#include "spmv_kernels.hpp"

namespace ipxk {

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// lhs = rhs ./ diag, partial[block] = sum lhs*rhs            (:151-155)
__global__ __launch_bounds__(kBlock) void diag_apply_kernel(int m, const double* __restrict__ rhs,
                                                            const double* __restrict__ diag,
                                                            double* __restrict__ lhs,
                                                            double* partial, const int* done) {
    if (done && *done) return;
    __shared__ double red[kBlock / 64 + 1];
    double acc = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) {
        const double r = rhs[i];
        const double l = r / diag[i];
        lhs[i] = l;
        acc += l * r;
    }
    acc = block_reduce<SumOp>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// u = rhs ./ diag                                            (:136)
__global__ void divide_kernel(int m, const double* __restrict__ rhs, const double* __restrict__ diag,
                              double* __restrict__ u, const int* done) {
    if (done && *done) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
        u[i] = rhs[i] / diag[i];
}

__global__ void copy_mask_kernel(int n, const double* __restrict__ W,
                                 const unsigned char* __restrict__ is_dense, double* __restrict__ out) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        out[j] = is_dense[j] ? 0.0 : W[j];
}

// u[Ai[p]] = set ? Ax[p]/diag[Ai[p]] : 0 over one column         (:76-78)
__global__ void column_scatter_kernel(int p0, int p1, const int* __restrict__ idx,
                                      const double* __restrict__ val,
                                      const double* __restrict__ diag, double* __restrict__ u,
                                      int set) {
    for (int p = p0 + blockIdx.x * blockDim.x + threadIdx.x; p < p1; p += gridDim.x * blockDim.x) {
        const int i = idx[p];
        u[i] = set ? val[p] / diag[i] : 0.0;
    }
}

// ---- blocked Schur complement ------------------------------------------------------------------
// S = Ad' inv(E) Ad for ALL pairs of dense columns in one pass over a dense copy of Ad (m x k, column major):
// a 64 x 64 tile of S per workgroup and slab of rows, partial tiles added in slab order afterwards (fixed order,
// no atomics).  The reference forms S one column at a time with a scatter and k sparse dot products
// (src/diagonal_precond.cc:68-85); every term is the same product a_ri * (a_rj / e_r), only the order of the sum
// differs (by slabs) -- rounding-level differences, like LAPACK's blocked dpotrf that follows.
__global__ void dense_panel_fill_kernel(int k, int64_t m, const int* __restrict__ colptr, const int* __restrict__ idx,
                                        const double* __restrict__ val, double* __restrict__ P) {
    const int kk = blockIdx.y;
    for (int p = colptr[kk] + blockIdx.x * blockDim.x + threadIdx.x; p < colptr[kk + 1]; p += gridDim.x * blockDim.x)
        P[(size_t)kk * m + idx[p]] = val[p];
}
constexpr int kSchurTile = 64, kSchurRows = 32;
__global__ __launch_bounds__(kBlock) void schur_gemm_kernel(int k, int64_t m, int64_t rows_per_slab, const double* __restrict__ P,
                                                            const double* __restrict__ diag, double* __restrict__ Spart) {
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (ti < tj) return;                                  // lower triangle of tiles
    __shared__ double As[kSchurRows][kSchurTile + 1], Bs[kSchurRows][kSchurTile + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;     // thread -> 4 x 4 outputs (rows 4*ty.., cols 4*tx..)
    const int64_t r0 = (int64_t)blockIdx.z * rows_per_slab, r1 = std::min<int64_t>(m, r0 + rows_per_slab);
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
    const int lr = tid & (kSchurRows - 1), lc0 = tid / kSchurRows;  // loader: row lr of the chunk, columns lc0, lc0 + 8, ...
    for (int64_t r = r0; r < r1; r += kSchurRows) {
        const int64_t rr = r + lr;
        const double d = rr < r1 ? diag[rr] : 1.0;
#pragma unroll
        for (int c = lc0; c < kSchurTile; c += kBlock / kSchurRows) {
            const int ci = ti * kSchurTile + c, cj = tj * kSchurTile + c;
            As[lr][c] = (rr < r1 && ci < k) ? P[(size_t)ci * m + rr] : 0.0;
            Bs[lr][c] = (rr < r1 && cj < k) ? P[(size_t)cj * m + rr] / d : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int q = 0; q < kSchurRows; q++) {
            double a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { a[t] = As[q][4 * ty + t]; b[t] = Bs[q][4 * tx + t]; }
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = 0; y < 4; y++) acc[x][y] += a[x] * b[y];
        }
        __syncthreads();
    }
    double* out = Spart + (size_t)blockIdx.z * k * k;
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const int i = ti * kSchurTile + 4 * ty + x, j = tj * kSchurTile + 4 * tx + y;
            if (i < k && j < k && i >= j) out[i + (size_t)j * k] = acc[x][y];
        }
}
// S[i,j] = S[j,i] = sum over the slabs, in slab order
__global__ void schur_reduce_kernel(int k, int nslabs, const double* __restrict__ Spart, double* __restrict__ S) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)k * k; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % k), j = (int)(e / k);
        if (i < j) continue;
        double sum = 0.0;
        for (int z = 0; z < nslabs; z++) sum += Spart[(size_t)z * k * k + e];
        S[i + (size_t)j * k] = sum;
        S[j + (size_t)i * k] = sum;
    }
}

// S[c,c] += 1/W[dense_col[c]]                                     (:82-83)
__global__ void schur_add_diag_kernel(int k, const int* __restrict__ dense_col,
                                      const double* __restrict__ W, double* __restrict__ S) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < k) S[c + (size_t)c * k] += 1.0 / W[dense_col[c]];
}

// Dense Cholesky A = L L' of the lower triangle, column major, one workgroup,
// left-looking by columns (the arithmetic of LAPACK's unblocked dpotf2):
// info = 0 or the 1-based index of the first non-positive pivot.
__global__ __launch_bounds__(1024) void cholesky_lower_kernel(int k, double* __restrict__ a,
                                                              int* info) {
    __shared__ int fail;
    __shared__ double pivot;
    if (threadIdx.x == 0) fail = 0;
    __syncthreads();
    for (int j = 0; j < k; j++) {
        // a[i,j] -= sum_{l<j} a[i,l]*a[j,l] for i >= j
        for (int i = j + threadIdx.x; i < k; i += blockDim.x) {
            double s = a[i + (size_t)j * k];
            for (int l = 0; l < j; l++) s -= a[i + (size_t)l * k] * a[j + (size_t)l * k];
            a[i + (size_t)j * k] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double d = a[j + (size_t)j * k];
            if (!(d > 0.0)) fail = j + 1;
            else { pivot = sqrt(d); a[j + (size_t)j * k] = pivot; }
        }
        __syncthreads();
        if (fail) break;
        const double d = pivot;
        for (int i = j + 1 + threadIdx.x; i < k; i += blockDim.x) a[i + (size_t)j * k] /= d;
        __syncthreads();
    }
    if (threadIdx.x == 0) *info = fail;
}

// Blocked right-looking variant for k > 64 (up to the 1000 dense columns src/model.cc:52-55 allows): panels of
// kCholPanel columns factorized by one workgroup, the trailing lower triangle updated by a grid of 64 x 64 tiles.
// An entry still receives its products one at a time in ascending l, rounded before they are subtracted, i.e.
// the arithmetic of cholesky_lower_kernel bit for bit.
constexpr int kCholPanel = 32;
__global__ __launch_bounds__(1024) void cholesky_panel_kernel(int k, int j0, int j1, double* __restrict__ a, int* info) {
    __shared__ int fail;
    __shared__ double pivot;
    if (*info) return;
    if (threadIdx.x == 0) fail = 0;
    __syncthreads();
    for (int j = j0; j < j1; j++) {
        for (int i = j + threadIdx.x; i < k; i += blockDim.x) {
            double s = a[i + (size_t)j * k];
            for (int l = j0; l < j; l++) s -= a[i + (size_t)l * k] * a[j + (size_t)l * k];
            a[i + (size_t)j * k] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double d = a[j + (size_t)j * k];
            if (!(d > 0.0)) fail = j + 1;
            else { pivot = sqrt(d); a[j + (size_t)j * k] = pivot; }
        }
        __syncthreads();
        if (fail) break;
        const double d = pivot;
        for (int i = j + 1 + threadIdx.x; i < k; i += blockDim.x) a[i + (size_t)j * k] /= d;
        __syncthreads();
    }
    if (threadIdx.x == 0 && fail) *info = fail;
}
// a[i,j] -= sum_{l in panel} a[i,l] a[j,l] for j1 <= j <= i < k
__global__ __launch_bounds__(kBlock) void cholesky_trailing_kernel(int k, int j0, int j1, double* __restrict__ a, const int* info) {
    __shared__ double Li[kCholPanel][64], Lj[kCholPanel][64];
    if (*info) return;
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bj > bi) return;                                     // lower triangle of tiles
    const int i0 = j1 + bi * 64, c0 = j1 + bj * 64, np = j1 - j0;
    for (int e = threadIdx.x; e < kCholPanel * 64; e += kBlock) {
        const int l = e / 64, x = e % 64;
        Li[l][x] = (l < np && i0 + x < k) ? a[(i0 + x) + (size_t)(j0 + l) * k] : 0.0;
        Lj[l][x] = (l < np && c0 + x < k) ? a[(c0 + x) + (size_t)(j0 + l) * k] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int y = ty + 16 * b, j = c0 + y;
        if (j >= k) continue;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int x = tx + 16 * q, i = i0 + x;
            if (i >= k || i < j) continue;
            double acc = a[i + (size_t)j * k];
            for (int l = 0; l < np; l++) acc -= Li[l][x] * Lj[l][y];
            a[i + (size_t)j * k] = acc;
        }
    }
}
// inverses of the 64 x 64 diagonal blocks of the factor (lower triangular), for the blocked solves
__global__ __launch_bounds__(64) void cholesky_invert_blocks_kernel(int k, const double* __restrict__ a, double* __restrict__ inv) {
    __shared__ double L[64][65];
    const int b0 = blockIdx.x * 64, nb = min(64, k - b0), c = threadIdx.x;
    for (int l = 0; l < 64; l++) L[c][l] = (c < nb && l < nb) ? a[(b0 + c) + (size_t)(b0 + l) * k] : (c == l ? 1.0 : 0.0);
    __syncthreads();
    // column c of the inverse: L x = e_c
    double x[64];
#pragma unroll 1
    for (int i = 0; i < 64; i++) {
        double s = i == c ? 1.0 : 0.0;
        for (int l = c; l < i; l++) s -= L[i][l] * x[l];
        x[i] = i < c ? 0.0 : s / L[i][i];
    }
    double* out = inv + (size_t)blockIdx.x * 64 * 64;
    for (int i = 0; i < 64; i++) out[i + 64 * c] = x[i];       // column major
}

// Solves L L' x = b in place for k <= 64 inside one wavefront: the factor is staged in LDS once
// (coalesced), lane i owns b[i] and each elimination step broadcasts the newly fixed unknown with
// a shuffle -- no global-memory access on the 2k-step dependency chain.
__global__ __launch_bounds__(64) void potrs_wave_kernel(int k, const double* __restrict__ a,
                                                        double* __restrict__ b, const int* done) {
    if (done && *done) return;
    __shared__ double L[64 * 65];                  // column l at L[l*65 ...] (padded)
    const int i = threadIdx.x;
    for (int l = 0; l < k; l++)
        if (i < k) L[l * 65 + i] = a[i + (size_t)l * k];
    __syncthreads();
    double bi = i < k ? b[i] : 0.0;
    for (int l = 0; l < k; l++) {                  // L z = b
        const double zl = __shfl(bi, l, 64) / L[l * 65 + l];
        if (i == l) bi = zl;
        else if (i > l && i < k) bi -= L[l * 65 + i] * zl;
    }
    for (int l = k - 1; l >= 0; l--) {             // L' x = z
        const double xl = __shfl(bi, l, 64) / L[l * 65 + l];
        if (i == l) bi = xl;
        else if (i < l) bi -= L[i * 65 + l] * xl;
    }
    if (i < k) b[i] = bi;
}

// 64 < k <= 1000 with the inverted diagonal blocks: per block one 64 x 64 product and one update of the rows below
// (above, for L'), all inside one workgroup with x in LDS -- 2 k / 64 dependent steps instead of 2 k.
__global__ __launch_bounds__(1024) void potrs_blocked_kernel(int k, const double* __restrict__ a, const double* __restrict__ inv,
                                                             double* __restrict__ b, const int* done) {
    if (done && *done) return;
    extern __shared__ double xs[];       // k + 64
    double* x = xs;
    double* xb = xs + k;
    const int nblk = (k + 63) / 64, tid = threadIdx.x;
    for (int i = tid; i < k; i += blockDim.x) x[i] = b[i];
    __syncthreads();
    for (int bq = 0; bq < nblk; bq++) {                        // L z = b
        const int b0 = bq * 64, nb = min(64, k - b0);
        const double* Ib = inv + (size_t)bq * 64 * 64;
        {   // 16 threads per row of the inverted block, partial sums combined in a fixed order
            const int r = tid >> 4, q = tid & 15;
            double s = 0.0;
            if (r < nb) for (int l = q; l <= r; l += 16) s += Ib[r + 64 * l] * x[b0 + l];
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
            if (q == 0) xb[r] = s;
        }
        __syncthreads();
        if (tid < nb) x[b0 + tid] = xb[tid];
        for (int i = b0 + nb + tid; i < k; i += blockDim.x) {
            double s = x[i];
            const double* ai = a + i + (size_t)b0 * k;
            int l = 0;
            for (; l + 8 <= nb; l += 8) {                     // the 8 loads are issued together
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = ai[(size_t)(l + u) * k];
#pragma unroll
                for (int u = 0; u < 8; u++) s -= v[u] * xb[l + u];
            }
            for (; l < nb; l++) s -= ai[(size_t)l * k] * xb[l];
            x[i] = s;
        }
        __syncthreads();
    }
    for (int bq = nblk - 1; bq >= 0; bq--) {                   // L' x = z
        const int b0 = bq * 64, nb = min(64, k - b0);
        const double* Ib = inv + (size_t)bq * 64 * 64;
        {
            const int r = tid >> 4, q = tid & 15;
            double s = 0.0;
            if (r < nb) for (int l = r + q; l < nb; l += 16) s += Ib[l + 64 * r] * x[b0 + l];    // inverse transposed
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
            if (q == 0) xb[r] = s;
        }
        __syncthreads();
        if (tid < nb) x[b0 + tid] = xb[tid];
        for (int i = tid; i < b0; i += blockDim.x) {
            double s = x[i];
            const double* ai = a + b0 + (size_t)i * k;        // row b0.. of column i: contiguous
            int l = 0;
            for (; l + 8 <= nb; l += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = ai[l + u];
#pragma unroll
                for (int u = 0; u < 8; u++) s -= v[u] * xb[l + u];
            }
            for (; l < nb; l++) s -= ai[l] * xb[l];
            x[i] = s;
        }
        __syncthreads();
    }
    for (int i = tid; i < k; i += blockDim.x) b[i] = x[i];
}

static int vec_grid(int64_t len) {
    int64_t g = (len + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return (int)(g < 1024 ? g : 1024);
}

// ---------------------------------------------------------------------------
// Factorize
// ---------------------------------------------------------------------------
static void build_dense_structures(Context* c) {
    // Ad as "rows = dense columns" (CopyColumns) and its transpose Atdense
    // (src/diagonal_precond.cc:59-65), both host-built once per model.
    const int64_t k = (int64_t)c->dense_cols.size(), m = c->m;
    if (c->AdCols.nrows == k && c->AdRows.nrows == m && k > 0) return;
    std::vector<ipxint> Cp, Ci;
    std::vector<double> Cx;
    fetch_columns(c, c->dense_cols, Cp, Ci, Cx);     // from the device's plain copy of the model
    // transpose (counting sort, ascending dense-column position within a row)
    std::vector<ipxint> Tp(m + 1, 0), Ti(Ci.size());
    std::vector<double> Tx(Ci.size());
    for (ipxint i : Ci) Tp[i + 1]++;
    for (int64_t i = 0; i < m; i++) Tp[i + 1] += Tp[i];
    std::vector<ipxint> next(Tp.begin(), Tp.end() - 1);
    for (int64_t kk = 0; kk < k; kk++)
        for (ipxint p = Cp[kk]; p < Cp[kk + 1]; p++) {
            const ipxint put = next[Ci[p]]++;
            Ti[put] = kk;
            Tx[put] = Cx[p];
        }
    c->AdCols.keep_plain = true;
    c->AdCols.tune_level = 0;           // k long rows: the long-row kernels do all the work in any layout
    c->AdRows.tune_level = 1;           // m short rows gathering from k numbers: phased against fused
    c->AdCols.build(k, m, Cp.data(), Ci.data(), Cx.data(), c->stream);
    c->AdRows.build(m, k, Tp.data(), Ti.data(), Tx.data(), c->stream);
    c->chol.resize((size_t)k * k);
    c->smw_work.resize(k);
    c->smw_u.resize(m);
    c->chol_info.resize(1);
}

// out[i] = w[i] + out[i]
__global__ void add_vector_kernel(int len, const double* __restrict__ w, double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x)
        out[i] = w[i] + out[i];
}

// Everything of the dense-column machinery that depends on the model alone, built when the model is (build_model): the
// first Factorize then costs what the others do (C5: 56 ms -> see DESIGN section 5).
void prepare_dense_columns(Context* c) {
    if (c->num_dense <= 0 || comm_active(c)) return;
    const int64_t m = c->m, n = c->n;
    const int k = (int)c->dense_cols.size();
    hipStream_t s = c->stream;
    build_dense_structures(c);
    std::vector<unsigned char> mask(n, 0);
    for (ipxint j : c->dense_cols) mask[j] = 1;
    c->dense_mask.upload(mask, s);
    c->Wnodense.resize(n);
    c->diagonal.resize(m);
    if ((int64_t)m * k * 8 <= (int64_t(8) << 30)) c->schur_panel.ensure((size_t)m * k);
    IPXK_HIP(hipStreamSynchronize(s));
}

void diag_factorize_dev(Context* c, const double* W, bool precond_dense_cols, ipxint* errflag) {
    const int64_t m = c->m, n = c->n;
    hipStream_t s = c->stream;
    *errflag = 0;
    c->diag_factorized = false;
    c->diagonal.resize(m);
    const bool smw = precond_dense_cols && c->num_dense > 0;
    if (smw && comm_active(c))
        throw Error(IPXK_E_UNSUPPORTED, "dense-column (SMW) preconditioning is not available on a partitioned "
                                        "system: pass precond_dense_cols = 0");
    const double* Wcols = W;
    if (smw) {
        c->Wnodense.resize(n);
        if (c->dense_mask.size() != (size_t)n) {
            std::vector<unsigned char> mask(n, 0);
            for (ipxint j : c->dense_cols) mask[j] = 1;
            c->dense_mask.upload(mask, s);
            IPXK_HIP(hipStreamSynchronize(s));
        }
        hipLaunchKernelGGL(copy_mask_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, s, (int)n, W,
                           c->dense_mask.get(), c->Wnodense.get());
        Wcols = c->Wnodense.get();
    }
    // :28-46
    if (comm_cols(c)) {
        // sum over ranks of the local columns' contributions, then the slack weights
        EpiDiagonal ed{{}, nullptr, c->diagonal.get()};
        launch_spmv(c->Arows, Wcols, ed, nullptr, nullptr, s);
        comm_allreduce_sum(c, c->diagonal.get(), (size_t)m);
        hipLaunchKernelGGL(add_vector_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, s, (int)m, W + n,
                           c->diagonal.get());
    } else {
        EpiDiagonal ed{{}, W + n, c->diagonal.get()};
        launch_spmv(c->Arows, Wcols, ed, nullptr, nullptr, s);
    }
    c->kdense = 0;
    if (smw) {
        build_dense_structures(c);
        const int k = (int)c->dense_cols.size();
        // :68-85 Schur complement, one column per dense column
        IPXK_HIP(hipMemsetAsync(c->smw_u.get(), 0, sizeof(double) * m, s));
        std::vector<int> dc(k);
        for (int kk = 0; kk < k; kk++) dc[kk] = (int)c->dense_cols[kk];
        DevBuf<int> dcols;
        dcols.upload(dc, s);
        const bool blocked = (int64_t)m * k * 8 <= (int64_t(8) << 30) && !(getenv("IPXK_SCHUR_BLOCKED") && getenv("IPXK_SCHUR_BLOCKED")[0] == '0');
        if (blocked) {
            // all k^2 entries in one pass over a dense copy of the dense columns
            DevBuf<double>& P = c->schur_panel;
            P.ensure((size_t)m * k);
            IPXK_HIP(hipMemsetAsync(P.get(), 0, (size_t)m * k * sizeof(double), s));
            if (c->schur_colptr.size() != (size_t)k + 1) {
                c->schur_colptr.upload(c->AdCols.h_plain_ptr, s);
                IPXK_HIP(hipStreamSynchronize(s));
            }
            hipLaunchKernelGGL(dense_panel_fill_kernel, dim3(64, k), dim3(kBlock), 0, s, k, m, c->schur_colptr.get(),
                               c->AdCols.plain_idx.get(), c->AdCols.plain_val.get(), P.get());
            const int nt = (k + kSchurTile - 1) / kSchurTile;
            const int64_t ntri = (int64_t)nt * (nt + 1) / 2;
            int nslabs = (int)std::max<int64_t>(1, std::min<int64_t>(256, 2048 / ntri));
            nslabs = (int)std::min<int64_t>(nslabs, std::max<int64_t>(1, m / 256));
            const int64_t per = ((m + nslabs - 1) / nslabs + kSchurRows - 1) / kSchurRows * kSchurRows;
            nslabs = (int)((m + per - 1) / per);
            c->schur_part.ensure((size_t)nslabs * k * k);
            hipLaunchKernelGGL(schur_gemm_kernel, dim3(nt, nt, nslabs), dim3(kBlock), 0, s, k, m, per, P.get(), c->diagonal.get(),
                               c->schur_part.get());
            hipLaunchKernelGGL(schur_reduce_kernel, dim3(vec_grid((int64_t)k * k)), dim3(kBlock), 0, s, k, nslabs, c->schur_part.get(),
                               c->chol.get());
        }
        for (int kk = 0; kk < k && !blocked; kk++) {
            const int p0 = c->AdCols.h_plain_ptr[kk], p1 = c->AdCols.h_plain_ptr[kk + 1];
            const int g = vec_grid(p1 - p0);
            hipLaunchKernelGGL(column_scatter_kernel, dim3(g), dim3(kBlock), 0, s, p0, p1,
                               c->AdCols.plain_idx.get(), c->AdCols.plain_val.get(), c->diagonal.get(),
                               c->smw_u.get(), 1);
            EpiScale es{{}, nullptr, c->chol.get() + (size_t)kk * k};
            launch_spmv(c->AdCols, c->smw_u.get(), es, nullptr, nullptr, s);
            hipLaunchKernelGGL(column_scatter_kernel, dim3(g), dim3(kBlock), 0, s, p0, p1,
                               c->AdCols.plain_idx.get(), c->AdCols.plain_val.get(), c->diagonal.get(),
                               c->smw_u.get(), 0);
        }
        hipLaunchKernelGGL(schur_add_diag_kernel, dim3((k + 63) / 64), dim3(64), 0, s, k, dcols.get(),
                           W, c->chol.get());
        // :88-92
        if (k <= 64) {
            hipLaunchKernelGGL(cholesky_lower_kernel, dim3(1), dim3(64), 0, s, k, c->chol.get(), c->chol_info.get());
        } else {
            IPXK_HIP(hipMemsetAsync(c->chol_info.get(), 0, sizeof(int), s));
            for (int j0 = 0; j0 < k; j0 += kCholPanel) {
                const int j1 = std::min(k, j0 + kCholPanel);
                hipLaunchKernelGGL(cholesky_panel_kernel, dim3(1), dim3(1024), 0, s, k, j0, j1, c->chol.get(), c->chol_info.get());
                if (j1 < k) {
                    const int nt = (k - j1 + 63) / 64;
                    hipLaunchKernelGGL(cholesky_trailing_kernel, dim3(nt, nt), dim3(kBlock), 0, s, k, j0, j1, c->chol.get(),
                                       c->chol_info.get());
                }
            }
            c->chol_inv.ensure((size_t)((k + 63) / 64) * 64 * 64);
            hipLaunchKernelGGL(cholesky_invert_blocks_kernel, dim3((k + 63) / 64), dim3(64), 0, s, k, c->chol.get(), c->chol_inv.get());
        }
        int info = 0;
        c->chol_info.download(&info, 1, s);
        IPXK_HIP(hipStreamSynchronize(s));
        if (info != 0) {
            *errflag = 401;  // IPX_ERROR_lapack_chol
            return;
        }
        c->kdense = k;
    }
    IPXK_HIP(hipGetLastError());
    c->diag_factorized = true;
}

// ---------------------------------------------------------------------------
// Apply
// ---------------------------------------------------------------------------
int diag_apply_dev(Context* c, const double* rhs, double* lhs, int slot, const int* done) {
    const int m = (int)c->m;
    hipStream_t s = c->stream;
    time_mark(c, kTimePrecond, true);
    struct EndMark { Context* c; ~EndMark() { time_mark(c, kTimePrecond, false); } } end_mark{c};
    if (c->kdense == 0) {
        const int g = vec_grid(m);
        hipLaunchKernelGGL(diag_apply_kernel, dim3(g), dim3(kBlock), 0, s, m, rhs,
                           c->reord.in_use ? c->reord.diagonal.get() : c->diagonal.get(), lhs, c->part(slot), done);
        return g;
    }
    const int k = (int)c->kdense;
    // :135-137 work = Ad' * (rhs ./ diagonal)
    hipLaunchKernelGGL(divide_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, s, m, rhs,
                       c->diagonal.get(), c->smw_u.get(), done);
    EpiScale es{{}, nullptr, c->smw_work.get()};
    launch_spmv(c->AdCols, c->smw_u.get(), es, nullptr, done, s);
    // :140-141
    if (k <= 64)
        hipLaunchKernelGGL(potrs_wave_kernel, dim3(1), dim3(64), 0, s, k, c->chol.get(),
                           c->smw_work.get(), done);
    else
        hipLaunchKernelGGL(potrs_blocked_kernel, dim3(1), dim3(1024), sizeof(double) * (k + 64), s, k,
                           c->chol.get(), c->chol_inv.get(), c->smw_work.get(), done);
    // :145-149
    EpiSmwRows er{{}, rhs, c->diagonal.get(), lhs};
    return launch_spmv(c->AdRows, c->smw_work.get(), er, c->part(slot), done, s);
}

}  // namespace ipxk
