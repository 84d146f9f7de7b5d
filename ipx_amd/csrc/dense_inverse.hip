// The inverse of a dense LU-factored block on the matrix cores (round 4; replaces the rocBLAS call of round 3).
//   D22 = (L22 + I) U22   (kb x kb, column major: U22 on and above the diagonal, L22 below -- LAPACK's LU storage)
//   inverse(D22) = inverse(U22) * inverse(L22 + I)
// Both triangular inverses by recursive doubling from the inverted 64 x 64 diagonal blocks (bump_invert_blocks_kernel,
// trisolve.hip):   [A 0; B C]^-1 = [A^-1 0; -C^-1 B A^-1, C^-1],   [A B; 0 C]^-1 = [A^-1, -A^-1 B C^-1; 0, C^-1]
// -- per level two batched products over all pairs of blocks, then one product of the two triangular inverses.  Every
// product is one launch of a tiled fp64 GEMM on v_mfma_f64_16x16x4_f64 that skips the k-tiles a triangular operand
// makes zero.  Measured on the MI355X: see profiles/r04_dense_inverse.txt (round 3: own kernel 134 ms / rocBLAS 4.8 ms
// at 4096 rows).  The result is probed like every explicit inverse (trisolve.hip: the guard) before it is used.
#include "context.hpp"

namespace ipxk {

namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int kT = 64;        // tile of C per workgroup
constexpr int kK = 16;        // k-step staged in LDS

// which k-tiles contribute to the C tile at (i0, j0): a triangular operand is zero elsewhere
enum KMode { kFull = 0, kBLower = 1, kALower = 2, kAUpper = 3, kBUpper = 4, kAUpperBLower = 5 };

// C (M x N) = alpha * A (M x K) * B (K x N), column major, all dimensions multiples of 64; blockIdx.z = batch (strides in elements).
// The TRANSPOSED product is formed on the matrix unit (its A operand = a row of B', its B operand = a row of A'), so that the lane
// index of a result runs along the rows of C: the stores of a tile are 128-byte segments of C's columns.
__global__ __launch_bounds__(256) void gemm_f64_kernel(int K, double alpha, const double* __restrict__ A, int lda, int64_t sA,
                                                       const double* __restrict__ B, int ldb, int64_t sB, double* __restrict__ C, int ldc,
                                                       int64_t sC, int kmode) {
    __shared__ double As[kK][kT + 1];
    __shared__ double Bs[kK][kT + 1];
    const int i0 = blockIdx.x * kT, j0 = blockIdx.y * kT;
    A += (int64_t)blockIdx.z * sA; B += (int64_t)blockIdx.z * sB; C += (int64_t)blockIdx.z * sC;
    int kbeg = 0, kend = K;
    if (kmode == kBLower) kbeg = j0;
    else if (kmode == kALower) kend = min(K, i0 + kT);
    else if (kmode == kAUpper) kbeg = i0;
    else if (kmode == kBUpper) kend = min(K, j0 + kT);
    else if (kmode == kAUpperBLower) kbeg = max(i0, j0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    d4_t acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; jt++) acc[jt] = d4_t{0.0, 0.0, 0.0, 0.0};
    for (int kt = kbeg; kt < kend; kt += kK) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int kk = (tid >> 6) + 4 * s;
            As[kk][tid & 63] = A[(int64_t)(kt + kk) * lda + i0 + (tid & 63)];
            const int j = (tid >> 4) + 16 * s;
            Bs[tid & 15][j] = B[(int64_t)(j0 + j) * ldb + kt + (tid & 15)];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < kK / 4; ks++) {
            const double a = As[4 * ks + lk][16 * wave + li];
#pragma unroll
            for (int jt = 0; jt < 4; jt++)
                acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bs[4 * ks + lk][16 * jt + li], a, acc[jt], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int jt = 0; jt < 4; jt++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            C[(int64_t)(j0 + 16 * jt + lk + 4 * q) * ldc + i0 + 16 * wave + li] = alpha * acc[jt][q];
}

// Lp = strictly lower part of D, Up = upper part with the diagonal, both n x n (n = kb rounded up to 64; the padding is zero,
// with a unit diagonal in Up); Li / Ui = the inverted 64 x 64 diagonal blocks on the block diagonal, zero elsewhere
__global__ void split_pad_kernel(int kb, int n, const double* __restrict__ D, const double* __restrict__ invL, const double* __restrict__ invU,
                                 double* __restrict__ Lp, double* __restrict__ Up, double* __restrict__ Li, double* __restrict__ Ui) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)n * n; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e % n), c = (int)(e / n);
        const bool in = r < kb && c < kb;
        const double d = in ? D[(size_t)c * kb + r] : 0.0;
        Lp[e] = in && r > c ? d : 0.0;
        Up[e] = in ? (r <= c ? d : 0.0) : (r == c ? 1.0 : 0.0);
        double li = 0.0, ui = 0.0;
        if (r / 64 == c / 64) {
            const size_t o = (size_t)(r / 64) * 64 * 64 + (size_t)(c % 64) * 64 + (r % 64);
            li = invL[o];
            ui = invU[o];
        }
        Li[e] = li;
        Ui[e] = ui;
    }
}
// X (kb x kb, column major) <- top left of P (n x n); Xt <- its transpose; 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void crop_transpose_kernel(int kb, int n, const double* __restrict__ P, double* __restrict__ X,
                                                             double* __restrict__ Xt) {
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int q = ty; q < 32; q += 8) {
        const int c = by + q, r = bx + tx;
        if (c < kb && r < kb) {
            const double v = P[(size_t)c * n + r];
            tile[q][tx] = v;
            X[(size_t)c * kb + r] = v;
        }
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int r = bx + q, c = by + tx;               // Xt[r][c] column major = X[c][r]:  Xt[(size_t)r * kb + c] = P[c * n + r]
        if (r < kb && c < kb) Xt[(size_t)r * kb + c] = tile[tx][q];
    }
}

// R = I - T1 - T2 (into T1);  P += Q
__global__ void refine_residual_kernel(int n, double* __restrict__ T1, const double* __restrict__ T2) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)n * n; e += (int64_t)gridDim.x * blockDim.x) {
        const double id = (e % n) == (e / n) ? 1.0 : 0.0;
        T1[e] = (id - T1[e]) - T2[e];
    }
}
__global__ void refine_add_kernel(int64_t nn, double* __restrict__ P, const double* __restrict__ Q) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nn; e += (int64_t)gridDim.x * blockDim.x) P[e] += Q[e];
}

void gemm(hipStream_t s, int M, int N, int K, double alpha, const double* A, int lda, int64_t sA, const double* B, int ldb, int64_t sB, double* C,
          int ldc, int64_t sC, int batch, int kmode) {
    if (M <= 0 || N <= 0 || batch <= 0) return;
    hipLaunchKernelGGL(gemm_f64_kernel, dim3(M / kT, N / kT, batch), dim3(256), 0, s, K, alpha, A, lda, sA, B, ldb, sB, C, ldc, sC, kmode);
}

}  // namespace

// X (kb x kb) <- inverse(D22), column major: X[j * kb + t] = inverse(D22)[t][j]; Xt <- its transpose (the same array read row
// major).  invL / invU: the inverted 64 x 64 diagonal blocks of L22 + I and U22 (column major, one after the other).
// refine: that many steps of X <- X + X (I - D22 X) afterwards (Newton-Schulz: the residual is squared per step).  The triangular
// inverses of an ill-conditioned block lose digits that the product cannot give back: on the bases of a 12 000 x 30 000 LP's
// IPM the probe |D22 (X z) - z| of the plain result is 1e-7 ... 3e-5 (the guard asks for 1e-8); one step brings it to 1e-12.
void dense_lu_inverse(Context* c, int kb, const double* D, const double* invL, const double* invU, double* X, double* Xt, int refine) {
    hipStream_t s = c->stream;
    const int n = (kb + 63) / 64 * 64;
    const size_t nn = (size_t)n * n;
    DevBuf<double>& W = c->dense_work;
    W.ensure((refine > 0 ? 7 : 5) * nn);
    double *Lp = W.get(), *Up = Lp + nn, *Li = Up + nn, *Ui = Li + nn, *T = Ui + nn;
    const int g = (int)std::min<int64_t>(8192, ((int64_t)nn + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(split_pad_kernel, dim3(g), dim3(kBlock), 0, s, kb, n, D, invL, invU, Lp, Up, Li, Ui);
    for (int h = 64; h < n; h *= 2) {
        // pairs of h x h blocks at offsets o = 2 p h; the last pair may have a shorter (or no) second block
        const int npairs = n / (2 * h), rest = n - npairs * 2 * h;          // rest in {0} or (h, 2h): one more pair with a short second block, or a lone block
        const int64_t stride = (int64_t)2 * h * ((int64_t)n + 1);           // from a pair's corner to the next pair's
        for (int part = 0; part < 2; part++) {
            const int batch = part == 0 ? npairs : (rest > h ? 1 : 0);
            const int h2 = part == 0 ? h : rest - h;
            const int64_t o = part == 0 ? 0 : (int64_t)npairs * 2 * h * ((int64_t)n + 1);
            if (batch == 0) continue;
            // lower:  T = B A^-1  (B = Lp[o+h.., o..], A^-1 = Li[o.., o..] lower),  Li[o+h.., o..] = -C^-1 T  (C^-1 = Li[o+h.., o+h..] lower)
            gemm(s, h2, h, h, 1.0, Lp + o + h, n, stride, Li + o, n, stride, T + o + h, n, stride, batch, kBLower);
            gemm(s, h2, h, h2, -1.0, Li + o + h + (int64_t)h * n, n, stride, T + o + h, n, stride, Li + o + h, n, stride, batch, kALower);
            // upper:  T = B C^-1  (B = Up[o.., o+h..], C^-1 = Ui[o+h.., o+h..] upper),  Ui[o.., o+h..] = -A^-1 T  (A^-1 = Ui[o.., o..] upper)
            gemm(s, h, h2, h2, 1.0, Up + o + (int64_t)h * n, n, stride, Ui + o + h + (int64_t)h * n, n, stride, T + o + (int64_t)h * n, n, stride, batch,
                 kBUpper);
            gemm(s, h, h2, h, -1.0, Ui + o, n, stride, T + o + (int64_t)h * n, n, stride, Ui + o + (int64_t)h * n, n, stride, batch, kAUpper);
        }
    }
    // inverse(D22) = inverse(U22) inverse(L22 + I): upper times lower
    gemm(s, n, n, n, 1.0, Ui, n, 0, Li, n, 0, T, n, 0, 1, kAUpperBLower);
    for (int r = 0; r < refine; r++) {
        // D22 X = (L22 + I) (U22 X):  T1 = U22 X,  T2 = L22 T1,  R = I - T1 - T2 (into T1),  T2 = X R,  X += T2
        double *T1 = T + nn, *T2 = T1 + nn;
        gemm(s, n, n, n, 1.0, Up, n, 0, T, n, 0, T1, n, 0, 1, kAUpper);
        gemm(s, n, n, n, 1.0, Lp, n, 0, T1, n, 0, T2, n, 0, 1, kALower);
        hipLaunchKernelGGL(refine_residual_kernel, dim3(g), dim3(kBlock), 0, s, n, T1, T2);
        gemm(s, n, n, n, 1.0, T, n, 0, T1, n, 0, T2, n, 0, 1, kFull);
        hipLaunchKernelGGL(refine_add_kernel, dim3(g), dim3(kBlock), 0, s, (int64_t)nn, T, T2);
    }
    const int nt = (kb + 31) / 32;
    hipLaunchKernelGGL(crop_transpose_kernel, dim3(nt, nt), dim3(256), 0, s, kb, n, T, X, Xt);
    IPXK_HIP(hipGetLastError());
}

}  // namespace ipxk
