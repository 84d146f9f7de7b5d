// The inverse of a dense LU-factored block through rocBLAS, loaded with dlopen like RCCL (comm.hip): a process that never
// meets a large dense block never loads it, and a box without librocblas keeps working with the library's own kernel.
//   D22 = (L22 + I) U22 (kb x kb, column major: U22 on and above the diagonal, L22 below -- LAPACK's LU storage)
//   X <- I;  X <- inverse(L22 + I) X;  X <- inverse(U22) X        (two rocblas_dtrsm)
// Measured on the MI355X (scripts/bench_rocblas_trsm.hip): 0.5 / 1.0 / 4.8 / 27.8 ms at 1024 / 2048 / 4096 / 8000 rows against
// 2 / 16 / 134 / 1000 ms for bump_inverse_kernel's one blocked solve per column (trisolve.hip); creating the handle costs 0.2 s
// once per context, the first call of a size class another 60 ms.  A plain library BLAS-3 call on an auxiliary dense block,
// outside every loop that is timed or profiled; the hot path has no library call.
#include <dlfcn.h>

#include <cstdlib>

#include "context.hpp"

namespace ipxk {

namespace {
// the four entry points used, with rocBLAS's own (C) signatures; enum values of rocblas-types.h
using handle_t = void*;
constexpr int kSideLeft = 141, kFillUpper = 121, kFillLower = 122, kOpNone = 111, kDiagNonUnit = 131, kDiagUnit = 132;
struct Blas {
    void* lib = nullptr;
    int (*create)(handle_t*) = nullptr;
    int (*destroy)(handle_t) = nullptr;
    int (*set_stream)(handle_t, hipStream_t) = nullptr;
    int (*dtrsm)(handle_t, int side, int uplo, int trans, int diag, int m, int n, const double* alpha, const double* A, int lda, double* B,
                 int ldb) = nullptr;
    bool ok = false;
};
Blas& blas() {
    static Blas b = [] {
        Blas r;
        if (const char* e = getenv("IPXK_ROCBLAS")) if (e[0] == '0') return r;
        r.lib = dlopen("librocblas.so.5", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) r.lib = dlopen("librocblas.so", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) return r;
        r.create = reinterpret_cast<decltype(r.create)>(dlsym(r.lib, "rocblas_create_handle"));
        r.destroy = reinterpret_cast<decltype(r.destroy)>(dlsym(r.lib, "rocblas_destroy_handle"));
        r.set_stream = reinterpret_cast<decltype(r.set_stream)>(dlsym(r.lib, "rocblas_set_stream"));
        r.dtrsm = reinterpret_cast<decltype(r.dtrsm)>(dlsym(r.lib, "rocblas_dtrsm"));
        r.ok = r.create && r.destroy && r.set_stream && r.dtrsm;
        return r;
    }();
    return b;
}
__global__ void identity_kernel(int kb, double* X) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)kb * kb; e += (int64_t)gridDim.x * blockDim.x)
        X[e] = e / kb == e % kb ? 1.0 : 0.0;
}
// out[c * kb + r] = in[r * kb + c], 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(int kb, const double* __restrict__ in, double* __restrict__ out) {
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int q = ty; q < 32; q += 8)
        if (by + q < kb && bx + tx < kb) tile[q][tx] = in[(size_t)(by + q) * kb + bx + tx];
    __syncthreads();
    for (int q = ty; q < 32; q += 8)
        if (bx + q < kb && by + tx < kb) out[(size_t)(bx + q) * kb + by + tx] = tile[tx][q];
}
}  // namespace

// X (kb x kb) <- inverse(D22), column major: X[j * kb + t] = inverse(D22)[t][j]; Xt <- its transpose (the same array read row
// major).  false: rocBLAS is not available (or disabled with IPXK_ROCBLAS=0) -- nothing was launched.
bool blas_lu_inverse(Context* c, int kb, const double* D, double* X, double* Xt) {
    Blas& b = blas();
    if (!b.ok) return false;
    if (!c->blas_handle) {
        handle_t h = nullptr;
        if (b.create(&h) != 0 || !h) return false;
        c->blas_handle = h;
    }
    hipStream_t s = c->stream;
    if (b.set_stream(c->blas_handle, s) != 0) return false;
    const int g = (int)std::min<int64_t>(4096, ((int64_t)kb * kb + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(identity_kernel, dim3(g), dim3(kBlock), 0, s, kb, X);
    const double one = 1.0;
    if (b.dtrsm(c->blas_handle, kSideLeft, kFillLower, kOpNone, kDiagUnit, kb, kb, &one, D, kb, X, kb) != 0 ||
        b.dtrsm(c->blas_handle, kSideLeft, kFillUpper, kOpNone, kDiagNonUnit, kb, kb, &one, D, kb, X, kb) != 0)
        throw Error(IPXK_E_HIP, "rocblas_dtrsm failed on the dense block of the factors");
    const int nt = (kb + 31) / 32;
    hipLaunchKernelGGL(transpose_kernel, dim3(nt, nt), dim3(256), 0, s, kb, X, Xt);
    IPXK_HIP(hipGetLastError());
    return true;
}

void blas_destroy(Context* c) {
    if (c->blas_handle && blas().ok) (void)blas().destroy(c->blas_handle);
    c->blas_handle = nullptr;
}

}  // namespace ipxk
