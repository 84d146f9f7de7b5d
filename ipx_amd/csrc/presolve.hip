// Model upload on the device: the two array transformations that define the
// solver-form matrix before the path starts,
//   Presolver::EquilibrateMatrix        reference src/presolver.cc:868-974
//   Transpose (AI -> AIt)               reference src/sparse_matrix.cc:120-151
// Both are exact arithmetic (maxima, powers of two, index permutations), so the results are bit-identical to
// the reference's whatever the order of evaluation.  Stand-alone entry points on host arrays (no context:
// the context is created FROM their output).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <cmath>

#include "internal.hpp"

namespace ipxk {

namespace {

int grid_for(int64_t n) { return (int)std::min<int64_t>(4096, std::max<int64_t>(1, (n + kBlock - 1) / kBlock)); }

#define IPXK_GRID_STRIDE(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

constexpr int kExpMin = 0, kExpMax = 3, kMaxRound = 10;     // src/presolver.cc:906-908

__device__ __forceinline__ double equilibration_factor(int exp) {   // :868-880
    if (exp < kExpMin) return ldexp(1.0, (kExpMin - exp + 1) / 2);
    if (exp > kExpMax) return ldexp(1.0, -((exp - kExpMax + 1) / 2));
    return 1.0;
}

// :912-924  is any entry outside [2^(expmin-1), 2^expmax) ?
__global__ void out_of_range_kernel(int64_t nz, const double* __restrict__ Ax, int* flag) {
    IPXK_GRID_STRIDE(p, nz) {
        int exp;
        frexp(fabs(Ax[p]), &exp);
        if (exp < kExpMin || exp > kExpMax) *flag = 1;
    }
}
// :934-944  infinity norm of each column and row.  A maximum of non-negative doubles is the maximum of
// their bit patterns as unsigned integers: exact and independent of the order, so an atomic max serves.
__global__ void col_row_max_kernel(int64_t n, const ipxint* __restrict__ Ap, const int* __restrict__ Ai,
                                   const double* __restrict__ Ax, double* __restrict__ colmax,
                                   unsigned long long* rowmax_bits) {
    IPXK_GRID_STRIDE(j, n) {
        double cm = 0.0;
        for (ipxint p = Ap[j]; p < Ap[j + 1]; p++) {
            const double xa = fabs(Ax[p]);
            cm = xa > cm ? xa : cm;
            atomicMax(rowmax_bits + Ai[p], (unsigned long long)__double_as_longlong(xa));
        }
        colmax[j] = cm;
    }
}
// :946-964  this round's factors; accumulates them into the scaling vectors
__global__ void factors_kernel(int64_t len, double* __restrict__ mx, double* __restrict__ scale, int* out_of_range) {
    IPXK_GRID_STRIDE(i, len) {
        int exp;
        frexp(mx[i], &exp);
        const double f = equilibration_factor(exp);
        mx[i] = f;
        if (f != 1.0) { *out_of_range = 1; scale[i] *= f; }
    }
}
// :967-972
__global__ void rescale_kernel(int64_t n, const ipxint* __restrict__ Ap, const int* __restrict__ Ai,
                               double* __restrict__ Ax, const double* __restrict__ colf, const double* __restrict__ rowf) {
    IPXK_GRID_STRIDE(j, n) {
        const double cf = colf[j];
        for (ipxint p = Ap[j]; p < Ap[j + 1]; p++) {
            double v = Ax[p];
            v *= cf;              // column scaling
            v *= rowf[Ai[p]];     // row scaling
            Ax[p] = v;
        }
    }
}

__global__ void fill_kernel(int64_t n, double v, double* __restrict__ x) { IPXK_GRID_STRIDE(i, n) x[i] = v; }
__global__ void narrow_index_kernel(int64_t nz, const ipxint* __restrict__ in, int* __restrict__ out, int limit, int* bad) {
    IPXK_GRID_STRIDE(p, nz) {
        const ipxint v = in[p];
        if (v < 0 || v >= limit) *bad = 1;
        out[p] = (int)v;
    }
}

// ---- transpose ----
// key = row of an entry, value = its position; the entries are enumerated column by column, so a STABLE sort
// by row leaves the entries of a row in ascending source-column order -- the order of the reference's
// counting sort (src/sparse_matrix.cc:120-151)
__global__ void entry_columns_kernel(int64_t n, const ipxint* __restrict__ Ap, int* __restrict__ colof, int* __restrict__ pos) {
    IPXK_GRID_STRIDE(j, n)
        for (ipxint p = Ap[j]; p < Ap[j + 1]; p++) { colof[p] = (int)j; pos[p] = (int)p; }
}
__global__ void row_pointers_kernel(int64_t m, int64_t nz, const int* __restrict__ sorted_rows, ipxint* __restrict__ Tp) {
    IPXK_GRID_STRIDE(i, m + 1) {
        int64_t lo = 0, hi = nz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (sorted_rows[mid] < (int)i) lo = mid + 1; else hi = mid;
        }
        Tp[i] = lo;
    }
}
__global__ void gather_transposed_kernel(int64_t nz, const int* __restrict__ perm, const int* __restrict__ colof,
                                         const double* __restrict__ Ax, ipxint* __restrict__ Ti, double* __restrict__ Tx) {
    IPXK_GRID_STRIDE(t, nz) {
        const int p = perm[t];
        Ti[t] = colof[p];
        Tx[t] = Ax[p];
    }
}

struct DeviceGuard {       // binds the device for the call, own stream
    hipStream_t s = nullptr;
    explicit DeviceGuard(int device) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count < 1) throw Error(IPXK_E_HIP, "no HIP device");
        IPXK_REQUIRE(device >= 0 && device < count, "device index out of range");
        IPXK_HIP(hipSetDevice(device));
        IPXK_HIP(hipStreamCreate(&s));
    }
    ~DeviceGuard() { if (s) (void)hipStreamDestroy(s); }
};

void check_csc(int64_t m, int64_t n, const ipxint* Ap) {
    IPXK_REQUIRE(m >= 0 && n >= 0 && m < (int64_t(1) << 31) - 1 && n < (int64_t(1) << 31) - 1, "dimension out of range");
    IPXK_REQUIRE(Ap[0] == 0, "colptr[0] must be 0");
    for (int64_t j = 0; j < n; j++) IPXK_REQUIRE(Ap[j] <= Ap[j + 1], "colptr not monotone");
    IPXK_REQUIRE(Ap[n] < (int64_t(1) << 31) - 1, "nnz exceeds 32-bit device indices");
}

}  // namespace

void equilibrate_device(int device, int64_t m, int64_t n, const ipxint* Ap, const ipxint* Ai, double* Ax,
                        double* colscale, double* rowscale, ipxint* rounds) {
    check_csc(m, n, Ap);
    DeviceGuard dev(device);
    hipStream_t s = dev.s;
    const int64_t nz = Ap[n];
    DevBuf<ipxint> dAp, dAi64;
    DevBuf<int> dAi((size_t)std::max<int64_t>(nz, 1)), flags(2);
    DevBuf<double> dAx, cmax((size_t)std::max<int64_t>(n, 1)), rmax((size_t)std::max<int64_t>(m, 1)),
        cs((size_t)std::max<int64_t>(n, 1)), rs((size_t)std::max<int64_t>(m, 1));
    dAp.upload(Ap, (size_t)n + 1, s);
    dAi64.upload(Ai, (size_t)nz, s);
    dAx.upload(Ax, (size_t)nz, s);
    IPXK_HIP(hipMemsetAsync(flags.get(), 0, 2 * sizeof(int), s));
    if (nz > 0) {
        hipLaunchKernelGGL(narrow_index_kernel, dim3(grid_for(nz)), dim3(kBlock), 0, s, nz, dAi64.get(), dAi.get(), (int)m, flags.get() + 1);
        hipLaunchKernelGGL(out_of_range_kernel, dim3(grid_for(nz)), dim3(kBlock), 0, s, nz, dAx.get(), flags.get());
    }
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, n, 1.0, cs.get());
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(m)), dim3(kBlock), 0, s, m, 1.0, rs.get());
    int h[2] = {0, 0};
    flags.download(h, 2, s);
    IPXK_REQUIRE(h[1] == 0, "row index out of range");
    *rounds = -1;
    if (h[0]) {                                           // :926-973
        *rounds = 0;
        for (int round = 0; round < kMaxRound; round++) {
            IPXK_HIP(hipMemsetAsync(rmax.get(), 0, sizeof(double) * (size_t)std::max<int64_t>(m, 1), s));
            IPXK_HIP(hipMemsetAsync(flags.get(), 0, sizeof(int), s));
            hipLaunchKernelGGL(col_row_max_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, n, dAp.get(), dAi.get(), dAx.get(),
                               cmax.get(), reinterpret_cast<unsigned long long*>(rmax.get()));
            hipLaunchKernelGGL(factors_kernel, dim3(grid_for(m)), dim3(kBlock), 0, s, m, rmax.get(), rs.get(), flags.get());
            hipLaunchKernelGGL(factors_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, n, cmax.get(), cs.get(), flags.get());
            int out = 0;
            flags.download(&out, 1, s);
            if (!out) break;
            hipLaunchKernelGGL(rescale_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, n, dAp.get(), dAi.get(), dAx.get(),
                               cmax.get(), rmax.get());
            (*rounds)++;
        }
        dAx.download(Ax, (size_t)nz, s);
    }
    cs.download(colscale, (size_t)n, s);
    rs.download(rowscale, (size_t)m, s);
    IPXK_HIP(hipStreamSynchronize(s));
    IPXK_HIP(hipGetLastError());
}

void transpose_device(int device, int64_t m, int64_t n, const ipxint* Ap, const ipxint* Ai, const double* Ax,
                      ipxint* Tp, ipxint* Ti, double* Tx) {
    check_csc(m, n, Ap);
    DeviceGuard dev(device);
    hipStream_t s = dev.s;
    const int64_t nz = Ap[n];
    const size_t nz1 = (size_t)std::max<int64_t>(nz, 1);
    DevBuf<ipxint> dAp, dAi64, dTp((size_t)m + 1), dTi(nz1);
    DevBuf<int> rows(nz1), rows2(nz1), pos(nz1), perm(nz1), colof(nz1), bad(1);
    DevBuf<double> dAx, dTx(nz1);
    dAp.upload(Ap, (size_t)n + 1, s);
    dAi64.upload(Ai, (size_t)nz, s);
    dAx.upload(Ax, (size_t)nz, s);
    IPXK_HIP(hipMemsetAsync(bad.get(), 0, sizeof(int), s));
    if (nz > 0) {
        hipLaunchKernelGGL(narrow_index_kernel, dim3(grid_for(nz)), dim3(kBlock), 0, s, nz, dAi64.get(), rows.get(), (int)m, bad.get());
        hipLaunchKernelGGL(entry_columns_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, n, dAp.get(), colof.get(), pos.get());
        int bits = 1;
        while ((int64_t(1) << bits) < std::max<int64_t>(m, 2)) bits++;
        size_t bytes = 0;
        IPXK_HIP(rocprim::radix_sort_pairs(nullptr, bytes, rows.get(), rows2.get(), pos.get(), perm.get(), (size_t)nz, 0u, (unsigned)bits, s));
        DevBuf<unsigned char> tmp(bytes);
        IPXK_HIP(rocprim::radix_sort_pairs(tmp.get(), bytes, rows.get(), rows2.get(), pos.get(), perm.get(), (size_t)nz, 0u, (unsigned)bits, s));
        hipLaunchKernelGGL(gather_transposed_kernel, dim3(grid_for(nz)), dim3(kBlock), 0, s, nz, perm.get(), colof.get(), dAx.get(),
                           dTi.get(), dTx.get());
        IPXK_HIP(hipStreamSynchronize(s));   // tmp goes out of scope below
    }
    hipLaunchKernelGGL(row_pointers_kernel, dim3(grid_for(m + 1)), dim3(kBlock), 0, s, m, nz, rows2.get(), dTp.get());
    int flag = 0;
    bad.download(&flag, 1, s);
    IPXK_REQUIRE(flag == 0, "row index out of range");
    dTp.download(Tp, (size_t)m + 1, s);
    dTi.download(Ti, (size_t)nz, s);
    dTx.download(Tx, (size_t)nz, s);
    IPXK_HIP(hipStreamSynchronize(s));
    IPXK_HIP(hipGetLastError());
}

}  // namespace ipxk
