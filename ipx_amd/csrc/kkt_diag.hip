// KKTSolverDiag on the device                    reference src/kkt_solver_diag.cc
//   _Factorize (:18-65): W_j = 1/(zl/xl + zu/xu), free variables regularized with
//     regval = min(mu, smallest nonzero g); resscale_i = 1/sqrt(W[n+i]).
//   _Solve (:82-118): rhs = -b + AI*(W.*a); PCR from y = 0; x_j = W_j (a_j - AI[:,j]'y),
//     x[n+i] = b_i - sum_j x_j a_ij.
// The three sparse products reuse the row-gather SpMV with epilogues that reproduce the
// reference's evaluation order (rows of the row-wise copy list columns in ascending order,
// which is the order in which the reference's column loop touches a row).
#include "context.hpp"
#include "spmv_kernels.hpp"

namespace ipxk {

static int vec_grid(int64_t len) {
    int64_t g = (len + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return (int)(g < 1024 ? g : 1024);
}

// :34-43  W = 1/g (inf where g == 0); partial min over nonzero g
__global__ __launch_bounds__(kBlock) void kkt_weights_kernel(int N, const double* __restrict__ xl,
                                                             const double* __restrict__ xu,
                                                             const double* __restrict__ zl,
                                                             const double* __restrict__ zu,
                                                             double* __restrict__ W, double* partial) {
    __shared__ double red[kBlock / 64 + 1];
    double mn = MinOp::identity();
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < N; j += gridDim.x * kBlock) {
        const double g = zl[j] / xl[j] + zu[j] / xu[j];
        if (g != 0.0 && g < mn) mn = g;
        W[j] = 1.0 / g;
    }
    mn = block_reduce<MinOp>(mn, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = mn;
}

// :44-48, :55-56
__global__ __launch_bounds__(kBlock) void kkt_regularize_kernel(int n, int m, double mu, PartRef gmin,
                                                                double* __restrict__ W,
                                                                double* __restrict__ resscale) {
    __shared__ double red[kBlock / 64 + 1];
    double regval = mu;
    if (gmin.p) {
        const double g = reduce_partials<MinOp>(gmin, red);
        if (g < regval) regval = g;
    }
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n + m; j += gridDim.x * kBlock) {
        double w = W[j];
        if (isinf(w)) { w = 1.0 / regval; W[j] = w; }
        if (j >= n) resscale[j - n] = 1.0 / sqrt(w);
    }
}

// x[j] = w[j]*(a[j] - aty[j])   (kkt_solver_diag.cc:111-112, after the cross-rank sum of A'y)
__global__ void recover_x_kernel(int n, const double* __restrict__ w, const double* __restrict__ a,
                                 const double* __restrict__ aty, double* __restrict__ x) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        x[j] = w[j] * (a[j] - aty[j]);
}

__global__ void fill_kernel(int len, double value, double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x)
        out[i] = value;
}

__global__ void multiply_kernel(int len, const double* __restrict__ x, const double* __restrict__ y,
                                double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x)
        out[i] = x[i] * y[i];
}

// rhs[i] = (-b[i] + rhs[i]) + wI[i]*aI[i]     (column partition: rhs holds the summed product)
__global__ void kkt_rhs_finish_kernel(int m, const double* __restrict__ b, const double* __restrict__ wI,
                                      const double* __restrict__ aI, double* __restrict__ rhs) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
        rhs[i] = (-b[i] + rhs[i]) + wI[i] * aI[i];
}

// out[i] = b[i] - out[i]
__global__ void subtract_from_kernel(int m, const double* __restrict__ b, double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
        out[i] = b[i] - out[i];
}

void kkt_diag_factorize_dev(Context* c, const double* xl, const double* xu, const double* zl,
                            const double* zu, double mu, bool precond_dense_cols, ipxint* errflag) {
    const int n = (int)c->n, m = (int)c->m;
    hipStream_t s = c->stream;
    *errflag = 0;
    c->kkt_diag_factorized = false;
    c->W_own.resize((size_t)n + m);
    c->resscale.resize(m > 0 ? m : 1);
    if (c->partials.size() == 0) c->partials.resize((size_t)kNumPartialSlots * kPartialStride);
    const int g = vec_grid(n + m);
    if (xl) {
        hipLaunchKernelGGL(kkt_weights_kernel, dim3(g), dim3(kBlock), 0, s, n + m, xl, xu, zl, zu,
                           c->W_own.get(), c->part(kPartScratch));
        PartRef gmin{c->part(kPartScratch), g, 1};
        if (comm_rows(c)) {
            // smallest nonzero g over all ranks (the slack part of G is partitioned)
            gmin = publish_scalar(c, kPartScratch, g, 2);
        } else if (comm_cols(c)) {
            // ... the structural part is partitioned
            gmin = allreduce_scalar(c, kPartScratch, g, 2);
        }
        hipLaunchKernelGGL(kkt_regularize_kernel, dim3(g), dim3(kBlock), 0, s, n, m, mu, gmin,
                           c->W_own.get(), c->resscale.get());
    } else {
        // :50-52 Factorize(nullptr): G = identity
        hipLaunchKernelGGL(fill_kernel, dim3(g), dim3(kBlock), 0, s, n + m, 1.0, c->W_own.get());
        hipLaunchKernelGGL(kkt_regularize_kernel, dim3(g), dim3(kBlock), 0, s, n, m, 1.0,
                           PartRef{nullptr, 0, 1}, c->W_own.get(), c->resscale.get());
    }
    // :59-62
    c->W = c->W_own.get();
    c->normal_prepared = true;
    diag_factorize_dev(c, c->W, precond_dense_cols, errflag);
    if (*errflag) return;
    if (c->reord.active && c->kdense == 0) {
        // the CR loop will run in the renumbered model: weights, preconditioner and residual scaling in its numbering
        reorder_permute_weights(c, c->W, c->reord.W.get());
        reorder_permute_rows(c, c->diagonal.get(), c->reord.diagonal.get());
        reorder_permute_rows(c, c->resscale.get(), c->reord.resscale.get());
    }
    c->kkt_diag_factorized = true;
}

CrResult kkt_diag_solve_dev(Context* c, const double* a, const double* b, double tol,
                            ipxint maxiter, double* x, double* y, ipxk_interrupt_fn interrupt,
                            void* user, ipxk_times* times) {
    IPXK_REQUIRE(c->kkt_diag_factorized, "KKTSolverDiag not factorized");
    const int n = (int)c->n, m = (int)c->m;
    hipStream_t s = c->stream;
    const double* W = c->W;
    c->k_tmp.resize((size_t)n + m);
    c->v_rhs.resize(m > 0 ? m : 1);

    // :90-92  rhs = -b + A*(Ws.*as) + W_I.*a_I
    hipLaunchKernelGGL(multiply_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, s, n, W, a, c->k_tmp.get());
    if (comm_cols(c)) {
        EpiScale ep{{}, nullptr, c->v_rhs.get()};
        launch_spmv(c->Arows, c->k_tmp.get(), ep, nullptr, nullptr, s);
        comm_allreduce_sum(c, c->v_rhs.get(), (size_t)m);
        hipLaunchKernelGGL(kkt_rhs_finish_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, s, m, b, W + n, a + n,
                           c->v_rhs.get());
    } else {
        EpiKktRhs er{{}, b, W + n, a + n, c->v_rhs.get()};
        launch_spmv(c->Arows, c->k_tmp.get(), er, nullptr, nullptr, s);   // rows are local: no exchange
    }

    // :95-105
    CrResult res;
    if (c->reord.active && c->kdense == 0 && !comm_active(c)) {
        // rows renumbered for locality: the right-hand side goes over, the loop runs on the renumbered copy, y comes back --
        // pure permutations on either side of the same arithmetic (row sums in the renumbered order of a row's entries)
        Reordered& R = c->reord;
        reorder_permute_rows(c, c->v_rhs.get(), R.rhs.get());
        IPXK_HIP(hipMemsetAsync(R.y.get(), 0, sizeof(double) * m, s));
        struct InUse { Reordered& R; explicit InUse(Reordered& r) : R(r) { R.in_use = true; } ~InUse() { R.in_use = false; } } in_use(R);
        res = pcr_solve_dev(c, R.rhs.get(), tol, R.resscale.get(), maxiter, R.y.get(), true, interrupt, user, nullptr, 0, times);
        reorder_unpermute_rows(c, R.y.get(), y);
    } else {
        IPXK_HIP(hipMemsetAsync(y, 0, sizeof(double) * m, s));
        res = pcr_solve_dev(c, c->v_rhs.get(), tol, c->resscale.get(), maxiter, y, true, interrupt, user, nullptr, 0, times);
    }

    // :108-117
    if (comm_rows(c)) {
        // A'y = sum over ranks of A_g' y_g
        EpiScale ea{{}, nullptr, c->tcols.get()};
        launch_spmv(c->Acols, y, ea, nullptr, nullptr, s);
        comm_allreduce_sum(c, c->tcols.get(), (size_t)n);
        hipLaunchKernelGGL(recover_x_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, s, n, W, a, c->tcols.get(), x);
    } else {
        EpiRecoverX ex{{}, W, a, x};
        launch_spmv(c->Acols, y, ex, nullptr, nullptr, s);
    }
    if (comm_cols(c)) {
        // slack part: b - sum over ranks of A_g x_g
        EpiScale ep{{}, nullptr, x + n};
        launch_spmv(c->Arows, x, ep, nullptr, nullptr, s);
        comm_allreduce_sum(c, x + n, (size_t)m);
        hipLaunchKernelGGL(subtract_from_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, s, m, b, x + n);
    } else {
        EpiResidualRows es{{}, b, x + n};
        launch_spmv(c->Arows, x, es, nullptr, nullptr, s);
    }
    IPXK_HIP(hipGetLastError());
    return res;
}

}  // namespace ipxk
