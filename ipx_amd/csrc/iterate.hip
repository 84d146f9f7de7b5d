// The IPM iterate on the device                       reference src/iterate.cc, src/ipm.cc
//   Iterate::Update                 iterate.cc:94-139   (steps truncated at kBarrierMin)
//   Iterate::ComputeResiduals       iterate.cc:536-588  rb = b - AI x, rc = c - AI'y - zl + zu, rl, ru
//   Iterate::ComputeComplementarity iterate.cc:642-670
//   StepToBoundary                  ipm.cc:320-339
// with the iterate resident, the vectors of an IPM iteration never
// cross PCIe.  The two sparse products reuse the gather SpMV with epilogues that reproduce the
// reference's evaluation order ((c - zl) + zu) - A_j'y and (b - sum_j a_ij x_j) - x_{n+i}.
#include "context.hpp"
#include "spmv_kernels.hpp"

namespace ipxk {

namespace {

constexpr double kBarrierMin = 1e-30;   // Iterate::kBarrierMin, src/iterate.h:204

int vec_grid(int64_t len) {
    int64_t g = (len + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return (int)(g < 1024 ? g : 1024);
}

__device__ __forceinline__ bool has_lb(unsigned char st) { return st == IPXK_STATE_BARRIER_LB || st == IPXK_STATE_BARRIER_BOXED; }
__device__ __forceinline__ bool has_ub(unsigned char st) { return st == IPXK_STATE_BARRIER_UB || st == IPXK_STATE_BARRIER_BOXED; }

__global__ void iterate_update_kernel(int N, int m, const unsigned char* __restrict__ state, double* __restrict__ x,
                                      double* __restrict__ xl, double* __restrict__ xu, double* __restrict__ y,
                                      double* __restrict__ zl, double* __restrict__ zu, double sp,
                                      const double* dx, const double* dxl, const double* dxu, double sd,
                                      const double* dy, const double* dzl, const double* dzu) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        const unsigned char st = state[j];
        if (dx && st != IPXK_STATE_FIXED) x[j] += sp * dx[j];
        if (has_lb(st)) {
            if (dxl) xl[j] = fmax(xl[j] + sp * dxl[j], kBarrierMin);
            if (dzl) zl[j] = fmax(zl[j] + sd * dzl[j], kBarrierMin);
        }
        if (has_ub(st)) {
            if (dxu) xu[j] = fmax(xu[j] + sp * dxu[j], kBarrierMin);
            if (dzu) zu[j] = fmax(zu[j] + sd * dzu[j], kBarrierMin);
        }
        if (dy && j < m) y[j] += sd * dy[j];
    }
}

// rb[i] = (b[i] - sum_j a_ij x_j) - x[n+i]
struct EpiIterRb : ProdMul {
    const double* b; const double* xI; double* out;
    static constexpr bool kNeg = true;
    __device__ __forceinline__ double init(int r) const { return b[r]; }
    __device__ __forceinline__ void finish(int r, double acc, double&) const { out[r] = acc - xI[r]; }
};

// rc[j] = ((c[j] - zl[j]) + zu[j]) - A_j'y, 0 on fixed variables
struct EpiIterRc : ProdMul {
    const double* c; const double* zl; const double* zu; const unsigned char* state; double* out;
    static constexpr bool kNeg = false;
    __device__ __forceinline__ double init(int) const { return 0.0; }
    __device__ __forceinline__ void finish(int j, double acc, double&) const {
        out[j] = state[j] == IPXK_STATE_FIXED ? 0.0 : ((c[j] - zl[j]) + zu[j]) - acc;
    }
};

// slack part of rc, and rl, ru for all variables
__global__ void iterate_bound_residuals_kernel(int n, int m, const unsigned char* __restrict__ state,
                                               const double* __restrict__ c, const double* __restrict__ lb,
                                               const double* __restrict__ ub, const double* __restrict__ x,
                                               const double* __restrict__ xl, const double* __restrict__ xu,
                                               const double* __restrict__ y, const double* __restrict__ zl,
                                               const double* __restrict__ zu, double* __restrict__ rc,
                                               double* __restrict__ rl, double* __restrict__ ru) {
    const int N = n + m;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        const unsigned char st = state[j];
        if (j >= n) rc[j] = st == IPXK_STATE_FIXED ? 0.0 : ((c[j] - zl[j]) + zu[j]) - y[j - n];
        rl[j] = has_lb(st) ? lb[j] - x[j] + xl[j] : 0.0;
        ru[j] = has_ub(st) ? ub[j] - x[j] - xu[j] : 0.0;
    }
}

// out[0*G + b] = max |rb|, |rl|, |ru| over the block's share; out[1*G + b] = max |rc|
__global__ __launch_bounds__(kBlock) void iterate_norms_kernel(int N, int m, const double* __restrict__ rb,
                                                               const double* __restrict__ rc,
                                                               const double* __restrict__ rl,
                                                               const double* __restrict__ ru, double* out) {
    __shared__ double red[kBlock / 64 + 1];
    double p = 0.0, d = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < N; j += gridDim.x * kBlock) {
        if (j < m) p = fmax(p, fabs(rb[j]));
        p = fmax(p, fmax(fabs(rl[j]), fabs(ru[j])));
        d = fmax(d, fabs(rc[j]));
    }
    p = block_reduce<MaxOp>(p, red);
    d = block_reduce<MaxOp>(d, red);
    if (threadIdx.x == 0) { out[blockIdx.x] = p; out[gridDim.x + blockIdx.x] = d; }
}

// per block: sum, min, max of the complementarity products and their count
__global__ __launch_bounds__(kBlock) void iterate_complementarity_kernel(int N, const unsigned char* __restrict__ state,
                                                                         const double* __restrict__ xl,
                                                                         const double* __restrict__ xu,
                                                                         const double* __restrict__ zl,
                                                                         const double* __restrict__ zu, double* out) {
    __shared__ double red[kBlock / 64 + 1];
    double sum = 0.0, mn = MinOp::identity(), mx = 0.0, cnt = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < N; j += gridDim.x * kBlock) {
        const unsigned char st = state[j];
        if (has_lb(st)) { const double p = xl[j] * zl[j]; sum += p; mn = fmin(mn, p); mx = fmax(mx, p); cnt += 1.0; }
        if (has_ub(st)) { const double p = xu[j] * zu[j]; sum += p; mn = fmin(mn, p); mx = fmax(mx, p); cnt += 1.0; }
    }
    sum = block_reduce<SumOp>(sum, red);
    mn = block_reduce<MinOp>(mn, red);
    mx = block_reduce<MaxOp>(mx, red);
    cnt = block_reduce<SumOp>(cnt, red);
    if (threadIdx.x == 0) {
        const int G = gridDim.x, b = blockIdx.x;
        out[b] = sum; out[G + b] = mn; out[2 * G + b] = mx; out[3 * G + b] = cnt;
    }
}

// per block: smallest candidate step and the smallest index attaining it
__global__ __launch_bounds__(kBlock) void step_to_boundary_kernel(int len, const double* __restrict__ x,
                                                                  const double* __restrict__ dx, double alpha0,
                                                                  double* out_alpha, double* out_index) {
    __shared__ double red[kBlock / 64 + 1];
    const double damp = 1.0 - 2.220446049250313e-16;
    double best = MinOp::identity();
    double bidx = 9.0e15;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < len; i += gridDim.x * kBlock) {
        if (x[i] + alpha0 * dx[i] < 0.0) {
            const double cand = -(x[i] * damp) / dx[i];
            if (cand < best) { best = cand; bidx = (double)i; }   // i ascends per thread: first index kept
        }
    }
    const double blockbest = block_reduce<MinOp>(best, red);
    const double myidx = best == blockbest ? bidx : 9.0e15;
    const double blockidx = block_reduce<MinOp>(myidx, red);
    if (threadIdx.x == 0) { out_alpha[blockIdx.x] = blockbest; out_index[blockIdx.x] = blockidx; }
}

// per-block partial sums for Iterate::ComputeObjectives (iterate.cc:590-640, the branch of an iterate that has
// not been postprocessed; fixed / free / barrier variables): [0] sum c_j x_j over non-fixed j, [1] over fixed j
// (offset_), [2] b'y + sum lb_j zl_j - sum ub_j zu_j - sum over fixed SLACK variables of x_j y_i
__global__ __launch_bounds__(kBlock) void iterate_objectives_kernel(int n, int m, const unsigned char* __restrict__ state,
                                                                    const double* __restrict__ b, const double* __restrict__ c,
                                                                    const double* __restrict__ lb, const double* __restrict__ ub,
                                                                    const double* __restrict__ x, const double* __restrict__ y,
                                                                    const double* __restrict__ zl, const double* __restrict__ zu,
                                                                    double* out) {
    __shared__ double red[kBlock / 64 + 1];
    const int N = n + m;
    double pobj = 0.0, offset = 0.0, dobj = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < N; j += gridDim.x * kBlock) {
        const unsigned char st = state[j];
        if (st != IPXK_STATE_FIXED) pobj += c[j] * x[j]; else offset += c[j] * x[j];
        if (has_lb(st)) dobj += lb[j] * zl[j];
        if (has_ub(st)) dobj -= ub[j] * zu[j];
        if (j < m) dobj += b[j] * y[j];
        if (st == IPXK_STATE_FIXED && j >= n) dobj -= x[j] * y[j - n];
    }
    pobj = block_reduce<SumOp>(pobj, red);
    offset = block_reduce<SumOp>(offset, red);
    dobj = block_reduce<SumOp>(dobj, red);
    if (threadIdx.x == 0) { out[blockIdx.x] = pobj; out[gridDim.x + blockIdx.x] = offset; out[2 * gridDim.x + blockIdx.x] = dobj; }
}
// fixed structural variables: dot partial += x_j * (A_j'y)
struct EpiObjFixed : ProdMul {
    const unsigned char* state; const double* x;
    static constexpr bool kNeg = false;
    __device__ __forceinline__ double init(int) const { return 0.0; }
    __device__ __forceinline__ void finish(int j, double acc, double& dot) const {
        if (state[j] == IPXK_STATE_FIXED) dot += x[j] * acc;
    }
};
// Model::ComputeNorms (model.cc:58-67): per-block maxima of |b|, finite |lb|, |ub| and of |c|
__global__ __launch_bounds__(kBlock) void model_norms_kernel(int n, int m, const double* __restrict__ b,
                                                             const double* __restrict__ c, const double* __restrict__ lb,
                                                             const double* __restrict__ ub, double* out) {
    __shared__ double red[kBlock / 64 + 1];
    double nb = 0.0, nc = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n + m; j += gridDim.x * kBlock) {
        if (j < m) nb = fmax(nb, fabs(b[j]));
        if (isfinite(lb[j])) nb = fmax(nb, fabs(lb[j]));
        if (isfinite(ub[j])) nb = fmax(nb, fabs(ub[j]));
        nc = fmax(nc, fabs(c[j]));
    }
    nb = block_reduce<MaxOp>(nb, red);
    nc = block_reduce<MaxOp>(nc, red);
    if (threadIdx.x == 0) { out[blockIdx.x] = nb; out[gridDim.x + blockIdx.x] = nc; }
}

}  // namespace

// out3 = pobjective, dobjective, offset
void iterate_objectives_dev(Context* c, const double* b, const double* cc, const double* lb, const double* ub,
                            double out3[3]) {
    IPXK_REQUIRE(c->it_set, "no iterate on the device (ipxk_iterate_set)");
    IPXK_REQUIRE(!comm_active(c), "the device iterate is not available on a partitioned system");
    const int n = (int)c->n, m = (int)c->m, N = n + m;
    hipStream_t s = c->stream;
    const int g = vec_grid(N);
    c->it_partials.resize((size_t)4 * 1024);
    if (c->partials.size() == 0) c->partials.resize((size_t)kNumPartialSlots * kPartialStride);
    hipLaunchKernelGGL(iterate_objectives_kernel, dim3(g), dim3(kBlock), 0, s, n, m, c->it_state.get(), b, cc, lb, ub,
                       c->it_x.get(), c->it_y.get(), c->it_zl.get(), c->it_zu.get(), c->it_partials.get());
    EpiObjFixed ef{{}, c->it_state.get(), c->it_x.get()};
    const int np = launch_spmv(c->Acols, c->it_y.get(), ef, c->part(kPartScratch), nullptr, s);
    std::vector<double> h((size_t)3 * g), hf((size_t)std::max(np, 1));
    c->it_partials.download(h.data(), h.size(), s);
    if (np > 0) staged_d2h(hf.data(), c->part(kPartScratch), sizeof(double) * (size_t)np, s);
    IPXK_HIP(hipGetLastError());
    double pobj = 0.0, offset = 0.0, dobj = 0.0, fixed = 0.0;
    for (int i = 0; i < g; i++) { pobj += h[i]; offset += h[(size_t)g + i]; dobj += h[(size_t)2 * g + i]; }
    for (int i = 0; i < np; i++) fixed += hf[i];
    out3[0] = pobj; out3[1] = dobj - fixed; out3[2] = offset;
}

// out2 = norm_bounds, norm_c
void model_norms_dev(Context* c, const double* b, const double* cc, const double* lb, const double* ub, double out2[2]) {
    const int n = (int)c->n, m = (int)c->m, N = n + m;
    const int g = vec_grid(N);
    c->it_partials.resize((size_t)4 * 1024);
    hipLaunchKernelGGL(model_norms_kernel, dim3(g), dim3(kBlock), 0, c->stream, n, m, b, cc, lb, ub, c->it_partials.get());
    std::vector<double> h((size_t)2 * g);
    c->it_partials.download(h.data(), h.size(), c->stream);
    IPXK_HIP(hipGetLastError());
    out2[0] = out2[1] = 0.0;
    for (int i = 0; i < g; i++) { out2[0] = std::max(out2[0], h[i]); out2[1] = std::max(out2[1], h[(size_t)g + i]); }
}

void iterate_update_dev(Context* c, double sp, const double* dx, const double* dxl, const double* dxu, double sd,
                        const double* dy, const double* dzl, const double* dzu) {
    IPXK_REQUIRE(c->it_set, "no iterate on the device (ipxk_iterate_set)");
    const int n = (int)c->n, m = (int)c->m, N = n + m;
    hipLaunchKernelGGL(iterate_update_kernel, dim3(vec_grid(N)), dim3(kBlock), 0, c->stream, N, m, c->it_state.get(),
                       c->it_x.get(), c->it_xl.get(), c->it_xu.get(), c->it_y.get(), c->it_zl.get(), c->it_zu.get(),
                       sp, dx, dxl, dxu, sd, dy, dzl, dzu);
    IPXK_HIP(hipGetLastError());
}

void iterate_residuals_dev(Context* c, const double* b, const double* cc, const double* lb, const double* ub,
                           double* rb, double* rc, double* rl, double* ru, double* presidual, double* dresidual) {
    IPXK_REQUIRE(c->it_set, "no iterate on the device (ipxk_iterate_set)");
    IPXK_REQUIRE(!comm_active(c), "the device iterate is not available on a partitioned system");
    const int n = (int)c->n, m = (int)c->m, N = n + m;
    hipStream_t s = c->stream;
    EpiIterRb eb{{}, b, c->it_x.get() + n, rb};
    launch_spmv(c->Arows, c->it_x.get(), eb, nullptr, nullptr, s);
    EpiIterRc ec{{}, cc, c->it_zl.get(), c->it_zu.get(), c->it_state.get(), rc};
    launch_spmv(c->Acols, c->it_y.get(), ec, nullptr, nullptr, s);
    hipLaunchKernelGGL(iterate_bound_residuals_kernel, dim3(vec_grid(N)), dim3(kBlock), 0, s, n, m, c->it_state.get(),
                       cc, lb, ub, c->it_x.get(), c->it_xl.get(), c->it_xu.get(), c->it_y.get(), c->it_zl.get(),
                       c->it_zu.get(), rc, rl, ru);
    const int g = vec_grid(N);
    c->it_partials.resize((size_t)4 * 1024);
    hipLaunchKernelGGL(iterate_norms_kernel, dim3(g), dim3(kBlock), 0, s, N, m, rb, rc, rl, ru, c->it_partials.get());
    std::vector<double> h((size_t)2 * g);
    c->it_partials.download(h.data(), h.size(), s);
    IPXK_HIP(hipGetLastError());
    double p = 0.0, d = 0.0;
    for (int i = 0; i < g; i++) { p = std::max(p, h[i]); d = std::max(d, h[(size_t)g + i]); }
    if (presidual) *presidual = p;
    if (dresidual) *dresidual = d;
}

void iterate_complementarity_dev(Context* c, double out4[4], double* num_terms) {
    IPXK_REQUIRE(c->it_set, "no iterate on the device (ipxk_iterate_set)");
    const int N = (int)(c->n + c->m);
    const int g = vec_grid(N);
    c->it_partials.resize((size_t)4 * 1024);
    hipLaunchKernelGGL(iterate_complementarity_kernel, dim3(g), dim3(kBlock), 0, c->stream, N, c->it_state.get(),
                       c->it_xl.get(), c->it_xu.get(), c->it_zl.get(), c->it_zu.get(), c->it_partials.get());
    std::vector<double> h((size_t)4 * g);
    c->it_partials.download(h.data(), h.size(), c->stream);
    IPXK_HIP(hipGetLastError());
    double sum = 0.0, mn = INFINITY, mx = 0.0, cnt = 0.0;
    for (int i = 0; i < g; i++) {
        sum += h[i]; mn = std::min(mn, h[(size_t)g + i]); mx = std::max(mx, h[(size_t)2 * g + i]); cnt += h[(size_t)3 * g + i];
    }
    // :666-669
    double mu = 0.0;
    if (cnt > 0) mu = sum / cnt; else mn = 0.0;
    out4[0] = sum; out4[1] = mu; out4[2] = mn; out4[3] = mx;
    if (num_terms) *num_terms = cnt;
}

double step_to_boundary_dev(Context* c, const double* x, const double* dx, int64_t len, double alpha0,
                            ipxint* blocking) {
    IPXK_REQUIRE(len >= 0 && len < (int64_t(1) << 31), "bad length");
    const int g = vec_grid(len);
    c->it_partials.resize((size_t)4 * 1024);
    hipLaunchKernelGGL(step_to_boundary_kernel, dim3(g), dim3(kBlock), 0, c->stream, (int)len, x, dx, alpha0,
                       c->it_partials.get(), c->it_partials.get() + g);
    std::vector<double> h((size_t)2 * g);
    c->it_partials.download(h.data(), h.size(), c->stream);
    IPXK_HIP(hipGetLastError());
    double alpha = alpha0;
    ipxint blk = -1;
    for (int i = 0; i < g; i++)
        if (h[i] < alpha) { alpha = h[i]; blk = (ipxint)h[(size_t)g + i]; }   // blocks ascend: first index kept
    if (blocking) *blocking = blk;
    return alpha;
}

}  // namespace ipxk
