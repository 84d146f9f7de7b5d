// IPM::SolveNewtonSystem on the device            reference src/ipm.cc:532-645
//
// The caller of KKTSolver::Solve in the reference: builds the right-hand side of the KKT system
// from the residuals (rb, rc, rl, ru) and the complementarity targets (sl, su), solves, and recovers
// the six components of the Newton step.  Everything outside the solve is O(n+m) elementwise work
// plus one A'dy product; doing it here keeps rhs1/rhs2/dx/dy on the device between the steps
// (SURVEY.md section 8f, row 3).  Variable states follow Iterate::StateOf / has_barrier_lb / _ub
// (src/iterate.h:99-108, 295-318): the caller passes one byte per variable.
#include "context.hpp"
#include "spmv_kernels.hpp"

namespace ipxk {

namespace {

int vec_grid(int64_t len) {
    int64_t g = (len + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return (int)(g < 1024 ? g : 1024);
}

__device__ __forceinline__ bool has_lb(unsigned char st) { return st == IPXK_STATE_BARRIER_LB || st == IPXK_STATE_BARRIER_BOXED; }
__device__ __forceinline__ bool has_ub(unsigned char st) { return st == IPXK_STATE_BARRIER_UB || st == IPXK_STATE_BARRIER_BOXED; }
__device__ __forceinline__ bool is_barrier(unsigned char st) { return st >= IPXK_STATE_BARRIER_LB; }

// :551-566
__global__ void newton_rhs_kernel(int N, const double* __restrict__ rc, const double* __restrict__ rl,
                                  const double* __restrict__ ru, const double* __restrict__ sl,
                                  const double* __restrict__ su, const double* __restrict__ xl,
                                  const double* __restrict__ xu, const double* __restrict__ zl,
                                  const double* __restrict__ zu, const unsigned char* __restrict__ state,
                                  double* __restrict__ rhs1) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        const unsigned char st = state[j];
        double r = rc ? -rc[j] : 0.0;
        const double rlj = rl ? rl[j] : 0.0, ruj = ru ? ru[j] : 0.0;
        if (has_lb(st)) r += (sl[j] + zl[j] * rlj) / xl[j];
        if (has_ub(st)) r -= (su[j] - zu[j] * ruj) / xu[j];
        if (st == IPXK_STATE_FIXED) r = 0.0;
        rhs1[j] = r;
    }
}

// :577-611 (dy *= -1 is done by the caller of this kernel on the m-vector)
__global__ void newton_recover_kernel(int N, const double* __restrict__ rl, const double* __restrict__ ru,
                                      const double* __restrict__ sl, const double* __restrict__ su,
                                      const double* __restrict__ xl, const double* __restrict__ xu,
                                      const double* __restrict__ zl, const double* __restrict__ zu,
                                      const unsigned char* __restrict__ state, const double* __restrict__ dx,
                                      double* __restrict__ dxl, double* __restrict__ dxu,
                                      double* __restrict__ dzl, double* __restrict__ dzu) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        if (!is_barrier(state[j])) {
            dxl[j] = 0.0; dzl[j] = 0.0; dxu[j] = 0.0; dzu[j] = 0.0;
            continue;
        }
        const double rlj = rl ? rl[j] : 0.0, ruj = ru ? ru[j] : 0.0;
        const double l = dx[j] - rlj;
        dxl[j] = l;
        dzl[j] = (sl[j] - zl[j] * l) / xl[j];
        const double u = ruj - dx[j];
        dxu[j] = u;
        dzu[j] = (su[j] - zu[j] * u) / xu[j];
    }
}

__global__ void negate_kernel(int len, double* __restrict__ v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) v[i] = -v[i];
}

// :617-633  shift the residual to the last two block equations, given atdy = AI[:,j]'dy
__device__ __forceinline__ void newton_shift(int j, double atdy, const double* rc, const double* xl,
                                             const double* xu, const double* zl, const double* zu,
                                             const unsigned char* state, double* dzl, double* dzu) {
    if (!is_barrier(state[j])) return;
    const double rcj = rc ? rc[j] : 0.0;
    const bool fl = isfinite(xl[j]), fu = isfinite(xu[j]);
    bool lower;
    if (fl && fu) lower = zl[j] * xu[j] >= zu[j] * xl[j];
    else lower = fl;
    if (lower) dzl[j] = rcj + dzu[j] - atdy;
    else dzu[j] = -rcj + dzl[j] + atdy;
}

struct EpiNewtonShift : ProdMul {   // structural columns: atdy = DotColumn(AI, j, dy)
    const double* rc; const double* xl; const double* xu; const double* zl; const double* zu;
    const unsigned char* state; double* dzl; double* dzu;
    static constexpr bool kNeg = false;
    __device__ __forceinline__ double init(int) const { return 0.0; }
    __device__ __forceinline__ void finish(int j, double acc, double&) const {
        newton_shift(j, acc, rc, xl, xu, zl, zu, state, dzl, dzu);
    }
};

__global__ void newton_shift_slack_kernel(int n, int m, const double* __restrict__ dy, const double* rc,
                                          const double* xl, const double* xu, const double* zl, const double* zu,
                                          const unsigned char* state, double* dzl, double* dzu) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
        newton_shift(n + i, dy[i], rc, xl, xu, zl, zu, state, dzl, dzu);
}

}  // namespace

CrResult newton_solve_dev(Context* c, bool use_basis, const double* rb, const double* rc, const double* rl,
                          const double* ru, const double* sl, const double* su, const double* xl, const double* xu,
                          const double* zl, const double* zu, const unsigned char* state, double tol,
                          ipxint maxiter, double* dx, double* dxl, double* dxu, double* dy, double* dzl,
                          double* dzu, ipxk_interrupt_fn interrupt, void* user, ipxk_times* times) {
    IPXK_REQUIRE(!comm_active(c), "ipxk_newton_solve is not available on a partitioned system");
    const int n = (int)c->n, m = (int)c->m, N = n + m;
    hipStream_t s = c->stream;
    c->nw_rhs1.resize((size_t)std::max(N, 1));
    c->nw_rhs2.resize((size_t)std::max(m, 1));
    hipLaunchKernelGGL(newton_rhs_kernel, dim3(vec_grid(N)), dim3(kBlock), 0, s, N, rc, rl, ru, sl, su, xl, xu, zl,
                       zu, state, c->nw_rhs1.get());
    if (rb) IPXK_HIP(hipMemcpyAsync(c->nw_rhs2.get(), rb, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    else IPXK_HIP(hipMemsetAsync(c->nw_rhs2.get(), 0, sizeof(double) * m, s));
    // :569-573
    CrResult res = use_basis
        ? kkt_basis_solve_dev(c, c->nw_rhs1.get(), c->nw_rhs2.get(), tol, maxiter, dx, dy, interrupt, user, times)
        : kkt_diag_solve_dev(c, c->nw_rhs1.get(), c->nw_rhs2.get(), tol, maxiter, dx, dy, interrupt, user, times);
    if (res.errflag) return res;
    // :576-611
    hipLaunchKernelGGL(negate_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, s, m, dy);
    hipLaunchKernelGGL(newton_recover_kernel, dim3(vec_grid(N)), dim3(kBlock), 0, s, N, rl, ru, sl, su, xl, xu, zl,
                       zu, state, dx, dxl, dxu, dzl, dzu);
    // :617-633
    EpiNewtonShift es{{}, rc, xl, xu, zl, zu, state, dzl, dzu};
    launch_spmv(c->Acols, dy, es, nullptr, nullptr, s);
    hipLaunchKernelGGL(newton_shift_slack_kernel, dim3(vec_grid(m)), dim3(kBlock), 0, s, n, m, dy, rc, xl, xu, zl,
                       zu, state, dzl, dzu);
    IPXK_HIP(hipGetLastError());
    return res;
}

}  // namespace ipxk
