// One predictor-corrector step of the IPM on the device     reference src/ipm.cc:340-530
//   IPM::Predictor (:340-371), IPM::AddCorrector (:373-435), IPM::StepSizes (:437-516),
//   IPM::MakeStep (:518-530: Iterate::Update)
// composed from the pieces already resident: the iterate and its residuals (iterate.hip), two
// Newton solves (newton.hip) around the factorized KKT solver, step-to-boundary reductions, and the
// complementarity of the trial point.  Only scalars (step lengths, mu, four numbers at the blocking
// indices) reach the host.  SURVEY.md section 8f, row 3.
#include "context.hpp"
#include "device_utils.hpp"

namespace ipxk {

namespace {

int vec_grid(int64_t len) {
    int64_t g = (len + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return (int)(g < 1024 ? g : 1024);
}

__device__ __forceinline__ bool has_lb(unsigned char st) { return st == IPXK_STATE_BARRIER_LB || st == IPXK_STATE_BARRIER_BOXED; }
__device__ __forceinline__ bool has_ub(unsigned char st) { return st == IPXK_STATE_BARRIER_UB || st == IPXK_STATE_BARRIER_BOXED; }

// sl = -xl.*zl + shift - dxl.*dzl on variables with a lower barrier term, 0 elsewhere; same for su.
// Predictor: shift = 0, no step (d* == nullptr), ipm.cc:349-366.  Corrector: shift = sigma*mu, :410-428.
__global__ void complementarity_rhs_kernel(int N, const unsigned char* __restrict__ state,
                                           const double* __restrict__ xl, const double* __restrict__ xu,
                                           const double* __restrict__ zl, const double* __restrict__ zu,
                                           double shift, const double* dxl, const double* dxu, const double* dzl,
                                           const double* dzu, double* __restrict__ sl, double* __restrict__ su) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        const unsigned char st = state[j];
        double l = 0.0, u = 0.0;
        if (has_lb(st)) l = dxl ? -xl[j] * zl[j] + shift - dxl[j] * dzl[j] : -xl[j] * zl[j];
        if (has_ub(st)) u = dxu ? -xu[j] * zu[j] + shift - dxu[j] * dzu[j] : -xu[j] * zu[j];
        sl[j] = l;
        su[j] = u;
    }
}

// sum over the barrier terms of (x + ap*dx)*(z + ad*dz), ipm.cc:394-407 / :459-473
__global__ __launch_bounds__(kBlock) void trial_complementarity_kernel(int N, const unsigned char* __restrict__ state,
                                                                       const double* __restrict__ xl,
                                                                       const double* __restrict__ xu,
                                                                       const double* __restrict__ zl,
                                                                       const double* __restrict__ zu,
                                                                       const double* __restrict__ dxl,
                                                                       const double* __restrict__ dxu,
                                                                       const double* __restrict__ dzl,
                                                                       const double* __restrict__ dzu, double ap,
                                                                       double ad, double* out) {
    __shared__ double red[kBlock / 64 + 1];
    double sum = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < N; j += gridDim.x * kBlock) {
        const unsigned char st = state[j];
        if (has_lb(st)) sum += (xl[j] + ap * dxl[j]) * (zl[j] + ad * dzl[j]);
        if (has_ub(st)) sum += (xu[j] + ap * dxu[j]) * (zu[j] + ad * dzu[j]);
    }
    sum = block_reduce<SumOp>(sum, red);
    if (threadIdx.x == 0) out[blockIdx.x] = sum;
}

double trial_complementarity(Context* c, const double* dxl, const double* dxu, const double* dzl, const double* dzu,
                             double ap, double ad) {
    const int N = (int)(c->n + c->m);
    const int g = vec_grid(N);
    c->it_partials.resize((size_t)4 * 1024);
    hipLaunchKernelGGL(trial_complementarity_kernel, dim3(g), dim3(kBlock), 0, c->stream, N, c->it_state.get(),
                       c->it_xl.get(), c->it_xu.get(), c->it_zl.get(), c->it_zu.get(), dxl, dxu, dzl, dzu, ap, ad,
                       c->it_partials.get());
    std::vector<double> h(g);
    c->it_partials.download(h.data(), h.size(), c->stream);
    double s = 0.0;
    for (double v : h) s += v;
    return s;
}

double element(Context* c, const double* dev, ipxint j) {
    double v = 0.0;
    staged_d2h(&v, dev + j, sizeof(double), c->stream);
    return v;
}

}  // namespace

void ipm_step_dev(Context* c, bool use_basis, const double* b, const double* cc, const double* lb, const double* ub,
                  double kkt_tol, ipxint maxiter, ipxk_ipm_step_info* info, ipxk_interrupt_fn interrupt, void* user) {
    IPXK_REQUIRE(c->it_set, "no iterate on the device (ipxk_iterate_set)");
    const int n = (int)c->n, m = (int)c->m, N = n + m;
    hipStream_t s = c->stream;
    *info = ipxk_ipm_step_info{};
    for (int k = 0; k < 12; k++) c->ipm[k].resize((size_t)std::max(k == 0 || k == 9 ? m : N, 1));
    double *rb = c->ipm[0].get(), *rc = c->ipm[1].get(), *rl = c->ipm[2].get(), *ru = c->ipm[3].get();
    double *sl = c->ipm[4].get(), *su = c->ipm[5].get();
    double *dx = c->ipm[6].get(), *dxl = c->ipm[7].get(), *dxu = c->ipm[8].get(), *dy = c->ipm[9].get();
    double *dzl = c->ipm[10].get(), *dzu = c->ipm[11].get();
    const double *xl = c->it_xl.get(), *xu = c->it_xu.get(), *zl = c->it_zl.get(), *zu = c->it_zu.get();
    const unsigned char* state = c->it_state.get();

    iterate_residuals_dev(c, b, cc, lb, ub, rb, rc, rl, ru, &info->presidual, &info->dresidual);
    double comp[4], num_finite = 0.0;
    iterate_complementarity_dev(c, comp, &num_finite);
    const double mu = comp[1];
    info->mu_before = mu;
    const double tol = kkt_tol * std::sqrt(mu);           // ipm.cc:572
    const int g = vec_grid(N);

    // ---- Predictor, :340-371
    hipLaunchKernelGGL(complementarity_rhs_kernel, dim3(g), dim3(kBlock), 0, s, N, state, xl, xu, zl, zu, 0.0,
                       (const double*)nullptr, (const double*)nullptr, (const double*)nullptr,
                       (const double*)nullptr, sl, su);
    CrResult r = newton_solve_dev(c, use_basis, rb, rc, rl, ru, sl, su, xl, xu, zl, zu, state, tol, maxiter, dx, dxl,
                                  dxu, dy, dzl, dzu, interrupt, user, nullptr);
    info->kktiter_predictor = r.iter;
    info->errflag = r.errflag;
    if (r.errflag) return;

    // ---- AddCorrector, :373-435
    ipxint blk;
    double step_xl = step_to_boundary_dev(c, xl, dxl, N, 1.0, &blk);
    double step_xu = step_to_boundary_dev(c, xu, dxu, N, 1.0, &blk);
    double step_zl = step_to_boundary_dev(c, zl, dzl, N, 1.0, &blk);
    double step_zu = step_to_boundary_dev(c, zu, dzu, N, 1.0, &blk);
    double maxp = std::min(step_xl, step_xu), maxd = std::min(step_zl, step_zu);
    IPXK_REQUIRE(num_finite > 0.0, "the iterate has no barrier term");
    const double muaff = trial_complementarity(c, dxl, dxu, dzl, dzu, maxp, maxd) / num_finite;
    const double ratio = muaff / mu;
    const double sigma = ratio * ratio * ratio;
    info->sigma = sigma;
    hipLaunchKernelGGL(complementarity_rhs_kernel, dim3(g), dim3(kBlock), 0, s, N, state, xl, xu, zl, zu, sigma * mu,
                       (const double*)dxl, (const double*)dxu, (const double*)dzl, (const double*)dzu, sl, su);
    r = newton_solve_dev(c, use_basis, rb, rc, rl, ru, sl, su, xl, xu, zl, zu, state, tol, maxiter, dx, dxl, dxu, dy,
                         dzl, dzu, interrupt, user, nullptr);
    info->kktiter_corrector = r.iter;
    info->errflag = r.errflag;
    if (r.errflag) return;

    // ---- StepSizes, :437-516
    const double gammaf = 0.9, gammaa = 1.0 / (1.0 - gammaf);
    ipxint block_xl, block_xu, block_zl, block_zu;
    step_xl = step_to_boundary_dev(c, xl, dxl, N, 1.0, &block_xl);
    step_xu = step_to_boundary_dev(c, xu, dxu, N, 1.0, &block_xu);
    step_zl = step_to_boundary_dev(c, zl, dzl, N, 1.0, &block_zl);
    step_zu = step_to_boundary_dev(c, zu, dzu, N, 1.0, &block_zu);
    maxp = std::fmin(step_xl, step_xu);
    maxd = std::fmin(step_zl, step_zu);
    double mufull = trial_complementarity(c, dxl, dxu, dzl, dzu, maxp, maxd) / num_finite;
    mufull /= gammaa;
    double alphap = 1.0, alphad = 1.0;
    if (maxp < 1.0) {
        const bool lower = step_xl <= step_xu;
        const ipxint bp = lower ? block_xl : block_xu;
        const double z = element(c, lower ? zl : zu, bp), dz = element(c, lower ? dzl : dzu, bp);
        const double x = element(c, lower ? xl : xu, bp), d = element(c, lower ? dxl : dxu, bp);
        const double buffer = mufull / (z + maxd * dz);
        alphap = (x - buffer) / (-d);
        alphap = std::max(alphap, gammaf * maxp);
        alphap = std::min(alphap, 1.0);
    }
    if (maxd < 1.0) {
        const bool lower = step_zl <= step_zu;
        const ipxint bd = lower ? block_zl : block_zu;
        const double x = element(c, lower ? xl : xu, bd), d = element(c, lower ? dxl : dxu, bd);
        const double z = element(c, lower ? zl : zu, bd), dz = element(c, lower ? dzl : dzu, bd);
        const double buffer = mufull / (x + maxp * d);
        alphad = (z - buffer) / (-dz);
        alphad = std::max(alphad, gammaf * maxd);
        alphad = std::min(alphad, 1.0);
    }
    info->step_primal = std::min(alphap, 1.0 - 1e-6);
    info->step_dual = std::min(alphad, 1.0 - 1e-6);

    // ---- MakeStep, :518-530
    iterate_update_dev(c, info->step_primal, dx, dxl, dxu, info->step_dual, dy, dzl, dzu);
    iterate_complementarity_dev(c, comp);
    info->mu_after = comp[1];
    IPXK_HIP(hipGetLastError());
}

// IPM::Driver (src/ipm.cc:56-123) on the resident iterate, around the diag solver: termination test
// (Iterate::term_crit_reached with crossover_start = 0, src/iterate.cc:221-249), the divergence / bad-iteration
// test with its infeasibility classification (:71-93, for a model that was not dualized), the iteration limit,
// InterruptCheck, Factorize (kkt_solver_diag.cc:18-65 from the resident iterate), the predictor-corrector step,
// and MakeStep's bad-iteration count and best complementarity (:520-530).  What ends the loop is reported as
// status_ipm; a CR failure of the diag solver (errflag 201-205 -> IPX_STATUS_failed) is where LpSolver switches
// to the basis solver (src/lp_solver.cc:399-418) -- the caller's decision, as in the reference.
// Iterate::ScalingFactor (src/iterate.cc:183-198): 0 for fixed, inf for free variables, else 1/sqrt(zl/xl + zu/xu);
// flags any variable that is not in a barrier state
__global__ void scaling_factor_kernel(int N, const unsigned char* __restrict__ state, const double* __restrict__ xl,
                                      const double* __restrict__ xu, const double* __restrict__ zl, const double* __restrict__ zu,
                                      double* __restrict__ colscale, int* nonbarrier) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        const unsigned char st = state[j];
        double d;
        if (st == IPXK_STATE_FIXED) { d = 0.0; *nonbarrier = 1; }
        else if (st == IPXK_STATE_FREE) { d = __builtin_huge_val(); *nonbarrier = 1; }
        else d = 1.0 / sqrt(zl[j] / xl[j] + zu[j] / xu[j]);
        colscale[j] = d;
    }
}

// KKTSolverBasis::_Factorize (src/kkt_solver_basis.cc:20-63) on the device: scaling factors from the iterate,
// Maxvolume, fresh factorization, Prepare.  DropPrimal / DropDual (:36-43) are not taken (they move nearly
// degenerate variables out of the barrier problem; leaving them in is a valid, if slower, interior point step).
// The first call builds the slack basis (Basis::SetToSlackBasis; ConstructBasisFromWeights with crash_basis = 0,
// src/basis.cc:353-385, for a model without free or fixed variables).
static void basis_factorize_dev(Context* c, std::vector<ipxint>& basis, std::vector<ipxint>& status, std::vector<double>& colscale,
                                bool first, ipxk_ipm_info* info) {
    const int m = (int)c->m, n = (int)c->n, N = n + m;
    hipStream_t s = c->stream;
    DevBuf<double> d((size_t)N);
    DevBuf<int> flag(1);
    IPXK_HIP(hipMemsetAsync(flag.get(), 0, sizeof(int), s));
    hipLaunchKernelGGL(scaling_factor_kernel, dim3(vec_grid(N)), dim3(kBlock), 0, s, N, c->it_state.get(), c->it_xl.get(), c->it_xu.get(),
                       c->it_zl.get(), c->it_zu.get(), d.get(), flag.get());
    int nonbarrier = 0;
    flag.download(&nonbarrier, 1, s);
    d.download(colscale.data(), (size_t)N, s);
    IPXK_HIP(hipStreamSynchronize(s));
    if (nonbarrier) throw Error(IPXK_E_UNSUPPORTED, "the basis phase of ipxk_ipm_driver handles barrier variables only (no free or fixed ones)");
    if (first) {
        for (int j = 0; j < n; j++) status[j] = IPXK_NONBASIC;
        for (int i = 0; i < m; i++) { status[n + i] = IPXK_BASIC; basis[i] = n + i; }
        ipxk_lu_info li{};
        lu_factorize_basis(c, basis.data(), 0.1, false, &li);
        split_prepare_lu(c, status.data(), colscale.data());
    }
    ipxk_maxvolume_info mi{};
    std::vector<ipxint> nb(basis.size()), ns(status.size());
    maxvolume_dev(c, status.data(), colscale.data(), nullptr, nb.data(), ns.data(), &mi, nullptr, 0);
    if (mi.errflag) { info->errflag = mi.errflag; return; }
    info->basis_updates += mi.updates;
    basis.swap(nb);
    status.swap(ns);
    if (mi.updates == 0 && !first) split_rescale_host(c, status.data(), colscale.data());   // same basis, new scaling (:59-64)
}

void ipm_driver_dev(Context* c, const double* b, const double* cc, const double* lb, const double* ub,
                    const ipxk_ipm_params* prm, ipxk_ipm_info* info, ipxk_interrupt_fn interrupt, void* user, bool use_basis,
                    ipxint* basis_out, ipxint* status_out) {
    IPXK_REQUIRE(c->it_set, "no iterate on the device (ipxk_iterate_set)");
    std::vector<ipxint> basis, status;
    std::vector<double> colscale;
    if (use_basis) { basis.resize((size_t)c->m); status.resize((size_t)(c->n + c->m)); colscale.resize((size_t)(c->n + c->m)); }
    bool first_factorize = true;
    constexpr double kDivergeTol = 1e6;                  // src/ipm.h:55
    *info = ipxk_ipm_info{};
    const int N = (int)(c->n + c->m);
    for (int k = 0; k < 12; k++) c->ipm[k].resize((size_t)std::max(k == 0 || k == 9 ? (int)c->m : N, 1));
    double norms[2];
    model_norms_dev(c, b, cc, lb, ub, norms);
    double comp[4];
    iterate_complementarity_dev(c, comp);
    double best_complementarity = comp[0];               // :315
    ipxint num_bad_iter = 0, errflag = 0;
    while (true) {
        double obj[3];
        iterate_residuals_dev(c, b, cc, lb, ub, c->ipm[0].get(), c->ipm[1].get(), c->ipm[2].get(), c->ipm[3].get(),
                              &info->presidual, &info->dresidual);
        iterate_complementarity_dev(c, comp);
        iterate_objectives_dev(c, b, cc, lb, ub, obj);
        info->pobjective = obj[0] + obj[2];              // after postprocessing, iterate.cc:203-211
        info->dobjective = obj[1] + obj[2];
        info->complementarity = comp[0];
        info->mu = comp[1];
        const bool feasible = info->presidual <= prm->feasibility_tol * (1.0 + norms[0]) &&
                              info->dresidual <= prm->feasibility_tol * (1.0 + norms[1]);
        const double mid = 0.5 * (info->pobjective + info->dobjective), gap = info->pobjective - info->dobjective;
        const bool optimal = std::abs(gap) <= prm->optimality_tol * (1.0 + std::abs(mid));
        if (feasible && optimal) { info->status_ipm = 1; break; }                  // IPX_STATUS_optimal
        if (num_bad_iter >= 5 || comp[0] > kDivergeTol * best_complementarity) {
            if (info->dobjective > std::max(10.0 * std::abs(info->pobjective), 1.0)) info->status_ipm = 3;        // primal_infeas
            else if (info->pobjective < -std::max(10.0 * std::abs(info->dobjective), 1.0)) info->status_ipm = 4;  // dual_infeas
            else info->status_ipm = 7;                                                                            // no_progress
            break;
        }
        if (info->iter >= prm->ipm_maxiter) { info->status_ipm = 6; break; }       // iter_limit
        if (interrupt && (errflag = interrupt(user)) != 0) break;
        if (use_basis) {
            basis_factorize_dev(c, basis, status, colscale, first_factorize, info);
            first_factorize = false;
            errflag = info->errflag;
        } else {
            kkt_diag_factorize_dev(c, c->it_xl.get(), c->it_xu.get(), c->it_zl.get(), c->it_zu.get(), comp[1],
                                   prm->precond_dense_cols != 0, &errflag);
        }
        if (errflag) break;
        ipxk_ipm_step_info st;
        ipm_step_dev(c, use_basis, b, cc, lb, ub, prm->kkt_tol, use_basis ? -1 : prm->kkt_maxiter, &st, interrupt, user);
        info->kktiter += st.kktiter_predictor + st.kktiter_corrector;
        errflag = st.errflag;
        if (errflag) break;
        info->step_primal = st.step_primal;
        info->step_dual = st.step_dual;
        if (std::min(st.step_primal, st.step_dual) < 0.05) num_bad_iter++; else num_bad_iter = 0;   // :524-527
        iterate_complementarity_dev(c, comp);
        best_complementarity = std::min(best_complementarity, comp[0]);
        info->iter++;
    }
    if (errflag) {                                       // :114-121
        if (errflag == 999) { info->status_ipm = 5; info->errflag = 0; }           // IPX_ERROR_interrupt_time -> time_limit
        else { info->status_ipm = 8; info->errflag = errflag; }                    // failed
    }
    if (use_basis && !first_factorize) {
        if (basis_out) std::copy(basis.begin(), basis.end(), basis_out);
        if (status_out) std::copy(status.begin(), status.end(), status_out);
    }
}

}  // namespace ipxk
