// TEST INFRASTRUCTURE.  The device-side IPM pieces of SURVEY 8f row 3 (ipxk_ipm_step, ipxk_ipm_driver: IPM::Predictor /
// AddCorrector / StepSizes / MakeStep / Driver restated on the resident iterate) against the reference ITSELF: the
// reference's ipx::IPM computes its starting point with its own KKTSolverDiag (IPM::ComputeStartingPoint), that point is
// copied into the device iterate, and then the reference's IPM::Driver (over the reference's KKTSolverDiag, CPU) and
// ipxk_ipm_driver (MI355X) run the same number of iterations from it.  Every KKT solve on either side stops at the
// reference's tolerance 0.3 sqrt(mu), so the iterates agree to that level, not to rounding: compared are, per
// iteration count K = 1, 2, 4, 8: mu, the primal / dual residuals and objectives after K iterations (relative 2e-3 of
// their scale; measured on the MI355X: equal in all printed digits, the CR iteration counts included, for the first
// four iterations) and, run to the end, the status and either the optimal value (1e-7) or the point at which the diag
// solver gives up (where LpSolver switches to the basis solver, src/lp_solver.cc:399-418).
// usage: test_ipm_dropin <m> <n> <seed>; built by `make -C oracle ipm_dropin`; run by tests/test_gpu_ipm_step.py.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "control.h"
#include "ipm.h"
#include "iterate.h"
#include "kkt_solver_diag.h"
#include "model.h"
#include "presolver.h"
#include "user_model.h"
#include "ipx_kkt_hip.h"

using ipx::Int;
using ipx::Vector;

#define CK(call)                                                                                   \
    do {                                                                                           \
        if ((call) != 0) { std::printf("%s failed: %s\n", #call, ipxk_last_error()); return 2; }   \
    } while (0)

static bool Close(double a, double b, double rel, double scale) { return std::abs(a - b) <= rel * (scale + std::max(std::abs(a), std::abs(b))); }

int main(int argc, char** argv) {
    const Int m = argc > 1 ? atol(argv[1]) : 2000;
    const Int n = argc > 2 ? atol(argv[2]) : 5000;
    const unsigned long seed = argc > 3 ? strtoul(argv[3], nullptr, 10) : 12345;
    const Int k = 6;
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> uval(0.5, 4.0), u(0.5, 2.0);
    std::vector<Int> Ap(n + 1), Ai;
    std::vector<double> Ax;
    for (Int j = 0; j < n; j++) {
        Ap[j] = (Int)Ai.size();
        std::vector<Int> rows;
        while ((Int)rows.size() < std::min(k, m)) {
            const Int r = (Int)(rng() % (uint64_t)m);
            if (std::find(rows.begin(), rows.end(), r) == rows.end()) rows.push_back(r);
        }
        std::sort(rows.begin(), rows.end());
        for (Int r : rows) { Ai.push_back(r); Ax.push_back((rng() & 1 ? 1.0 : -1.0) * uval(rng)); }
    }
    Ap[n] = (Int)Ai.size();
    // a feasible, bounded LP: an interior primal-dual point exists by construction (x0, s0 > 0; y0 < 0, z0 > 0)
    std::vector<double> x0(n), y0(m), obj(n), lb(n, 0.0), ub(n, INFINITY), rhs(m);
    for (auto& v : x0) v = u(rng);
    for (Int i = 0; i < m; i++) { rhs[i] = u(rng); y0[i] = -u(rng); }
    for (Int j = 0; j < n; j++) {
        double aty = 0.0;
        for (Int p = Ap[j]; p < Ap[j + 1]; p++) { rhs[Ai[p]] += Ax[p] * x0[j]; aty += Ax[p] * y0[Ai[p]]; }
        obj[j] = aty + u(rng);
    }
    std::vector<char> ct(m, '<');

    ipx::Control control;
    ipx::Parameters params;
    params.display = 0;
    control.parameters(params);
    ipx::UserModel user_model;
    ipx::Model model;
    if (user_model.Load(control, m, n, Ap.data(), Ai.data(), Ax.data(), rhs.data(), ct.data(), obj.data(), lb.data(), ub.data()) != 0) return 2;
    ipx::Presolver presolver(user_model, model);
    if (presolver.PresolveModel(control) != 0 || model.rows() != m || model.cols() != n) return 2;
    const Int N = n + m;

    // the reference's starting point
    ipx::KKTSolverDiag kkt0(control, model);
    ipx::Iterate start(model);
    ipx::Info info0{};
    ipx::IPM ipm0(control);
    ipm0.ComputeStartingPoint(&kkt0, &start, &info0);
    if (info0.errflag) { std::printf("ComputeStartingPoint errflag %ld\n", (long)info0.errflag); return 2; }
    const Vector sx = start.x(), sxl = start.xl(), sxu = start.xu(), sy = start.y(), szl = start.zl(), szu = start.zu();
    std::vector<unsigned char> state(N);
    for (Int j = 0; j < N; j++) {
        const bool l = start.has_barrier_lb(j), ub2 = start.has_barrier_ub(j);
        state[j] = l && ub2 ? IPXK_STATE_BARRIER_BOXED : l ? IPXK_STATE_BARRIER_LB : ub2 ? IPXK_STATE_BARRIER_UB
                   : start.StateOf(j) == ipx::Iterate::State::fixed ? IPXK_STATE_FIXED : IPXK_STATE_FREE;
    }

    const ipx::SparseMatrix& AI = model.AI();
    std::vector<ipxint> dAp(AI.colptr(), AI.colptr() + n + 1), dAi(AI.rowidx(), AI.rowidx() + AI.colptr()[n]);
    std::vector<double> dAx(AI.values(), AI.values() + AI.colptr()[n]);
    ipxk_context* ctx = nullptr;
    CK(ipxk_create(m, n, dAp.data(), dAi.data(), dAx.data(), 0, &ctx));
    const double *b = &model.b()[0], *c = &model.c()[0], *mlb = &model.lb()[0], *mub = &model.ub()[0];

    int failures = 0;
    const Int counts[] = {1, 2, 4, 8, -1};
    for (Int K : counts) {
        // reference: a fresh solver and iterate from the stored starting point
        ipx::KKTSolverDiag kkt(control, model);
        ipx::Iterate it(model);
        ipx::Info ri{};
        ipx::IPM ipm(control);
        ipm.LoadStartingPoint(sx, sxl, sxu, sy, szl, szu, &it, &ri);      // (also sets the IPM's record of the best complementarity gap)
        {
            const Vector &ax = it.x(), &axl = it.xl(), &axu = it.xu(), &ay = it.y(), &azl = it.zl(), &azu = it.zu();
            for (Int j = 0; j < N; j++)
                if (ax[j] != sx[j] || axl[j] != sxl[j] || axu[j] != sxu[j] || azl[j] != szl[j] || azu[j] != szu[j]) { std::printf("LoadStartingPoint changed the point\n"); return 2; }
            for (Int i = 0; i < m; i++) if (ay[i] != sy[i]) { std::printf("LoadStartingPoint changed y\n"); return 2; }
        }
        // device: the same point
        CK(ipxk_iterate_set(ctx, &sx[0], &sxl[0], &sxu[0], &sy[0], &szl[0], &szu[0], state.data()));
        // CR cap of the diag solver: src/lp_solver.cc:393 for the counted runs (the last of them may end where LpSolver
        // would switch to the basis solver: status failed, errflag cr_iter_limit); a generous one for the run to the end
        const Int cr_cap = K < 0 && m <= 1000 ? 20000 : std::min<Int>(500, 10 + m / 20);
        kkt.maxiter(cr_cap);
        ipm.maxiter(K < 0 ? 300 : K);
        ipm.Driver(&kkt, &it, &ri);
        ipxk_ipm_params prm{control.kkt_tol(), control.ipm_feasibility_tol(), control.ipm_optimality_tol(), cr_cap,
                            K < 0 ? 300 : K, 1};
        ipxk_ipm_info di{};
        CK(ipxk_ipm_driver(ctx, b, c, mlb, mub, &prm, &di, nullptr, nullptr));
        const double rp = it.pobjective_after_postproc(), rd = it.dobjective_after_postproc();
        const double oscale = 1.0 + std::abs(rp);
        bool ok;
        if (K > 0)
            ok = ri.iter == di.iter && Close(it.mu(), di.mu, 2e-3, 0.0) && Close(rp, di.pobjective, 2e-3, oscale) &&
                 Close(rd, di.dobjective, 2e-3, oscale) && Close(it.presidual(), di.presidual, 2e-2, 1e-9) &&
                 Close(it.dresidual(), di.dresidual, 2e-2, 1e-9);
        else if (ri.status_ipm == IPX_STATUS_optimal)     // the last iterations hinge on tolerance-level differences: 22 / 19 at 500 x 1200
            ok = di.status_ipm == IPX_STATUS_optimal && std::labs((long)(ri.iter - di.iter)) <= std::max<long>(3, ri.iter / 6) &&
                 Close(rp, di.pobjective, 1e-7, oscale) && Close(rd, di.dobjective, 1e-7, oscale);
        else                                              // the diag solver gave up: where LpSolver switches to the basis solver
            ok = ri.status_ipm == di.status_ipm && ri.errflag == di.errflag && std::labs((long)(ri.iter - di.iter)) <= 1 &&
                 Close(it.mu(), di.mu, 2e-3, 0.0) && Close(rp, di.pobjective, 2e-3, oscale);
        std::printf("%s iterations: reference status %ld iter %ld mu %.6e pobj %.10e dobj %.10e pres %.3e dres %.3e kktiter %ld | device status %ld iter %ld mu "
                    "%.6e pobj %.10e dobj %.10e pres %.3e dres %.3e kktiter %ld -> %s\n", K < 0 ? "all" : std::to_string(K).c_str(), (long)ri.status_ipm,
                    (long)ri.iter, it.mu(), rp, rd, it.presidual(), it.dresidual(), (long)ri.kktiter1, (long)di.status_ipm, (long)di.iter, di.mu, di.pobjective,
                    di.dobjective, di.presidual, di.dresidual, (long)di.kktiter, ok ? "PASS" : "FAIL");
        failures += !ok;
    }
    ipxk_destroy(ctx);
    std::printf(failures ? "FAILED\n" : "DONE\n");
    return failures ? 1 : 0;
}
