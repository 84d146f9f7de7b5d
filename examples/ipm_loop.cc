// Example: a few interior-point iterations that never leave the GPU, through the C ABI only
// (include/ipx_kkt_hip.h; no reference code, no Python).  Builds a small random LP
//     min c'x  s.t.  A x + s = b,  x >= 0,  s >= 0
// in solver form (AI = [A I]), puts an interior iterate on the device and repeats what IPM::Driver
// does per iteration (reference src/ipm.cc:74-150): KKTSolver::Factorize for the current iterate, then
// predictor + corrector + step sizes + update (ipxk_ipm_step).  Prints the residual norms and mu.
// Then -- as LpSolver does when it switches from the initial to the main IPM (src/lp_solver.cc:375-462) -- the
// basis-preconditioned phase: ipxk_ipm_driver_basis runs IPM::Driver around the basis solver (Maxvolume, LU
// factorization and Prepare on the device every iteration) from the slack basis until the termination test holds.
//
//   g++ -std=c++14 -O2 -Iinclude examples/ipm_loop.cc -Lipx_amd/lib -lipx_kkt_hip -Wl,-rpath,$PWD/ipx_amd/lib -o ipm_loop
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "ipx_kkt_hip.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        const int rc_ = (call);                                                       \
        if (rc_ != IPXK_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ipxk_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    const ipxint m = argc > 1 ? std::atol(argv[1]) : 2000, n = argc > 2 ? std::atol(argv[2]) : 5000;
    const int iterations = argc > 3 ? std::atoi(argv[3]) : 6;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> mag(0.5, 4.0), unit(0.0, 1.0);
    // A: 8 entries per column, sorted rows
    std::vector<ipxint> Ap(n + 1, 0), Ai;
    std::vector<double> Ax;
    for (ipxint j = 0; j < n; j++) {
        std::vector<ipxint> rows;
        while ((ipxint)rows.size() < std::min<ipxint>(8, m)) {
            const ipxint r = (ipxint)(rng() % (uint64_t)m);
            bool dup = false;
            for (ipxint q : rows) dup |= q == r;
            if (!dup) rows.push_back(r);
        }
        std::sort(rows.begin(), rows.end());
        for (ipxint r : rows) { Ai.push_back(r); Ax.push_back((rng() & 1 ? 1.0 : -1.0) * mag(rng)); }
        Ap[j + 1] = (ipxint)Ai.size();
    }
    const ipxint N = n + m;
    // a feasible, bounded LP: b = A x0 + s0 with x0, s0 > 0;  c = A'y0 + z0 with z0 > 0
    std::vector<double> x0(N), y0(m), z0(N), b(m, 0.0), c(N, 0.0), lb(N, 0.0), ub(N, INFINITY);
    for (auto& v : x0) v = 0.5 + unit(rng);
    for (auto& v : z0) v = 0.5 + unit(rng);
    for (auto& v : y0) v = unit(rng) - 0.5;
    for (ipxint j = 0; j < n; j++)
        for (ipxint p = Ap[j]; p < Ap[j + 1]; p++) { b[Ai[p]] += Ax[p] * x0[j]; c[j] += Ax[p] * y0[Ai[p]]; }
    for (ipxint i = 0; i < m; i++) { b[i] += x0[n + i]; c[n + i] = y0[i]; }
    for (ipxint j = 0; j < N; j++) c[j] += z0[j];

    ipxk_context* ctx = nullptr;
    CHECK(ipxk_create(m, n, Ap.data(), Ai.data(), Ax.data(), 0, &ctx));
    // interior start away from the solution: x = xl = 1, y = 0, zl = 1 (all variables have a lower bound only)
    std::vector<double> x(N, 1.0), xl(N, 1.0), xu(N, INFINITY), y(m, 0.0), zl(N, 1.0), zu(N, 0.0);
    std::vector<unsigned char> state(N, IPXK_STATE_BARRIER_LB);
    CHECK(ipxk_iterate_set(ctx, x.data(), xl.data(), xu.data(), y.data(), zl.data(), zu.data(), state.data()));
    std::printf("%4s %10s %10s %10s %8s %8s %6s %6s\n", "iter", "presidual", "dresidual", "mu", "step_p", "step_d", "kkt1", "kkt2");
    for (int it = 0; it < iterations; it++) {
        ipxint errflag = 0;
        CHECK(ipxk_iterate_factorize_diag(ctx, 1, &errflag));
        if (errflag) { std::fprintf(stderr, "factorize errflag %ld\n", (long)errflag); break; }
        ipxk_ipm_step_info info;
        CHECK(ipxk_ipm_step(ctx, 0, b.data(), c.data(), lb.data(), ub.data(), 0.3, 2000, &info, nullptr, nullptr));
        std::printf("%4d %10.3e %10.3e %10.3e %8.4f %8.4f %6ld %6ld\n", it, info.presidual, info.dresidual, info.mu_before,
                    info.step_primal, info.step_dual, (long)info.kktiter_predictor, (long)info.kktiter_corrector);
        if (info.errflag) { std::fprintf(stderr, "KKT solve errflag %ld\n", (long)info.errflag); break; }
    }
    double comp[4];
    CHECK(ipxk_iterate_complementarity(ctx, comp));
    std::printf("final mu %.3e\n", comp[1]);
    if (argc > 4 && std::atoi(argv[4]) != 0) {           // main IPM with the basis solver
        ipxk_ipm_params prm{0.3, 1e-6, 1e-8, -1, 100, 1};
        ipxk_ipm_info info;
        CHECK(ipxk_ipm_driver_basis(ctx, b.data(), c.data(), lb.data(), ub.data(), &prm, &info, nullptr, nullptr, nullptr, nullptr));
        std::printf("main IPM: status %ld after %ld iterations, %ld CR iterations, %ld basis updates; pobjective %.10e dobjective %.10e "
                    "presidual %.2e dresidual %.2e\n", (long)info.status_ipm, (long)info.iter, (long)info.kktiter, (long)info.basis_updates,
                    info.pobjective, info.dobjective, info.presidual, info.dresidual);
    }
    ipxk_destroy(ctx);
    return 0;
}
