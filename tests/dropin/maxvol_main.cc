// TEST INFRASTRUCTURE.  Maxvolume against the reference ITSELF: the reference's ipx::Maxvolume
// (src/maxvolume.cc: RunHeuristic / RunSequential) drives the reference's ipx::Basis -- Forrest-Tomlin updates, its own
// SolveForUpdate / TableauRow / ExchangeIfStable, the LU kernel being ipx::LuKernelHip (tests/dropin/basiclu_absent.cc) --
// from the slack basis of a presolved model, and ipxk_maxvolume / ipxk_maxvolume_sequential (product-form etas on the
// resident factors, MI355X) start from the same basis with the same scaling factors and parameters.  Compared: the
// final basis (as a set of variables), the number of updates and of skipped candidates, the volume gained.  The two
// keep their factorizations current in different ways, so the tableau entries agree to rounding only and a borderline
// threshold decision may flip; the program prints how far the two runs agree and fails when the final bases differ in
// more than 1 % of their columns or the volume gained differs by more than 1e-6 relative.
// usage: test_maxvol_dropin <m> <n> <entering> <seed> [sequential]; built by `make -C oracle maxvol_dropin`;
// run by tests/test_gpu_maxvolume.py.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>

#include "basis.h"
#include "control.h"
#include "maxvolume.h"
#include "model.h"
#include "presolver.h"
#include "user_model.h"
#include "ipx_kkt_hip.h"

using ipx::Int;

#define CK(call)                                                                                   \
    do {                                                                                           \
        if ((call) != 0) { std::printf("%s failed: %s\n", #call, ipxk_last_error()); return 2; }   \
    } while (0)

int main(int argc, char** argv) {
    const Int m = argc > 1 ? atol(argv[1]) : 2000;
    const Int n = argc > 2 ? atol(argv[2]) : 5000;
    const Int entering = argc > 3 ? atol(argv[3]) : 300;
    const unsigned long seed = argc > 4 ? strtoul(argv[4], nullptr, 10) : 12345;
    const bool sequential = argc > 5 && atoi(argv[5]) != 0;
    const Int k = 6;
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> uval(0.5, 4.0), u01(-1.0, 1.0);
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
    std::vector<double> obj(n, 1.0), lb(n, 0.0), ub(n, INFINITY), rhs(m, 1.0);
    std::vector<char> ct(m, '<');

    ipx::Control control;
    ipx::Parameters params;
    params.display = 0;
    params.lu_kernel = 1;
    control.parameters(params);
    ipx::UserModel user_model;
    ipx::Model model;
    if (user_model.Load(control, m, n, Ap.data(), Ai.data(), Ax.data(), rhs.data(), ct.data(), obj.data(), lb.data(), ub.data()) != 0) return 2;
    ipx::Presolver presolver(user_model, model);
    if (presolver.PresolveModel(control) != 0 || model.rows() != m || model.cols() != n) return 2;

    // scaling factors: `entering` structural columns far above the slacks (they belong into the basis), the rest below
    std::vector<double> colscale(n + m, 1.0);
    for (Int j = 0; j < n; j++) colscale[j] = std::pow(10.0, -1.0 + 0.5 * u01(rng));
    for (Int t = 0; t < entering; t++) colscale[(Int)(rng() % (uint64_t)n)] = std::pow(10.0, 2.0 + u01(rng));
    for (Int i = 0; i < m; i++) colscale[n + i] = std::pow(10.0, 0.3 * u01(rng));

    // ---- the reference
    ipx::Basis basis(control, model);                 // the slack basis (src/basis.cc:31)
    ipx::Maxvolume maxvol(control);
    const Int err_ref = sequential ? maxvol.RunSequential(colscale.data(), basis) : maxvol.RunHeuristic(colscale.data(), basis);
    std::set<Int> ref_basis;
    for (Int p = 0; p < m; p++) ref_basis.insert(basis[p]);
    std::printf("reference: errflag %ld updates %ld skipped %ld %s %ld volinc %.12g time %.3f s, %ld LU factorizations\n", (long)err_ref,
                (long)maxvol.updates(), (long)maxvol.skipped(), sequential ? "passes" : "slices",
                (long)(sequential ? maxvol.passes() : maxvol.slices()), maxvol.volinc(), maxvol.time(), (long)basis.factorizations());

    // ---- the device, from the same model (the solver's form: the structural part of AI), basis, scaling, parameters
    const ipx::SparseMatrix& AI = model.AI();
    std::vector<ipxint> dAp(AI.colptr(), AI.colptr() + n + 1), dAi(AI.rowidx(), AI.rowidx() + AI.colptr()[n]);
    std::vector<double> dAx(AI.values(), AI.values() + AI.colptr()[n]);
    ipxk_context* ctx = nullptr;
    CK(ipxk_create(m, n, dAp.data(), dAi.data(), dAx.data(), 0, &ctx));
    std::vector<ipxint> bas(m), status(n + m, -1), bas_out(m), status_out(n + m);
    for (Int p = 0; p < m; p++) { bas[p] = n + p; status[n + p] = 0; }
    ipxk_lu_info li{};
    CK(ipxk_lu_factorize_basis(ctx, bas.data(), 0.1, 0, &li));
    CK(ipxk_split_prepare_lu(ctx, status.data(), colscale.data()));
    ipxk_maxvolume_info mi{};
    if (sequential) {
        CK(ipxk_maxvolume_sequential(ctx, status.data(), colscale.data(), control.volume_tol(), control.maxpasses(), 100, bas_out.data(),
                                     status_out.data(), &mi, nullptr, 0));
    } else {
        ipxk_maxvolume_params prm{control.volume_tol(), control.maxskip_updates(), control.rows_per_slice(), 100};
        CK(ipxk_maxvolume(ctx, status.data(), colscale.data(), &prm, bas_out.data(), status_out.data(), &mi, nullptr, 0));
    }
    ipxk_destroy(ctx);
    std::set<Int> hip_basis(bas_out.begin(), bas_out.end());
    std::printf("device:    errflag %ld updates %ld skipped %ld %s %ld volinc %.12g time %.3f s, %ld refactorizations, %ld refused\n",
                (long)mi.errflag, (long)mi.updates, (long)mi.skipped, sequential ? "passes" : "slices", (long)mi.slices, mi.volinc, mi.seconds,
                (long)mi.factorizations, (long)mi.refused);
    Int differ = 0;
    for (Int j : ref_basis) differ += hip_basis.count(j) == 0;
    const double dv = std::abs(mi.volinc - maxvol.volinc()) / std::max(1.0, std::abs(maxvol.volinc()));
    const bool same = differ == 0 && mi.updates == maxvol.updates() && mi.skipped == maxvol.skipped();
    const bool ok = err_ref == 0 && mi.errflag == 0 && (Int)hip_basis.size() == m && differ * 100 <= m && dv <= 1e-6;
    std::printf("basic variables of the reference's final basis missing from the device's: %ld of %ld; volinc rel diff %.2e; %s -> %s\n",
                (long)differ, (long)m, dv, same ? "IDENTICAL decisions (final basis, updates, skipped)" : "decisions differ", ok ? "PASS" : "FAIL");
    std::printf(ok ? "DONE\n" : "FAILED\n");
    return ok ? 0 : 1;
}
