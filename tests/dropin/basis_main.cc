// TEST INFRASTRUCTURE.  The basis path at the reference's own boundary, object against object on ONE ipx::Basis:
//   * ipx::SplittedNormalMatrix (reference, CPU) and ipx::SplittedNormalMatrixHip (this repo, MI355X), both prepared
//     from the same Basis and the same scaling factors: single applications, and the reference's own plain
//     ConjugateResiduals::Solve (src/conjugate_residuals.cc:14-88) running on the host over either operator;
//   * ipx::KKTSolverBasis::Solve (src/kkt_solver_basis.cc:75-194) and ipx::KKTSolverBasisHip::Solve on the same
//     right-hand sides after both have been factorized for the same basis and iterate.
// This pins the two rows of the oracle that no reference run pinned before -- SplittedNormalMatrix::Prepare (a8) and
// KKTSolverBasis::_Solve (a14) -- against the reference itself.  The Basis is the reference's own (StartingBasis,
// Maxvolume, Forrest-Tomlin updates); its LU kernel is ipx::LuKernelHip (tests/dropin/basiclu_absent.cc: BASICLU is not
// in the image), i.e. the factors are INPUTS that both sides share, as SURVEY 8c defines parity for these rows.
// usage: test_basis_dropin <m> <n> [seed]; built by `make -C oracle basis_dropin`; run by tests/test_gpu_lp_dropin.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "basis.h"
#include "conjugate_residuals.h"
#include "control.h"
#include "iterate.h"
#include "kkt_solver_basis.h"
#include "kkt_solver_basis_hip.h"
#include "linear_operators_hip.h"
#include "model.h"
#include "presolver.h"
#include "splitted_normal_matrix.h"
#include "starting_basis.h"
#include "user_model.h"

using ipx::Int;
using ipx::Vector;

static double RelErr(const Vector& a, const Vector& b) {
    double num = 0.0, den = 0.0;
    for (size_t i = 0; i < a.size(); i++) { num = std::max(num, std::abs(a[i] - b[i])); den = std::max(den, std::abs(b[i])); }
    return den > 0.0 ? num / den : num;
}

int main(int argc, char** argv) {
    const Int m = argc > 1 ? atol(argv[1]) : 1500;
    const Int n = argc > 2 ? atol(argv[2]) : 3500;
    const unsigned long seed = argc > 3 ? strtoul(argv[3], nullptr, 10) : 12345;
    const Int k = 6;
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> uval(0.5, 4.0), u01(-1.0, 1.0), uab(-0.5, 0.5);
    std::vector<Int> Ap(n + 1), Ai;
    std::vector<double> Ax;
    for (Int j = 0; j < n; j++) {
        Ap[j] = (Int)Ai.size();
        std::vector<Int> rows;
        while ((Int)rows.size() < std::min(k, m)) {
            const Int r = (Int)(rng() % (uint64_t)m);
            bool dup = false;
            for (Int q : rows) dup |= q == r;
            if (!dup) rows.push_back(r);
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

    // an interior iterate whose scaling factors spread over four decades
    Vector x0(1.0, n + m), y0(0.0, m), xl(n + m), xu(INFINITY, n + m), zl(n + m), zu(0.0, n + m);
    for (Int j = 0; j < n + m; j++) { xl[j] = std::pow(10.0, u01(rng)); zl[j] = std::pow(10.0, u01(rng)); }
    ipx::Iterate iterate(model);
    iterate.Initialize(x0, xl, xu, y0, zl, zu);

    ipx::Info info;
    ipx::Basis basis(control, model);
    ipx::StartingBasis(&iterate, &basis, &info);
    if (info.errflag) { std::printf("StartingBasis errflag %ld\n", (long)info.errflag); return 2; }
    ipx::KKTSolverBasis ref(control, basis);
    ipx::KKTSolverBasisHip hip(control, basis);
    // both solvers factorize for the same basis: Factorize() may still update the basis (Maxvolume, drops), so they
    // take turns until a call of the reference's solver changes nothing after the Hip solver's
    int rounds = 0;
    for (;; rounds++) {
        ipx::Info i1, i2;
        hip.Factorize(&iterate, &i1);
        if (!i1.errflag) i1.errflag = hip.FlushBasis();     // (the reference's solver reads the shared Basis next)
        ref.Factorize(&iterate, &i2);
        if (i1.errflag || i2.errflag) { std::printf("Factorize errflag %ld %ld\n", (long)i1.errflag, (long)i2.errflag); return 2; }
        if (ref.basis_changes() == 0) break;
        if (rounds > 20) { std::printf("the basis does not settle\n"); return 2; }
    }
    Int nstruct_basic = 0;
    for (Int p = 0; p < m; p++) nstruct_basic += basis[p] < n;
    std::printf("basis settled after %d extra rounds: %ld structural columns basic, %ld LU factorizations\n", rounds, (long)nstruct_basic,
                (long)basis.factorizations());

    int failures = 0;
    // ---- operator level: SplittedNormalMatrix against SplittedNormalMatrixHip on this basis
    {
        Vector colscale(n + m);
        for (Int j = 0; j < n + m; j++) colscale[j] = iterate.ScalingFactor(j);
        ipx::SplittedNormalMatrix C_ref(model);
        C_ref.Prepare(basis, &colscale[0]);
        ipx::HipModel device(model);
        ipx::SplittedNormalMatrixHip C_hip(device);
        C_hip.Prepare(basis, &colscale[0]);
        Vector v(m), l1(m), l2(m);
        for (auto& t : v) t = uab(rng);
        double d1 = 0, d2 = 0;
        ipx::LinearOperator &Cr = C_ref, &Ch = C_hip;
        Cr.Apply(v, l1, &d1);
        Ch.Apply(v, l2, &d2);
        const double ea = RelErr(l2, l1);
        ipx::ConjugateResiduals cr1(control), cr2(control);
        Vector y1(0.0, m), y2(0.0, m);
        cr1.Solve(Cr, v, 1e-9, nullptr, -1, y1);
        cr2.Solve(Ch, v, 1e-9, nullptr, -1, y2);
        const bool ok = ea <= 1e-10 && std::abs(d1 - d2) <= 1e-10 * std::abs(d1) && cr1.errflag() == cr2.errflag() &&
                        std::labs((long)(cr1.iter() - cr2.iter())) <= 3 + cr1.iter() / 10 && RelErr(y2, y1) < 1e-6;
        std::printf("SplittedNormalMatrix: apply relerr %.2e dot relerr %.2e | reference CR over the reference operator %ld its errflag %ld, "
                    "over the Hip operator %ld its errflag %ld, y relerr %.2e -> %s\n", ea, std::abs(d1 - d2) / std::abs(d1),
                    (long)cr1.iter(), (long)cr1.errflag(), (long)cr2.iter(), (long)cr2.errflag(), RelErr(y2, y1), ok ? "PASS" : "FAIL");
        failures += !ok;
    }
    // ---- solver level: KKTSolverBasis::Solve against KKTSolverBasisHip::Solve
    for (int pass = 0; pass < 2; pass++) {
        Vector a(n + m), b(m), xr(n + m), yr(m), xh(n + m), yh(m);
        for (auto& t : a) t = uab(rng);
        for (auto& t : b) t = uab(rng);
        const double tol = pass == 0 ? 1e-9 : 0.3 * std::sqrt(iterate.mu());
        ipx::Info i1, i2;
        ref.Solve(a, b, tol, xr, yr, &i1);
        hip.Solve(a, b, tol, xh, yh, &i2);
        const double ex = RelErr(xh, xr), ey = RelErr(yh, yr);
        // at the IPM's tolerance 0.3 sqrt(mu) (0.6 for this iterate) two correct solvers agree to about tol relative to the
        // solution, not better, and may stop a few iterations apart: only the counts are compared there
        // (iteration counts to 10 %: the last iterations of a run to 1e-9 hinge on the rounding of the operator -- with the
        // dense block of the factors applied as an explicit inverse 337 against 355, the solutions agreeing to 2e-10)
        const bool ok = pass == 0 ? (i1.errflag == i2.errflag && std::labs((long)(i1.kktiter2 - i2.kktiter2)) <= 3 + i1.kktiter2 / 10 &&
                                     ex < 1e-6 && ey < 1e-6)
                                  : (i1.errflag == i2.errflag && std::labs((long)(i1.kktiter2 - i2.kktiter2)) <= 3 + i1.kktiter2 / 10);
        std::printf("KKTSolverBasis::Solve tol %.1e: reference %ld its errflag %ld | Hip %ld its errflag %ld | x relerr %.2e y relerr %.2e | "
                    "time_cr2 %.4f s / %.4f s -> %s\n", tol, (long)i1.kktiter2, (long)i1.errflag, (long)i2.kktiter2, (long)i2.errflag, ex, ey,
                    i1.time_cr2, i2.time_cr2, ok ? "PASS" : "FAIL");
        failures += !ok;
    }
    std::printf(failures ? "FAILED\n" : "DONE\n");
    return failures ? 1 : 0;
}
