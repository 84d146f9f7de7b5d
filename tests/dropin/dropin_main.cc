// TEST INFRASTRUCTURE.  Drop-in check at the reference's own boundary: the reference's
// Model / Iterate / Control objects (from oracle/_ref, built from the reference's sources) drive
// BOTH ipx::KKTSolverDiag (reference, CPU) and ipx::KKTSolverDiagHip (this repo, MI355X) through
// the abstract ipx::KKTSolver interface (Factorize / Solve / iter), and the results are compared.
// Built by `make -C oracle dropin` into oracle/_ref/test_dropin; run by tests/test_gpu_dropin.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "conjugate_residuals.h"
#include "control.h"
#include "diagonal_precond.h"
#include "iterate.h"
#include "kkt_solver_diag.h"
#include "kkt_solver_diag_hip.h"
#include "linear_operators_hip.h"
#include "model.h"
#include "normal_matrix.h"
#include "presolver.h"
#include "user_model.h"

using ipx::Int;
using ipx::Vector;

static double RelErr(const Vector& a, const Vector& b) {
    double num = 0.0, den = 0.0;
    for (size_t i = 0; i < a.size(); i++) {
        num = std::max(num, std::abs(a[i] - b[i]));
        den = std::max(den, std::abs(b[i]));
    }
    return den > 0.0 ? num / den : num;
}

struct Result { Vector x, y; Int iter; Int errflag; double time_cr1; Int kktiter1; };

static Result Run(ipx::KKTSolver& kkt, ipx::Iterate* iterate, const Vector& a, const Vector& b,
                  double tol, Int n, Int m) {
    ipx::Info info;
    Result r;
    r.x.resize(n + m);
    r.y.resize(m);
    kkt.Factorize(iterate, &info);
    if (info.errflag) { r.errflag = info.errflag; r.iter = -1; return r; }
    kkt.Solve(a, b, tol, r.x, r.y, &info);
    r.iter = kkt.iter();
    r.errflag = info.errflag;
    r.time_cr1 = info.time_cr1;
    r.kktiter1 = info.kktiter1;
    return r;
}

int main(int argc, char** argv) {
    const Int m = argc > 1 ? atol(argv[1]) : 3000;
    const Int n = argc > 2 ? atol(argv[2]) : 7000;
    const Int k = 8;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> uval(0.5, 4.0), u01(-1.0, 1.0), uab(-0.5, 0.5);
    std::vector<Int> Ap(n + 1), Ai;
    std::vector<double> Ax;
    for (Int j = 0; j < n; j++) {
        Ap[j] = (Int)Ai.size();
        std::vector<Int> rows;
        while ((Int)rows.size() < std::min(k, m)) {
            Int r = (Int)(rng() % (uint64_t)m);
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
    control.parameters(params);
    ipx::UserModel user_model;
    ipx::Model model;
    if (user_model.Load(control, m, n, Ap.data(), Ai.data(), Ax.data(), rhs.data(), ct.data(),
                        obj.data(), lb.data(), ub.data()) != 0) return 2;
    ipx::Presolver presolver(user_model, model);
    if (presolver.PresolveModel(control) != 0 || model.rows() != m || model.cols() != n) return 2;

    Vector x0(1.0, n + m), y0(0.0, m), xl(n + m), xu(INFINITY, n + m), zl(n + m), zu(0.0, n + m);
    for (Int j = 0; j < n + m; j++) { xl[j] = std::pow(10.0, u01(rng)); zl[j] = std::pow(10.0, u01(rng)); }
    ipx::Iterate iterate(model);
    iterate.Initialize(x0, xl, xu, y0, zl, zu);
    Vector a(n + m), b(m);
    for (auto& v : a) v = uab(rng);
    for (auto& v : b) v = uab(rng);
    const double tol = 0.3 * std::sqrt(iterate.mu());

    int failures = 0;
    for (int pass = 0; pass < 2; pass++) {
        ipx::Iterate* it = pass == 0 ? &iterate : nullptr;      // pass 1: Factorize(nullptr)
        const double t = pass == 0 ? tol : 1e-6;
        ipx::KKTSolverDiag cpu(control, model);
        ipx::KKTSolverDiagHip gpu(control, model);
        cpu.maxiter(500);
        gpu.maxiter(500);
        Result rc = Run(cpu, it, a, b, t, n, m);
        Result rg = Run(gpu, it, a, b, t, n, m);
        const double ey = RelErr(rg.y, rc.y), ex = RelErr(rg.x, rc.x);
        const bool ok = rc.errflag == rg.errflag && std::labs((long)(rc.iter - rg.iter)) <= 2 &&
                        ey < 1e-6 && ex < 1e-5 && rg.kktiter1 == rg.iter && gpu.maxiter() == 500;
        printf("%s: cpu iter %ld errflag %ld | hip iter %ld errflag %ld | y relerr %.2e x relerr %.2e | "
               "time_cr1 cpu %.4fs hip %.4fs -> %s\n",
               pass == 0 ? "Factorize(iterate)" : "Factorize(nullptr)", (long)rc.iter, (long)rc.errflag,
               (long)rg.iter, (long)rg.errflag, ey, ex, rc.time_cr1, rg.time_cr1, ok ? "PASS" : "FAIL");
        failures += !ok;
    }
    // ---- operator level (reference src/linear_operator.h:10-24): the reference's OWN
    //      ConjugateResiduals loop, running on the host, applies the operator and the preconditioner
    //      through LinearOperator& -- once its own NormalMatrix / DiagonalPrecond, once the HIP ones
    {
        Vector W(n + m), resscale(m), rhs(m);
        for (Int j = 0; j < n + m; j++) W[j] = xl[j] / zl[j];
        for (Int i = 0; i < m; i++) { resscale[i] = 1.0 / std::sqrt(W[n + i]); rhs[i] = uab(rng); }
        ipx::Info info;
        ipx::NormalMatrix C_cpu(model);
        ipx::DiagonalPrecond P_cpu(model);
        C_cpu.Prepare(&W[0]);
        P_cpu.Factorize(&W[0], false, &info);
        ipx::HipModel device(model);
        ipx::NormalMatrixHip C_hip(device);
        ipx::DiagonalPrecondHip P_hip(device);
        C_hip.Prepare(&W[0]);
        P_hip.Factorize(&W[0], false, &info);
        // single applications
        Vector l1(m), l2(m), p1(m), p2(m);
        double d1 = 0, d2 = 0, e1 = 0, e2 = 0;
        ipx::LinearOperator &Cc = C_cpu, &Ch = C_hip, &Pc = P_cpu, &Ph = P_hip;
        Cc.Apply(rhs, l1, &d1); Ch.Apply(rhs, l2, &d2);
        Pc.Apply(rhs, p1, &e1); Ph.Apply(rhs, p2, &e2);
        const bool ok_apply = RelErr(l2, l1) <= 1e-12 && RelErr(p2, p1) <= 1e-12 &&
                              std::abs(d1 - d2) <= 1e-12 * std::abs(d1) && std::abs(e1 - e2) <= 1e-12 * std::abs(e1);
        // the reference's CR loop over both operator pairs
        ipx::ConjugateResiduals cr_cpu(control), cr_hip(control);
        Vector y1(0.0, m), y2(0.0, m);
        cr_cpu.Solve(Cc, Pc, rhs, tol, &resscale[0], 500, y1);
        cr_hip.Solve(Ch, Ph, rhs, tol, &resscale[0], 500, y2);
        const bool ok_cr = cr_cpu.errflag() == cr_hip.errflag() &&
                           std::labs((long)(cr_cpu.iter() - cr_hip.iter())) <= 2 && RelErr(y2, y1) < 1e-6;
        printf("LinearOperator: apply relerr %.2e precond relerr %.2e | reference CR loop over reference operators: "
               "%ld its errflag %ld, over HIP operators: %ld its errflag %ld, y relerr %.2e -> %s\n",
               RelErr(l2, l1), RelErr(p2, p1), (long)cr_cpu.iter(), (long)cr_cpu.errflag(), (long)cr_hip.iter(),
               (long)cr_hip.errflag(), RelErr(y2, y1), ok_apply && ok_cr ? "PASS" : "FAIL");
        failures += !(ok_apply && ok_cr);
    }
    // ---- two DIFFERENT models with equal dimensions, equal nnz and equal colptr back to back (the registry of
    //      hip_device.h must not hand the first model's device matrix to the second): model B = the model above with
    //      ONE value changed, model C = with ONE row index changed.  Each is solved by a fresh KKTSolverDiagHip and
    //      by the reference's KKTSolverDiag; a cached context of the first model is idle at that time.
    {
        const long creations_before = ipx::HipModel::creations(), hits_before = ipx::HipModel::hits();
        { ipx::KKTSolverDiagHip again(control, model); }              // the same model once more: served from the cache
        const bool hit = ipx::HipModel::hits() == hits_before + 1 && ipx::HipModel::creations() == creations_before;
        bool ok_models = hit;
        for (int variant = 0; variant < 2; variant++) {
            std::vector<Int> Ai2 = Ai;
            std::vector<double> Ax2 = Ax;
            const Int j = n / 2, p = Ap[j];
            if (variant == 0) Ax2[p] = -Ax2[p] * 1.5;
            else {                                                     // another row for the column's first entry
                Int r = (Ai2[p] + 1) % m;
                for (bool clash = true; clash; ) {
                    clash = false;
                    for (Int q = Ap[j]; q < Ap[j + 1]; q++) if (q != p && Ai2[q] == r) { clash = true; r = (r + 1) % m; }
                }
                Ai2[p] = r;
            }
            ipx::UserModel um2;
            ipx::Model model2;
            if (um2.Load(control, m, n, Ap.data(), Ai2.data(), Ax2.data(), rhs.data(), ct.data(), obj.data(), lb.data(),
                         ub.data()) != 0) return 2;
            ipx::Presolver pre2(um2, model2);
            if (pre2.PresolveModel(control) != 0 || model2.rows() != m || model2.cols() != n) return 2;
            ipx::Iterate it2(model2);
            it2.Initialize(x0, xl, xu, y0, zl, zu);
            const long c0 = ipx::HipModel::creations();
            ipx::KKTSolverDiag cpu(control, model2);
            ipx::KKTSolverDiagHip gpu(control, model2);
            cpu.maxiter(500);
            gpu.maxiter(500);
            Result rc = Run(cpu, &it2, a, b, tol, n, m);
            Result rg = Run(gpu, &it2, a, b, tol, n, m);
            const bool fresh = ipx::HipModel::creations() == c0 + 1;       // not the first model's context
            const bool same = rc.errflag == rg.errflag && std::labs((long)(rc.iter - rg.iter)) <= 2 &&
                              RelErr(rg.y, rc.y) < 1e-6 && RelErr(rg.x, rc.x) < 1e-5;
            printf("model %c (one %s changed): own device model %s, y relerr %.2e x relerr %.2e\n", 'B' + variant,
                   variant == 0 ? "value" : "row index", fresh ? "yes" : "NO", RelErr(rg.y, rc.y), RelErr(rg.x, rc.x));
            ok_models = ok_models && fresh && same;
        }
        printf("equal dimensions / nnz / colptr, different content: cache hit for the same model %s -> %s\n", hit ? "yes" : "NO",
               ok_models ? "PASS" : "FAIL");
        failures += !ok_models;
    }
    return failures ? 1 : 0;
}
