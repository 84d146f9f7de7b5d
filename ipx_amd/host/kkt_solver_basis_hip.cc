#include "kkt_solver_basis_hip.h"

#include <cassert>

#include <cstdlib>

#include "sparse_matrix.h"
#include "splitted_normal_matrix.h"
#include "timer.h"

// measurement aid (IPXK_TIME_CPU_PREPARE=1): seconds the reference's own SplittedNormalMatrix::Prepare takes on
// the bases of this run -- the work KKTSolverBasis::_Factorize does for its CPU operator and this class discards
static double g_cpu_prepare_seconds = 0.0;
static long g_cpu_prepare_calls = 0;
extern "C" double ipx_hip_cpu_prepare_seconds() { return g_cpu_prepare_seconds; }
extern "C" long ipx_hip_cpu_prepare_calls() { return g_cpu_prepare_calls; }

namespace ipx {

namespace {
ipxint PollInterrupt(void* control) {
    return static_cast<const Control*>(control)->InterruptCheck();
}
}  // namespace

KKTSolverBasisHip::KKTSolverBasisHip(const Control& control, Basis& basis)
    : control_(control), model_(basis.model()), basis_(basis), cpu_(control, basis),
      device_(basis.model()) {
    // time_cr2_NNt / _B / _Bt are only printed at debug level >= 2 (src/lp_solver.cc:107)
    HipCheck(ipxk_set_profiling(device_.get(), control_.Debug(2) ? 1 : 0));
}

void KKTSolverBasisHip::_Factorize(Iterate* iterate, Info* info) {
    const Int m = model_.rows();
    const Int n = model_.cols();
    factorized_ = false;
    iter_ = 0;

    // Basis maintenance exactly as the reference does it (src/kkt_solver_basis.cc:20-63); on
    // return the factorization of basis_ is fresh and variable states are final.  (The reference's
    // _Factorize is one private function, so this also runs its CPU SplittedNormalMatrix::Prepare,
    // whose result is not used here: duplicated O(nnz) host work per IPM iteration that only a change
    // inside src/kkt_solver_basis.cc could remove -- see INTEGRATION.md.)
    // (KKTSolver::Factorize, src/kkt_solver.cc:8-12, adds its elapsed time to info->time_kkt_factorize; the wrapper
    // around THIS function adds the whole, so the inner share is taken back out)
    const double time_kkt_factorize = info->time_kkt_factorize;
    cpu_.Factorize(iterate, info);
    info->time_kkt_factorize = time_kkt_factorize;
    if (info->errflag)
        return;
    // The device keeps the factors of the previous hand-off.  They are still the factors Basis holds iff no LU
    // factorization happened since (Basis::factorizations(), src/basis.h:214, counts every one -- also a
    // refactorization of an UNCHANGED basis after Basis::TightenLuPivotTol, which basis_changes() == 0 would
    // miss) and no update was applied on top of them (the factorization is fresh, as _Factorize leaves it).
    // Then only the scaling changed.
    const bool same_factors = prepared_once_ && basis_.FactorizationIsFresh() &&
                              basis_.factorizations() == factorizations_at_handoff_;
    factorizations_at_handoff_ = basis_.factorizations();

    // Interior point column scaling after the state changes (src/iterate.cc:183-198): fixed
    // variables scale by 0, free/implied ones by infinity -- the values the reference's
    // drop procedures leave in its colscale_.
    std::vector<double> colscale(n + m);
    std::vector<Int> status(n + m), basic(m);
    for (Int j = 0; j < n + m; j++) {
        colscale[j] = iterate->ScalingFactor(j);
        status[j] = basis_.StatusOf(j);
    }
    for (Int p = 0; p < m; p++)
        basic[p] = basis_[p];
    static const bool time_cpu_prepare = std::getenv("IPXK_TIME_CPU_PREPARE") != nullptr;
    if (time_cpu_prepare) {
        SplittedNormalMatrix probe(model_);
        Timer timer;
        probe.Prepare(basis_, colscale.data());
        g_cpu_prepare_seconds += timer.Elapsed();
        g_cpu_prepare_calls++;
    }

    // Hand-off of the LU factors: B[rowperm,colperm] = (L+I)*U (src/lu_update.h:43-60).
    SparseMatrix L, U;
    std::vector<Int> rowperm(m), colperm(m);
    if (!same_factors)
        basis_.GetLuFactors(&L, &U, rowperm.data(), colperm.data());
    const ipx_hip::BasisHandoff h{m, n, L.colptr(), L.rowidx(), L.values(), U.colptr(), U.rowidx(),
                                  U.values(), rowperm.data(), colperm.data(), basic.data(),
                                  status.data(), colscale.data()};
    ipx_hip::HandOffBasis(device_.get(), h, same_factors);
    prepared_once_ = true;
    factorized_ = true;
}

void KKTSolverBasisHip::_Solve(const Vector& a, const Vector& b, double tol,
                                Vector& x, Vector& y, Info* info) {
    assert(factorized_);
    const ipx_hip::SolveOutcome r = ipx_hip::SolveBasisOnDevice(
        device_.get(), &a[0], &b[0], tol, maxiter_, &x[0], &y[0], PollInterrupt,
        const_cast<Control*>(&control_), control_.parameters().debug >= 3);
    if (!r.debug3.empty())
        control_.Debug(3) << r.debug3;      // the reference's messages (src/conjugate_residuals.cc:53-56)
    info->errflag = r.errflag;
    info->kktiter2 += r.iter;
    info->time_cr2 += r.times.cr;
    info->time_cr2_NNt += r.times.op;
    info->time_cr2_B += r.times.solve_B;
    info->time_cr2_Bt += r.times.solve_Bt;
    iter_ += r.iter;
}

}  // namespace ipx
