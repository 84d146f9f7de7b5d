#include "kkt_solver_basis_hip.h"

#include <cassert>

#include "sparse_matrix.h"

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
    // return the factorization of basis_ is fresh and variable states are final.
    cpu_.Factorize(iterate, info);
    if (info->errflag)
        return;

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

    // Hand-off of the LU factors: B[rowperm,colperm] = (L+I)*U (src/lu_update.h:43-60).
    SparseMatrix L, U;
    std::vector<Int> rowperm(m), colperm(m);
    basis_.GetLuFactors(&L, &U, rowperm.data(), colperm.data());
    HipCheck(ipxk_split_prepare(device_.get(), L.colptr(), L.rowidx(), L.values(),
                                U.colptr(), U.rowidx(), U.values(), rowperm.data(),
                                colperm.data(), basic.data(), status.data(), colscale.data()));
    factorized_ = true;
}

void KKTSolverBasisHip::_Solve(const Vector& a, const Vector& b, double tol,
                                Vector& x, Vector& y, Info* info) {
    assert(factorized_);
    ipxint iter = 0, errflag = 0;
    ipxk_times times;
    HipCheck(ipxk_kkt_basis_solve(device_.get(), &a[0], &b[0], tol, maxiter_, &x[0], &y[0],
                                  &iter, &errflag, PollInterrupt,
                                  const_cast<Control*>(&control_), &times));
    info->errflag = errflag;
    info->kktiter2 += iter;
    info->time_cr2 += times.cr;
    info->time_cr2_NNt += times.op;
    info->time_cr2_B += times.solve_B;
    info->time_cr2_Bt += times.solve_Bt;
    iter_ += iter;
}

}  // namespace ipx
