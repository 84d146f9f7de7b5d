#include "kkt_solver_diag_hip.h"

#include <cassert>

namespace ipx {

namespace {
ipxint PollInterrupt(void* control) {
    return static_cast<const Control*>(control)->InterruptCheck();
}
}  // namespace

KKTSolverDiagHip::KKTSolverDiagHip(const Control& control, const Model& model)
    : control_(control), model_(model), device_(model) {
    // The per-operator timers (time_cr1_AAt, time_cr1_pre) are only printed at debug level >= 2
    // (reference src/lp_solver.cc:107, src/info.cc:20-105); collecting them costs stream markers.
    HipCheck(ipxk_set_profiling(device_.get(), control_.Debug(2) ? 1 : 0));
}

// Builds W, resscale, the normal matrix and the (dense-column aware) diagonal
// preconditioner on the device; see reference src/kkt_solver_diag.cc:18-65.
void KKTSolverDiagHip::_Factorize(Iterate* pt, Info* info) {
    iter_ = 0;
    factorized_ = false;
    ipxint errflag = 0;
    const int dense = control_.precond_dense_cols() ? 1 : 0;
    if (pt) {
        const Vector& xl = pt->xl();
        const Vector& xu = pt->xu();
        const Vector& zl = pt->zl();
        const Vector& zu = pt->zu();
        HipCheck(ipxk_kkt_diag_factorize(device_.get(), &xl[0], &xu[0], &zl[0], &zu[0],
                                         pt->mu(), dense, &errflag));
    } else {
        HipCheck(ipxk_kkt_diag_factorize(device_.get(), nullptr, nullptr, nullptr, nullptr,
                                         0.0, dense, &errflag));
    }
    info->errflag = errflag;          // 0 or IPX_ERROR_lapack_chol
    if (errflag)
        return;
    factorized_ = true;
}

// Reference src/kkt_solver_diag.cc:82-118.
void KKTSolverDiagHip::_Solve(const Vector& a, const Vector& b, double tol,
                               Vector& x, Vector& y, Info* info) {
    assert(factorized_);
    const ipx_hip::SolveOutcome r = ipx_hip::SolveDiagOnDevice(
        device_.get(), &a[0], &b[0], tol, maxiter_, &x[0], &y[0], PollInterrupt,
        const_cast<Control*>(&control_), control_.parameters().debug >= 3);
    if (!r.debug3.empty())
        control_.Debug(3) << r.debug3;      // the reference's messages (src/conjugate_residuals.cc:140-152,198-202)
    info->errflag = r.errflag;
    info->kktiter1 += r.iter;
    info->time_cr1 += r.times.cr;
    info->time_cr1_AAt += r.times.op;
    info->time_cr1_pre += r.times.precond;
    iter_ += r.iter;
}

}  // namespace ipx
