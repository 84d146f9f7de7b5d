// KKTSolverDiagHip: drop-in for ipx::KKTSolverDiag (reference src/kkt_solver_diag.h:23-49)
// whose Factorize/Solve run on an MI355X through the C ABI of ipx_kkt_hip.h.  Same
// constructor, same maxiter() accessors, same Info bookkeeping, same error channel.
#ifndef IPX_KKT_SOLVER_DIAG_HIP_H_
#define IPX_KKT_SOLVER_DIAG_HIP_H_

#include "control.h"
#include "hip_device.h"
#include "kkt_solver.h"
#include "model.h"

namespace ipx {

class KKTSolverDiagHip : public KKTSolver {
public:
    KKTSolverDiagHip(const Control& control, const Model& model);

    Int maxiter() const { return maxiter_; }
    void maxiter(Int new_maxiter) { maxiter_ = new_maxiter; }

private:
    void _Factorize(Iterate* iterate, Info* info) override;
    void _Solve(const Vector& a, const Vector& b, double tol,
                Vector& x, Vector& y, Info* info) override;
    Int _iter() const override { return iter_; }

    const Control& control_;
    const Model& model_;
    HipModel device_;
    bool factorized_{false};
    Int maxiter_{-1};
    Int iter_{0};
};

}  // namespace ipx

#endif  // IPX_KKT_SOLVER_DIAG_HIP_H_
