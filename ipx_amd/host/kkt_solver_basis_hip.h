// KKTSolverBasisHip: drop-in for ipx::KKTSolverBasis (reference src/kkt_solver_basis.h:21-66).
// The basis maintenance of Factorize (dropping degenerate variables, maxvolume, LU
// refactorization) is sequential pivoting and stays on the CPU in the reference's own class;
// the hand-off of the fresh LU factors (Basis::GetLuFactors, src/basis.cc:162-166) feeds
// ipxk_split_prepare and Solve runs on the GPU (src/kkt_solver_basis.cc:75-194).
#ifndef IPX_KKT_SOLVER_BASIS_HIP_H_
#define IPX_KKT_SOLVER_BASIS_HIP_H_

#include <vector>

#include "basis.h"
#include "control.h"
#include "hip_device.h"
#include "kkt_solver.h"
#include "kkt_solver_basis.h"
#include "model.h"

namespace ipx {

class KKTSolverBasisHip : public KKTSolver {
public:
    KKTSolverBasisHip(const Control& control, Basis& basis);

    Int maxiter() const { return maxiter_; }
    void maxiter(Int new_maxiter) { maxiter_ = new_maxiter; cpu_.maxiter(new_maxiter); }

private:
    void _Factorize(Iterate* iterate, Info* info) override;
    void _Solve(const Vector& a, const Vector& b, double tol,
                Vector& x, Vector& y, Info* info) override;
    Int _iter() const override { return iter_; }
    Int _basis_changes() const override { return cpu_.basis_changes(); }
    const Basis* _basis() const override { return &basis_; }

    const Control& control_;
    const Model& model_;
    Basis& basis_;
    KKTSolverBasis cpu_;        // the reference's Factorize (drop / maxvolume / refactorize)
    HipModel device_;
    bool factorized_{false};
    bool prepared_once_{false};  // the device holds the factors of an earlier hand-off
    Int factorizations_at_handoff_{-1};   // Basis::factorizations() when those factors were handed over
    Int maxiter_{-1};
    Int iter_{0};
};

}  // namespace ipx

#endif  // IPX_KKT_SOLVER_BASIS_HIP_H_
