// KKTSolverBasisHip: drop-in for ipx::KKTSolverBasis (reference src/kkt_solver_basis.h:21-66).
// _Factorize restates the reference's (src/kkt_solver_basis.cc:20-67) on the public members of Basis / Iterate:
// the scaling factors, DropPrimal / DropDual (:196-387; a handful of hypersparse solves on the reference's own
// Basis, CPU), then Maxvolume ON THE DEVICE (ipxk_maxvolume / ipxk_maxvolume_sequential: the identical decisions as
// ipx::Maxvolume, pinned against it up to 1M rows) from a device factorization of the current basis, which ends with
// the fresh factorization and the split operator of the final basis; the final basis goes back into the reference's
// Basis with Basis::Load (src/basis.h:86-94: loads and factorizes -- the factorization the reference performs at
// this point anyway, :57-61).  The reference's CPU Maxvolume and its CPU SplittedNormalMatrix::Prepare leave the
// main phase.  A basis the device LU declines (IPXK_E_UNSUPPORTED, dependent columns) takes the reference's
// Maxvolume on its Basis and the hand-off of Basis::GetLuFactors instead (IPXK_DEVICE_MAXVOLUME=0 forces that path).
// _Solve runs on the GPU (src/kkt_solver_basis.cc:75-194).
#ifndef IPX_KKT_SOLVER_BASIS_HIP_H_
#define IPX_KKT_SOLVER_BASIS_HIP_H_

#include <vector>

#include "basis.h"
#include "control.h"
#include "hip_device.h"
#include "kkt_solver.h"
#include "model.h"

namespace ipx {

class KKTSolverBasisHip : public KKTSolver {
public:
    KKTSolverBasisHip(const Control& control, Basis& basis);
    ~KKTSolverBasisHip();

    Int maxiter() const { return maxiter_; }
    void maxiter(Int new_maxiter) { maxiter_ = new_maxiter; }

    // # Factorize calls whose Maxvolume ran on the device / on the reference's Basis (CPU) so far
    Int device_maxvolume_calls() const { return device_maxvolume_calls_; }
    Int cpu_maxvolume_calls() const { return cpu_maxvolume_calls_; }

private:
    void _Factorize(Iterate* iterate, Info* info) override;
    void _Solve(const Vector& a, const Vector& b, double tol,
                Vector& x, Vector& y, Info* info) override;
    Int _iter() const override { return iter_; }
    Int _basis_changes() const override { return basis_changes_; }
    const Basis* _basis() const override { return &basis_; }

    // src/kkt_solver_basis.cc:196-387 on the public interface of Basis / Iterate
    void DropPrimal(Iterate* iterate, Info* info);
    void DropDual(Iterate* iterate, Info* info);
    // Maxvolume + fresh factorization + operator on the device; false: declined (nothing changed, take the CPU path)
    bool MaxvolumeOnDevice(Info* info);
    void MaxvolumeOnBasis(Info* info);

    static constexpr double kPivotZeroTol = 1e-7;      // src/kkt_solver_basis.h:33

    const Control& control_;
    const Model& model_;
    Basis& basis_;
    HipModel device_;
    Vector colscale_;            // interior point column scaling factors (src/kkt_solver_basis.h:60)
    bool factorized_{false};
    bool prepared_once_{false};  // the device holds the factors of an earlier hand-off from Basis::GetLuFactors
    Int factorizations_at_handoff_{-1};   // Basis::factorizations() when those factors were handed over
    std::vector<signed char> device_member_;   // per variable: 1 if in the basis whose factors the device LU holds
    bool device_lu_valid_{false};
    Int maxiter_{-1};
    Int iter_{0};
    Int basis_changes_{0};
    Int device_maxvolume_calls_{0}, cpu_maxvolume_calls_{0};
};

}  // namespace ipx

#endif  // IPX_KKT_SOLVER_BASIS_HIP_H_
