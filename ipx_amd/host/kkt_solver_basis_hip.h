// KKTSolverBasisHip: drop-in for ipx::KKTSolverBasis (reference src/kkt_solver_basis.h:21-66).
// _Factorize follows the reference's (src/kkt_solver_basis.cc:20-67): the scaling factors, then the reference's OWN
// DropPrimal / DropDual (:196-387; a handful of hypersparse solves on the reference's Basis, CPU) -- this class holds
// a KKTSolverBasis object and calls its two private members, for which src/kkt_solver_basis.h gains ONE line,
//     friend class KKTSolverBasisHip;
// (the test harness applies it with sed on the way into the compiler, oracle/Makefile) -- nothing
// of those functions is restated here.  Then Maxvolume ON THE DEVICE (ipxk_maxvolume / ipxk_maxvolume_sequential: the identical decisions as
// ipx::Maxvolume, pinned against it up to 1M rows) from a device factorization of the current basis, which ends with
// the split operator of the final basis (a fresh factorization, or the earlier factors with the last exchanges behind them
// as etas when that is cheaper: ipxk_maxvolume_info.kept_etas); the final basis goes back into the reference's
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
#include "kkt_solver_basis.h"
#include "model.h"

namespace ipx {

class KKTSolverBasisHip : public KKTSolver {
public:
    KKTSolverBasisHip(const Control& control, Basis& basis);
    ~KKTSolverBasisHip();

    Int maxiter() const { return maxiter_; }
    void maxiter(Int new_maxiter) { maxiter_ = new_maxiter; }

    // The reference's Basis learns the basis of the last Factorize() only when it is needed there (see basis_pending_ below);
    // code that reads the Basis itself while this solver object is alive calls this first.  Returns Basis::Load's error code.
    Int FlushBasis() { return SyncBasis(); }

    // # Factorize calls whose Maxvolume ran on the device / on the reference's Basis (CPU) so far
    Int device_maxvolume_calls() const { return device_maxvolume_calls_; }
    Int cpu_maxvolume_calls() const { return cpu_maxvolume_calls_; }

private:
    void _Factorize(Iterate* iterate, Info* info) override;
    void _Solve(const Vector& a, const Vector& b, double tol,
                Vector& x, Vector& y, Info* info) override;
    Int _iter() const override { return iter_; }
    Int _basis_changes() const override;
    const Basis* _basis() const override;

    // the reference's DropPrimal / DropDual (src/kkt_solver_basis.cc:196-387) on reference_'s colscale_ / basis_changes_
    void DropDegenerateVariables(Iterate* iterate, Info* info);
    // true if DropPrimal / DropDual would find a candidate in the basis the device holds (the screening conditions of
    // src/kkt_solver_basis.cc:208-216, 306-314 on the iterate; nothing is pivoted here)
    bool DegenerateCandidateExists(const Iterate& iterate) const;
    // hands the basis the device holds to the reference's Basis (Basis::Load); returns its error code
    Int SyncBasis();
    // Maxvolume + the operator of the final basis on the device; false: declined (nothing changed, take the CPU path)
    bool MaxvolumeOnDevice(Info* info);
    void MaxvolumeOnBasis(Info* info);

    const Control& control_;
    const Model& model_;
    Basis& basis_;
    HipModel device_;
    // The reference's solver object on the same Basis: owner of the scaling factors (colscale_) and of the basis change
    // counter, which its DropPrimal / DropDual read and update.  Its _Factorize / _Solve are never called.
    KKTSolverBasis reference_;
    bool factorized_{false};
    bool prepared_once_{false};  // the device holds the factors of an earlier hand-off from Basis::GetLuFactors
    Int factorizations_at_handoff_{-1};   // Basis::factorizations() when those factors were handed over
    std::vector<signed char> device_member_;   // per variable: 1 if in the basis whose factors the device LU holds
    // The basis after Maxvolume on the device, as the device holds it.  basis_pending_: the reference's Basis has not been told yet --
    // Basis::Load costs a pass of the reference's own host code over the factors (download, stability estimate: 0.4 s at 24 000 rows
    // with 54 M entries), and between two Factorize calls nobody asks the Basis anything unless DropPrimal / DropDual have a
    // candidate.  So the Load happens when one exists, when IPM prints Basis statistics (Debug(4)), on the CPU path, and at the latest
    // in the destructor (crossover and LpSolver read the Basis afterwards).
    std::vector<Int> device_status_, device_basis_;
    bool basis_pending_{false};
    bool device_lu_valid_{false};
    Int device_lu_generation_{-1};              // ipxk_lu_generation() when those factors were computed: any other factorization through the context since then invalidates them
    Int maxiter_{-1};
    Int iter_{0};
    Int device_maxvolume_calls_{0}, cpu_maxvolume_calls_{0};
};

}  // namespace ipx

#endif  // IPX_KKT_SOLVER_BASIS_HIP_H_
