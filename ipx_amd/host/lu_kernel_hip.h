// ipx::LuKernelHip -- the reference's LuFactorization interface (src/lu_factorization.h:21-58) on the MI355X.
// It takes the place of ipx::BasicLuKernel (src/basiclu_kernel.h:10-18, BASICLU) where Basis::Basis builds its
// LU object for lu_kernel = 1 (src/basis.cc:24-29):
//
//     std::unique_ptr<LuFactorization> lu(new LuKernelHip(ctx));      // was: new BasicLuKernel
//     lu_.reset(new ForrestTomlin(control_, m, lu));
//
// ForrestTomlin (src/forrest_tomlin.cc) then runs its dense SolveDense / FtranForUpdate / BtranForUpdate /
// Update on factors computed on the device, and LuFactorization::Factorize computes its stability estimate on
// them (src/lu_factorization.cc:87-127) exactly as for BASICLU's.
#ifndef IPX_LU_KERNEL_HIP_H_
#define IPX_LU_KERNEL_HIP_H_

#include <memory>

#include "hip_device.h"
#include "ipx_kkt_hip.h"
#include "lu_factorization.h"

namespace ipx {

class LuKernelHip : public LuFactorization {
public:
    // @ctx: a context on the GPU to use (e.g. the one of KKTSolverBasisHip); not owned, must outlive the object
    // @fallback: optional kernel for the bases the device LU declines (a bump beyond its dense limit,
    //            IPXK_E_UNSUPPORTED) -- inside IPX: new BasicLuKernel.  Without one such a basis makes
    //            Factorize() throw std::runtime_error.
    explicit LuKernelHip(ipxk_context* ctx, std::unique_ptr<LuFactorization> fallback = nullptr)
        : ctx_(ctx), fallback_(std::move(fallback)) {}
    // As above, but every Factorize() first asks HipModel for the context a live KKT solver object of this thread holds for the
    // model whose AI() arrays it is handed (hip_device.h) and uses @ctx only when there is none: the factorization that
    // Basis::Load / Basis::Factorize request right after Maxvolume on the device is then not computed a second time
    // (ipxk_lu_info.reused; src/basis.cc:81-114, src/kkt_solver_basis.cc:57-61).
    struct SharedWithSolver {};
    LuKernelHip(SharedWithSolver, ipxk_context* ctx, std::unique_ptr<LuFactorization> fallback = nullptr)
        : ctx_(ctx), fallback_(std::move(fallback)), share_(true) {}

    // # factorizations served from the resident factors of the same basis so far
    Int reused() const { return reused_; }

    // # factorizations handed to the fallback kernel so far
    Int fallbacks() const { return fallbacks_; }

    // statistics of the last factorization (singletons, bump size, phase timings)
    const ipxk_lu_info& info() const { return info_; }

private:
    void _Factorize(Int dim, const Int* Bbegin, const Int* Bend, const Int* Bi, const double* Bx,
                    double pivottol, bool strict_abs_pivottol, SparseMatrix* L, SparseMatrix* U,
                    std::vector<Int>* rowperm, std::vector<Int>* colperm,
                    std::vector<Int>* dependent_cols) override;

    ipxk_context* ctx_;
    std::unique_ptr<LuFactorization> fallback_;
    bool share_{false};
    Int fallbacks_{0}, reused_{0};
    ipxk_lu_info info_{};
};

}  // namespace ipx

#endif  // IPX_LU_KERNEL_HIP_H_
