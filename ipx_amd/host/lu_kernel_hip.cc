#include "lu_kernel_hip.h"

#include "device_glue.h"

namespace ipx {

static_assert(sizeof(Int) == sizeof(ipxint), "IPX integer type does not match the device library's");

void LuKernelHip::_Factorize(Int dim, const Int* Bbegin, const Int* Bend, const Int* Bi, const double* Bx,
                             double pivottol, bool strict_abs_pivottol, SparseMatrix* L, SparseMatrix* U,
                             std::vector<Int>* rowperm, std::vector<Int>* colperm,
                             std::vector<Int>* dependent_cols) {
    // Errors: out of memory -> std::bad_alloc (the one failure lu_factorization.h:49-50 allows); a bump beyond
    // the dense limit (IPXK_E_UNSUPPORTED) -> std::runtime_error, i.e. IPX_STATUS_internal_error at
    // src/lp_solver.cc:98-105 -- a caller that wants to go on would factorize that basis with BasicLuKernel.
    ipxk_context* ctx = ctx_;
    if (share_)
        if (ipxk_context* solver_ctx = HipModel::InUse(Bi, Bx))
            ctx = solver_ctx;
    const int rc = ipxk_lu_factorize(ctx, dim, Bbegin, Bend, Bi, Bx, pivottol, strict_abs_pivottol ? 1 : 0, &info_);
    if (rc == IPXK_E_UNSUPPORTED && fallback_) {      // not a nearly triangular basis: the CPU kernel takes it
        fallbacks_++;
        fallback_->Factorize(dim, Bbegin, Bend, Bi, Bx, pivottol, strict_abs_pivottol, L, U, rowperm, colperm,
                             dependent_cols);
        return;
    }
    ipx_hip::Check(rc);
    reused_ += info_.reused;
    L->resize(dim, dim, info_.lnz);
    U->resize(dim, dim, info_.unz);
    rowperm->resize(dim);
    colperm->resize(dim);
    dependent_cols->resize(info_.num_dependent);
    ipx_hip::Check(ipxk_lu_get_factors(ctx, L->colptr(), L->rowidx(), L->values(), U->colptr(), U->rowidx(),
                                       U->values(), rowperm->data(), colperm->data(), dependent_cols->data()));
}

}  // namespace ipx
