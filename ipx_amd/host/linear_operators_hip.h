// LinearOperator subclasses over the C ABI (reference src/linear_operator.h:10-24) for
// host-side use: every Apply moves its vectors over PCIe, so the IPM path uses the KKT
// solver classes (whole CR loop on the device); these exist for operator-level tests and for
// callers that want a single product.
#ifndef IPX_LINEAR_OPERATORS_HIP_H_
#define IPX_LINEAR_OPERATORS_HIP_H_

#include <vector>

#include "basis.h"
#include "hip_device.h"
#include "linear_operator.h"
#include "sparse_matrix.h"

namespace ipx {

// reference src/normal_matrix.h:18-43
class NormalMatrixHip : public LinearOperator {
public:
    explicit NormalMatrixHip(HipModel& device) : device_(device) {}
    void Prepare(const double* W) { HipCheck(ipxk_normal_prepare(device_.get(), W)); }
private:
    void _Apply(const Vector& rhs, Vector& lhs, double* rhs_dot_lhs) override {
        HipCheck(ipxk_normal_apply(device_.get(), &rhs[0], &lhs[0], rhs_dot_lhs));
    }
    HipModel& device_;
};

// reference src/diagonal_precond.h:25-57
class DiagonalPrecondHip : public LinearOperator {
public:
    explicit DiagonalPrecondHip(HipModel& device) : device_(device) {}
    void Factorize(const double* W, bool precond_dense_cols, Info* info) {
        ipxint errflag = 0;
        HipCheck(ipxk_diag_factorize(device_.get(), W, precond_dense_cols ? 1 : 0, &errflag));
        info->errflag = errflag;
    }
private:
    void _Apply(const Vector& rhs, Vector& lhs, double* rhs_dot_lhs) override {
        HipCheck(ipxk_diag_apply(device_.get(), &rhs[0], &lhs[0], rhs_dot_lhs));
    }
    HipModel& device_;
};

// reference src/splitted_normal_matrix.h:25-67
class SplittedNormalMatrixHip : public LinearOperator {
public:
    explicit SplittedNormalMatrixHip(HipModel& device) : device_(device) {}
    // SplittedNormalMatrix::Prepare (src/splitted_normal_matrix.cc:18-66): the fresh LU factors of @basis
    // (Basis::GetLuFactors, src/basis.cc:162-166), the basis, the variable statuses and @colscale go to the device
    void Prepare(const Basis& basis, const double* colscale) {
        const Model& model = basis.model();
        const Int m = model.rows(), n = model.cols();
        SparseMatrix L, U;
        std::vector<Int> rowperm(m), colperm(m), status(n + m), basic(m);
        basis.GetLuFactors(&L, &U, rowperm.data(), colperm.data());
        for (Int j = 0; j < n + m; j++) status[j] = basis.StatusOf(j);
        for (Int p = 0; p < m; p++) basic[p] = basis[p];
        HipCheck(ipxk_split_prepare(device_.get(), L.colptr(), L.rowidx(), L.values(), U.colptr(), U.rowidx(), U.values(),
                                    rowperm.data(), colperm.data(), basic.data(), status.data(), colscale));
    }
private:
    void _Apply(const Vector& rhs, Vector& lhs, double* rhs_dot_lhs) override {
        HipCheck(ipxk_split_apply(device_.get(), &rhs[0], &lhs[0], rhs_dot_lhs));
    }
    HipModel& device_;
};

}  // namespace ipx

#endif  // IPX_LINEAR_OPERATORS_HIP_H_
