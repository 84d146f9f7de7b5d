// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product path.
//
// Thin extern "C" driver over the *reference's own objects*.  It is compiled by
// oracle/Makefile against the headers under /root/reference/{include,src} and
// linked with the reference's object files built from the sources where they
// lie (nothing is copied into this repo).  The resulting oracle/_ref/libipx_ref.so
// is used (i) by tests/golden/make_golden.py to produce the committed golden
// vectors and (ii) by the tests to validate oracle/ipx_oracle.cc.
//
// What can be driven here:
//   * everything on the diag path (Model via UserModel::Load + Presolver,
//     NormalMatrix, DiagonalPrecond, ConjugateResiduals, KKTSolverDiag);
//   * the sparse kernels of the basis path (TriangularSolve, ForwardSolve,
//     BackwardSolve, AddNormalProduct, Transpose, CopyColumns, PermuteRows,
//     InversePerm) and ConjugateResiduals (plain CR).
// What cannot: SplittedNormalMatrix::Prepare / KKTSolverBasis need a live
// ipx::Basis, which needs BASICLU (absent from the image) -> "unbuildable".
// For those rows the operator below composes the reference's kernels in the
// order of src/splitted_normal_matrix.cc:90-117 from explicit (L,U,N) inputs.

#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "conjugate_residuals.h"
#include "control.h"
#include "diagonal_precond.h"
#include "forrest_tomlin.h"
#include "indexed_vector.h"
#include "iterate.h"
#include "kkt_solver_diag.h"
#include "lu_factorization.h"
#include "model.h"
#include "normal_matrix.h"
#include "presolver.h"
#include "sparse_matrix.h"
#include "user_model.h"
#include "utils.h"

using ipx::Int;
using ipx::Vector;

namespace {

struct RefModel {
    ipx::Control control;
    ipx::UserModel user_model;
    ipx::Model model;
    std::unique_ptr<ipx::Presolver> presolver;
    RefModel() {
        ipx::Parameters p;
        p.display = 0;
        control.parameters(p);
    }
};

Vector ToVector(const double* x, Int n) {
    Vector v(n);
    if (n > 0) std::memcpy(&v[0], x, sizeof(double) * n);
    return v;
}
void FromVector(const Vector& v, double* x) {
    if (v.size() > 0) std::memcpy(x, &v[0], sizeof(double) * v.size());
}

ipx::SparseMatrix ToCsc(Int nrow, Int ncol, const Int* Ap, const Int* Ai,
                        const double* Ax) {
    ipx::SparseMatrix A(nrow, ncol, Ap[ncol]);
    std::memcpy(A.colptr(), Ap, sizeof(Int) * (ncol + 1));
    std::memcpy(A.rowidx(), Ai, sizeof(Int) * Ap[ncol]);
    std::memcpy(A.values(), Ax, sizeof(double) * Ap[ncol]);
    return A;
}

// Records the scalar returned by every Apply() so that CR trajectories can be
// compared iteration by iteration.
class RecordingOperator : public ipx::LinearOperator {
public:
    RecordingOperator(ipx::LinearOperator& op, double* hist, Int cap)
        : op_(op), hist_(hist), cap_(cap) {}
    Int calls() const { return calls_; }
private:
    void _Apply(const Vector& rhs, Vector& lhs, double* dot) override {
        double d = 0.0;
        op_.Apply(rhs, lhs, &d);
        if (hist_ && calls_ < cap_) hist_[calls_] = d;
        calls_++;
        if (dot) *dot = d;
    }
    ipx::LinearOperator& op_;
    double* hist_;
    Int cap_;
    Int calls_{0};
};

// C = I + inv(B) N N' inv(B') from explicit factors; the three steps are the
// reference's own BackwardSolve / AddNormalProduct / ForwardSolve.
class SplitOperatorFromFactors : public ipx::LinearOperator {
public:
    ipx::SparseMatrix L, U, N;
    std::vector<Int> free_positions;
    Vector work;
private:
    void _Apply(const Vector& rhs, Vector& lhs, double* dot) override {
        work = rhs;
        ipx::BackwardSolve(L, U, work);
        lhs = 0.0;
        ipx::AddNormalProduct(N, nullptr, work, lhs);
        ipx::ForwardSolve(L, U, lhs);
        lhs += rhs;
        for (Int i : free_positions) lhs[i] = 0.0;
        if (dot) *dot = ipx::Dot(rhs, lhs);
    }
};

struct RefKktDiag {
    RefModel* rm;
    ipx::Control control;
    std::unique_ptr<ipx::KKTSolverDiag> kkt;
    std::unique_ptr<ipx::Iterate> iterate;
};

}  // namespace

extern "C" {

// ---- model ---------------------------------------------------------------
void* ref_model_new(Int num_constr, Int num_var, const Int* Ap, const Int* Ai,
                    const double* Ax, const double* rhs,
                    const char* constr_type, const double* obj,
                    const double* lb, const double* ub, Int* dims) {
    auto* rm = new RefModel;
    Int err = rm->user_model.Load(rm->control, num_constr, num_var, Ap, Ai, Ax,
                                  rhs, constr_type, obj, lb, ub);
    if (err) { dims[0] = -err; delete rm; return nullptr; }
    rm->presolver.reset(new ipx::Presolver(rm->user_model, rm->model));
    err = rm->presolver->PresolveModel(rm->control);
    if (err) { dims[0] = -err; delete rm; return nullptr; }
    dims[0] = rm->model.rows();
    dims[1] = rm->model.cols();
    dims[2] = rm->model.AI().entries();
    dims[3] = rm->model.dualized();
    dims[4] = rm->model.num_dense_cols();
    return rm;
}
void ref_model_free(void* h) { delete static_cast<RefModel*>(h); }

void ref_model_get_AI(void* h, Int* Ap, Int* Ai, double* Ax) {
    const ipx::SparseMatrix& AI = static_cast<RefModel*>(h)->model.AI();
    std::memcpy(Ap, AI.colptr(), sizeof(Int) * (AI.cols() + 1));
    std::memcpy(Ai, AI.rowidx(), sizeof(Int) * AI.entries());
    std::memcpy(Ax, AI.values(), sizeof(double) * AI.entries());
}
void ref_model_get_AIt(void* h, Int* Ap, Int* Ai, double* Ax) {
    const ipx::SparseMatrix& AIt = static_cast<RefModel*>(h)->model.AIt();
    std::memcpy(Ap, AIt.colptr(), sizeof(Int) * (AIt.cols() + 1));
    std::memcpy(Ai, AIt.rowidx(), sizeof(Int) * AIt.entries());
    std::memcpy(Ax, AIt.values(), sizeof(double) * AIt.entries());
}
void ref_model_get_vectors(void* h, double* b, double* c, double* lb,
                           double* ub) {
    const ipx::Model& model = static_cast<RefModel*>(h)->model;
    FromVector(model.b(), b);
    FromVector(model.c(), c);
    FromVector(model.lb(), lb);
    FromVector(model.ub(), ub);
}
Int ref_model_is_dense(void* h, Int j) {
    return static_cast<RefModel*>(h)->model.IsDenseColumn(j) ? 1 : 0;
}

// ---- NormalMatrix / DiagonalPrecond --------------------------------------
void ref_normal_apply(void* h, const double* W, const double* rhs, double* lhs,
                      double* dot) {
    const ipx::Model& model = static_cast<RefModel*>(h)->model;
    ipx::NormalMatrix C(model);
    C.Prepare(W);
    Vector r = ToVector(rhs, model.rows());
    Vector l(model.rows());
    C.Apply(r, l, dot);
    FromVector(l, lhs);
}

// Factorize + one Apply; exports nothing of the private state but the result.
Int ref_diagprec_apply(void* h, const double* W, Int precond_dense_cols,
                       const double* rhs, double* lhs, double* dot) {
    const ipx::Model& model = static_cast<RefModel*>(h)->model;
    ipx::DiagonalPrecond P(model);
    ipx::Info info;
    P.Factorize(W, precond_dense_cols != 0, &info);
    if (info.errflag) return info.errflag;
    Vector r = ToVector(rhs, model.rows());
    Vector l(model.rows());
    P.Apply(r, l, dot);
    FromVector(l, lhs);
    return 0;
}

// Preconditioned CR on C = AI W AI' with P = DiagonalPrecond.  cdot_hist /
// pdot_hist receive the scalar of every C.Apply / P.Apply call in call order.
Int ref_pcr_solve(void* h, const double* W, Int precond_dense_cols,
                  const double* rhs, double tol, const double* resscale,
                  Int maxiter, double* lhs, Int* iter, double* cdot_hist,
                  double* pdot_hist, Int hist_cap, Int* ncalls) {
    RefModel* rm = static_cast<RefModel*>(h);
    const ipx::Model& model = rm->model;
    const Int m = model.rows();
    ipx::NormalMatrix C(model);
    C.Prepare(W);
    ipx::DiagonalPrecond P(model);
    ipx::Info info;
    P.Factorize(W, precond_dense_cols != 0, &info);
    if (info.errflag) return info.errflag;
    RecordingOperator Crec(C, cdot_hist, hist_cap);
    RecordingOperator Prec(P, pdot_hist, hist_cap);
    ipx::ConjugateResiduals cr(rm->control);
    Vector r = ToVector(rhs, m);
    Vector l = ToVector(lhs, m);
    cr.Solve(Crec, Prec, r, tol, resscale, maxiter, l);
    FromVector(l, lhs);
    *iter = cr.iter();
    if (ncalls) { ncalls[0] = Crec.calls(); ncalls[1] = Prec.calls(); }
    return cr.errflag();
}

// ---- KKTSolverDiag --------------------------------------------------------
void* ref_kktdiag_new(void* h, Int maxiter, Int precond_dense_cols) {
    auto* k = new RefKktDiag;
    k->rm = static_cast<RefModel*>(h);
    ipx::Parameters p;
    p.display = 0;
    p.precond_dense_cols = precond_dense_cols;
    k->control.parameters(p);
    k->kkt.reset(new ipx::KKTSolverDiag(k->control, k->rm->model));
    k->kkt->maxiter(maxiter);
    return k;
}
void ref_kktdiag_free(void* k) { delete static_cast<RefKktDiag*>(k); }

// x == NULL -> Factorize(nullptr) (G = identity).
Int ref_kktdiag_factorize(void* kh, const double* x, const double* xl,
                          const double* xu, const double* y, const double* zl,
                          const double* zu, double* mu_out) {
    RefKktDiag* k = static_cast<RefKktDiag*>(kh);
    const ipx::Model& model = k->rm->model;
    const Int m = model.rows(), n = model.cols();
    ipx::Info info;
    if (!x) {
        k->kkt->Factorize(nullptr, &info);
        return info.errflag;
    }
    k->iterate.reset(new ipx::Iterate(model));
    k->iterate->Initialize(ToVector(x, n + m), ToVector(xl, n + m),
                           ToVector(xu, n + m), ToVector(y, m),
                           ToVector(zl, n + m), ToVector(zu, n + m));
    if (mu_out) *mu_out = k->iterate->mu();
    k->kkt->Factorize(k->iterate.get(), &info);
    return info.errflag;
}

Int ref_kktdiag_solve(void* kh, const double* a, const double* b, double tol,
                      double* x, double* y, Int* iter) {
    RefKktDiag* k = static_cast<RefKktDiag*>(kh);
    const ipx::Model& model = k->rm->model;
    const Int m = model.rows(), n = model.cols();
    ipx::Info info;
    Vector xv(n + m), yv(m);
    Int before = k->kkt->iter();
    k->kkt->Solve(ToVector(a, n + m), ToVector(b, m), tol, xv, yv, &info);
    FromVector(xv, x);
    FromVector(yv, y);
    *iter = k->kkt->iter() - before;
    return info.errflag;
}

// ---- sparse kernels ---------------------------------------------------------
Int ref_trisolve(Int dim, const Int* Ap, const Int* Ai, const double* Ax,
                 double* x, char trans, char uplo, Int unitdiag) {
    ipx::SparseMatrix A = ToCsc(dim, dim, Ap, Ai, Ax);
    Vector xv = ToVector(x, dim);
    const char u[2] = {uplo, 0};
    Int nz = ipx::TriangularSolve(A, xv, trans, u, (int)unitdiag);
    FromVector(xv, x);
    return nz;
}
void ref_forward_solve(Int dim, const Int* Lp, const Int* Li, const double* Lx,
                       const Int* Up, const Int* Ui, const double* Ux,
                       double* x) {
    ipx::SparseMatrix L = ToCsc(dim, dim, Lp, Li, Lx);
    ipx::SparseMatrix U = ToCsc(dim, dim, Up, Ui, Ux);
    Vector xv = ToVector(x, dim);
    ipx::ForwardSolve(L, U, xv);
    FromVector(xv, x);
}
void ref_backward_solve(Int dim, const Int* Lp, const Int* Li, const double* Lx,
                        const Int* Up, const Int* Ui, const double* Ux,
                        double* x) {
    ipx::SparseMatrix L = ToCsc(dim, dim, Lp, Li, Lx);
    ipx::SparseMatrix U = ToCsc(dim, dim, Up, Ui, Ux);
    Vector xv = ToVector(x, dim);
    ipx::BackwardSolve(L, U, xv);
    FromVector(xv, x);
}
void ref_add_normal_product(Int nrow, Int ncol, const Int* Ap, const Int* Ai,
                            const double* Ax, const double* D,
                            const double* rhs, double* lhs) {
    ipx::SparseMatrix A = ToCsc(nrow, ncol, Ap, Ai, Ax);
    Vector r = ToVector(rhs, nrow);
    Vector l = ToVector(lhs, nrow);
    ipx::AddNormalProduct(A, D, r, l);
    FromVector(l, lhs);
}
void ref_multiply_add(Int nrow, Int ncol, const Int* Ap, const Int* Ai,
                      const double* Ax, const double* rhs, double alpha,
                      double* lhs, char trans) {
    ipx::SparseMatrix A = ToCsc(nrow, ncol, Ap, Ai, Ax);
    const bool t = trans == 't' || trans == 'T';
    Vector r = ToVector(rhs, t ? nrow : ncol);
    Vector l = ToVector(lhs, t ? ncol : nrow);
    ipx::MultiplyAdd(A, r, alpha, l, trans);
    FromVector(l, lhs);
}
void ref_transpose(Int nrow, Int ncol, const Int* Ap, const Int* Ai,
                   const double* Ax, Int* ATp, Int* ATi, double* ATx) {
    ipx::SparseMatrix A = ToCsc(nrow, ncol, Ap, Ai, Ax);
    ipx::SparseMatrix AT = ipx::Transpose(A);
    std::memcpy(ATp, AT.colptr(), sizeof(Int) * (nrow + 1));
    std::memcpy(ATi, AT.rowidx(), sizeof(Int) * AT.entries());
    std::memcpy(ATx, AT.values(), sizeof(double) * AT.entries());
}
// N = PermuteRows(CopyColumns(A, cols), perm), then ScaleColumn(N, k, scale[k])
// (the sequence of src/splitted_normal_matrix.cc:42-55).
void ref_copy_permute_scale(Int nrow, Int ncol, const Int* Ap, const Int* Ai,
                            const double* Ax, Int ncols_sel, const Int* cols,
                            const Int* perm, const double* scale, Int* Np,
                            Int* Ni, double* Nx) {
    ipx::SparseMatrix A = ToCsc(nrow, ncol, Ap, Ai, Ax);
    std::vector<Int> sel(cols, cols + ncols_sel);
    ipx::SparseMatrix N = ipx::CopyColumns(A, sel);
    if (perm) {
        std::vector<Int> p(perm, perm + nrow);
        ipx::PermuteRows(N, p);
    }
    if (scale)
        for (Int k = 0; k < ncols_sel; k++) ipx::ScaleColumn(N, k, scale[k]);
    std::memcpy(Np, N.colptr(), sizeof(Int) * (ncols_sel + 1));
    std::memcpy(Ni, N.rowidx(), sizeof(Int) * N.entries());
    std::memcpy(Nx, N.values(), sizeof(double) * N.entries());
}
void ref_inverse_perm(Int m, const Int* perm, Int* invperm) {
    std::vector<Int> p(perm, perm + m);
    std::vector<Int> q = ipx::InversePerm(p);
    std::memcpy(invperm, q.data(), sizeof(Int) * m);
}
double ref_dot(Int m, const double* x, const double* y) {
    return ipx::Dot(ToVector(x, m), ToVector(y, m));
}
double ref_infnorm(Int m, const double* x) {
    return ipx::Infnorm(ToVector(x, m));
}

// ---- basis-split operator from explicit factors ------------------------------
void* ref_split_new(Int m, const Int* Lp, const Int* Li, const double* Lx,
                    const Int* Up, const Int* Ui, const double* Ux, Int ncolN,
                    const Int* Np, const Int* Ni, const double* Nx, Int nfree,
                    const Int* free_positions) {
    auto* op = new SplitOperatorFromFactors;
    op->L = ToCsc(m, m, Lp, Li, Lx);
    op->U = ToCsc(m, m, Up, Ui, Ux);
    op->N = ToCsc(m, ncolN, Np, Ni, Nx);
    op->free_positions.assign(free_positions, free_positions + nfree);
    op->work.resize(m);
    return op;
}
void ref_split_free(void* op) {
    delete static_cast<SplitOperatorFromFactors*>(op);
}
void ref_split_apply(void* oph, Int m, const double* rhs, double* lhs,
                     double* dot) {
    auto* op = static_cast<SplitOperatorFromFactors*>(oph);
    Vector r = ToVector(rhs, m);
    Vector l(m);
    op->Apply(r, l, dot);
    FromVector(l, lhs);
}
// Plain CR (src/conjugate_residuals.cc:14-88) on the operator above.
Int ref_split_cr_solve(void* oph, Int m, const double* rhs, double tol,
                       const double* resscale, Int maxiter, double* lhs,
                       Int* iter, double* cdot_hist, Int hist_cap) {
    auto* op = static_cast<SplitOperatorFromFactors*>(oph);
    ipx::Control control;
    ipx::Parameters p;
    p.display = 0;
    control.parameters(p);
    RecordingOperator Crec(*op, cdot_hist, hist_cap);
    ipx::ConjugateResiduals cr(control);
    Vector r = ToVector(rhs, m);
    Vector l = ToVector(lhs, m);
    cr.Solve(Crec, r, tol, resscale, maxiter, l);
    FromVector(l, lhs);
    *iter = cr.iter();
    return cr.errflag();
}

// ---- Iterate (src/iterate.cc) --------------------------------------------------
void* ref_iterate_new(void* model_h) {
    return new ipx::Iterate(static_cast<RefModel*>(model_h)->model);
}
void ref_iterate_free(void* it) { delete static_cast<ipx::Iterate*>(it); }

void ref_iterate_initialize(void* ith, const double* x, const double* xl,
                            const double* xu, const double* y, const double* zl,
                            const double* zu) {
    ipx::Iterate* it = static_cast<ipx::Iterate*>(ith);
    const Int m = it->model().rows(), n = it->model().cols();
    it->Initialize(ToVector(x, n + m), ToVector(xl, n + m), ToVector(xu, n + m),
                   ToVector(y, m), ToVector(zl, n + m), ToVector(zu, n + m));
}

// Iterate::Update, :94-139.  NULL components are skipped as in the reference.
void ref_iterate_update(void* ith, double sp, const double* dx, const double* dxl,
                        const double* dxu, double sd, const double* dy,
                        const double* dzl, const double* dzu) {
    static_cast<ipx::Iterate*>(ith)->Update(sp, dx, dxl, dxu, sd, dy, dzl, dzu);
}

void ref_iterate_get(void* ith, double* x, double* xl, double* xu, double* y,
                     double* zl, double* zu) {
    ipx::Iterate* it = static_cast<ipx::Iterate*>(ith);
    FromVector(it->x(), x); FromVector(it->xl(), xl); FromVector(it->xu(), xu);
    FromVector(it->y(), y); FromVector(it->zl(), zl); FromVector(it->zu(), zu);
}

// state codes of include/ipx_kkt_hip.h: 0 fixed, 1 free, 2 barrier lb, 3 barrier ub, 4 boxed
void ref_iterate_states(void* ith, unsigned char* state) {
    ipx::Iterate* it = static_cast<ipx::Iterate*>(ith);
    const Int N = it->model().rows() + it->model().cols();
    for (Int j = 0; j < N; j++) {
        const bool l = it->has_barrier_lb(j), u = it->has_barrier_ub(j);
        if (l && u) state[j] = 4;
        else if (l) state[j] = 2;
        else if (u) state[j] = 3;
        else state[j] = it->StateOf(j) == ipx::Iterate::State::fixed ? 0 : 1;
    }
}

// ComputeResiduals, :536-588 (through the evaluating accessors); norms[0] = presidual, [1] = dresidual
void ref_iterate_residuals(void* ith, double* rb, double* rc, double* rl,
                           double* ru, double* norms) {
    ipx::Iterate* it = static_cast<ipx::Iterate*>(ith);
    FromVector(it->rb(), rb); FromVector(it->rc(), rc);
    FromVector(it->rl(), rl); FromVector(it->ru(), ru);
    norms[0] = it->presidual();
    norms[1] = it->dresidual();
}

// ComputeComplementarity, :642-670: complementarity, mu, mu_min, mu_max
void ref_iterate_complementarity(void* ith, double* out4) {
    ipx::Iterate* it = static_cast<ipx::Iterate*>(ith);
    out4[0] = it->complementarity();
    out4[1] = it->mu();
    out4[2] = it->mu_min();
    out4[3] = it->mu_max();
}

// ComputeObjectives, :590-640: pobjective, dobjective, and both after postprocessing (+ offset_)
void ref_iterate_objectives(void* ith, double* out4) {
    ipx::Iterate* it = static_cast<ipx::Iterate*>(ith);
    out4[0] = it->pobjective();
    out4[1] = it->dobjective();
    out4[2] = it->pobjective_after_postproc();
    out4[3] = it->dobjective_after_postproc();
}

// feasible(), optimal(), term_crit_reached() (:221-249) for the given tolerances, crossover_start = 0
void ref_iterate_termination(void* ith, double feasibility_tol, double optimality_tol, Int* out3) {
    ipx::Iterate* it = static_cast<ipx::Iterate*>(ith);
    it->feasibility_tol(feasibility_tol);
    it->optimality_tol(optimality_tol);
    it->crossover_start(0.0);
    out3[0] = it->feasible();
    out3[1] = it->optimal();
    out3[2] = it->term_crit_reached();
}


// ---- LU: the reference's LuFactorization / ForrestTomlin on factors computed elsewhere -----------------------
// The reference's LU kernel is BASICLU (absent).  Its OWN code around that kernel builds here: the wrapper
// LuFactorization::Factorize with the stability estimate (src/lu_factorization.cc:87-127) and ForrestTomlin,
// the LuUpdate the reference uses with lu_kernel = 1 (src/basis.cc:24-29; dense SolveDense / FtranForUpdate /
// BtranForUpdate / Update, src/forrest_tomlin.cc).  GivenFactors is an LuFactorization whose kernel hands back
// factors computed by the code under test (the oracle's or the device's LU), so that the reference's own
// objects judge them (stability) and run on them.
class GivenFactors : public ipx::LuFactorization {
public:
    ipx::SparseMatrix L, U;
    std::vector<Int> rowperm, colperm, dependent;
private:
    void _Factorize(Int, const Int*, const Int*, const Int*, const double*, double, bool,
                    ipx::SparseMatrix* Lout, ipx::SparseMatrix* Uout, std::vector<Int>* rp,
                    std::vector<Int>* cp, std::vector<Int>* dep) override {
        *Lout = L; *Uout = U; *rp = rowperm; *cp = colperm; *dep = dependent;
    }
};

struct RefLu {
    ipx::Control control;
    std::unique_ptr<ipx::ForrestTomlin> ft;
    Int dim = 0, flag = 0;
    GivenFactors* given = nullptr;      // owned by ft
};

void* ref_lu_new(Int dim, const Int* Bbegin, const Int* Bend, const Int* Bi, const double* Bx,
                 const Int* Lp, const Int* Li, const double* Lx, const Int* Up, const Int* Ui, const double* Ux,
                 const Int* rowperm, const Int* colperm, Int ndep, const Int* dependent) {
    RefLu* R = new RefLu;
    ipx::Parameters p;
    p.display = 0;
    R->control.parameters(p);
    R->dim = dim;
    GivenFactors* G = new GivenFactors;
    G->L = ToCsc(dim, dim, Lp, Li, Lx);
    G->U = ToCsc(dim, dim, Up, Ui, Ux);
    G->rowperm.assign(rowperm, rowperm + dim);
    G->colperm.assign(colperm, colperm + dim);
    G->dependent.assign(dependent, dependent + ndep);
    R->given = G;
    std::unique_ptr<ipx::LuFactorization> lu(G);
    R->ft.reset(new ipx::ForrestTomlin(R->control, dim, lu));
    R->flag = R->ft->Factorize(Bbegin, Bend, Bi, Bx, false);
    return R;
}
void ref_lu_free(void* h) { delete static_cast<RefLu*>(h); }
// out[3] = return flag of LuUpdate::Factorize (bit 0 unstable, bit 1 singular), stability(), fill_factor()
void ref_lu_info(void* h, double* out) {
    RefLu* R = static_cast<RefLu*>(h);
    out[0] = (double)R->flag;
    out[1] = R->given->stability();
    out[2] = R->ft->fill_factor();
}
void ref_lu_solve_dense(void* h, const double* rhs, double* lhs, Int trans) {
    RefLu* R = static_cast<RefLu*>(h);
    Vector r = ToVector(rhs, R->dim), l(R->dim);
    R->ft->SolveDense(r, l, trans ? 'T' : 'N');
    FromVector(l, lhs);
}
void ref_lu_ftran(void* h, Int nz, const Int* bi, const double* bx, double* lhs) {
    RefLu* R = static_cast<RefLu*>(h);
    ipx::IndexedVector v(R->dim);
    R->ft->FtranForUpdate(nz, bi, bx, v);
    for (Int i = 0; i < R->dim; i++) lhs[i] = v[i];
}
void ref_lu_btran(void* h, Int p, double* lhs) {
    RefLu* R = static_cast<RefLu*>(h);
    ipx::IndexedVector v(R->dim);
    R->ft->BtranForUpdate(p, v);
    for (Int i = 0; i < R->dim; i++) lhs[i] = v[i];
}
Int ref_lu_update(void* h, double pivot) { return static_cast<RefLu*>(h)->ft->Update(pivot); }
Int ref_lu_updates(void* h) { return static_cast<RefLu*>(h)->ft->updates(); }

}  // extern "C"
