/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's
 * KKT normal-equations hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (ipx_amd/) never does.
 *
 * Conventions: ipxint = int64_t indices, fp64 values, CSC matrices given as
 * (colptr[ncol+1], rowidx[nnz], values[nnz]).  Every function cites the
 * reference file:line whose arithmetic (including loop/summation order) it
 * restates; paths are relative to the reference root.
 */
#ifndef IPX_ORACLE_H_
#define IPX_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t orc_int;

/* error flags, values of include/ipx_status.h:31-47 */
#define ORC_ERROR_cr_iter_limit 201
#define ORC_ERROR_cr_matrix_not_posdef 202
#define ORC_ERROR_cr_precond_not_posdef 203
#define ORC_ERROR_cr_no_progress 204
#define ORC_ERROR_cr_inf_or_nan 205
#define ORC_ERROR_lapack_chol 401

/* variable statuses used by the basis path (src/basis.h BasicStatus) */
#define ORC_NONBASIC_FIXED -2
#define ORC_NONBASIC -1
#define ORC_BASIC 0
#define ORC_BASIC_FREE 1

/* lhs = F(rhs); *dot = rhs'lhs if dot != NULL (src/linear_operator.h:10-24) */
typedef void (*orc_apply_fn)(void* ctx, const double* rhs, double* lhs,
                             double* dot);

/* ---- vector kernels (src/utils.cc:32-45) -------------------------------- */
double orc_dot(orc_int m, const double* x, const double* y);
double orc_infnorm(orc_int m, const double* x);

/* ---- index / permutation arithmetic, bit-exact (row a15, a16) ----------- */
/* src/sparse_matrix.cc:120-151 */
/* Presolver::EquilibrateMatrix (src/presolver.cc:883-974); Ax is scaled in place */
orc_int orc_equilibrate(orc_int m, orc_int n, const orc_int* Ap, const orc_int* Ai, double* Ax, double* colscale,
                        double* rowscale);
void orc_transpose(orc_int nrow, orc_int ncol, const orc_int* Ap,
                   const orc_int* Ai, const double* Ax, orc_int* ATp,
                   orc_int* ATi, double* ATx);
/* src/utils.cc:73-80 */
void orc_inverse_perm(orc_int m, const orc_int* perm, orc_int* invperm);
/* CopyColumns + PermuteRows + ScaleColumn, src/sparse_matrix.cc:153-166,
 * src/splitted_normal_matrix.cc:42-55.  perm / scale may be NULL. */
void orc_copy_permute_scale(orc_int nrow, const orc_int* Ap, const orc_int* Ai,
                            const double* Ax, orc_int nsel, const orc_int* cols,
                            const orc_int* perm, const double* scale,
                            orc_int* Np, orc_int* Ni, double* Nx);
/* src/model.cc:34-56: returns num_dense_cols, *nz_dense = threshold
 * (nrow+1 if none).  colptr covers the ncol structural columns. */
orc_int orc_find_dense_columns(orc_int nrow, orc_int ncol, const orc_int* Ap,
                               orc_int* nz_dense);

/* ---- NormalMatrix (src/normal_matrix.cc:45-126, one-pass variant) ------- */
/* Ap/Ai/Ax: first n columns of AI (the slack identity is never read).
 * W has n+m entries or is NULL (W=1 on structurals, 0 on slacks). */
void orc_normal_apply(orc_int m, orc_int n, const orc_int* Ap,
                      const orc_int* Ai, const double* Ax, const double* W,
                      const double* rhs, double* lhs, double* dot);

/* ---- DiagonalPrecond (src/diagonal_precond.cc) --------------------------- */
typedef struct orc_diag_precond orc_diag_precond;
/* Factorize, :17-111.  Columns j<n with Ap[j+1]-Ap[j] >= nz_dense are "dense"
 * (src/model.h:52-55).  Returns NULL and *errflag = 401 if Cholesky fails. */
orc_diag_precond* orc_diag_factorize(orc_int m, orc_int n, const orc_int* Ap,
                                     const orc_int* Ai, const double* Ax,
                                     const double* W, orc_int nz_dense,
                                     orc_int precond_dense_cols,
                                     orc_int* errflag);
/* _Apply, :121-159 */
void orc_diag_apply(orc_diag_precond* P, const double* rhs, double* lhs,
                    double* dot);
orc_int orc_diag_num_dense(const orc_diag_precond* P);
/* copies diagonal_[m] and (if num_dense>0) chol_factor_[k*k]; NULL skips */
void orc_diag_get(const orc_diag_precond* P, double* diagonal, double* chol);
void orc_diag_free(orc_diag_precond* P);
/* dense Cholesky used in place of LAPACK dpotrf('L')/dpotrs('L'); column
 * major, lower triangle (src/lapack.cc:25-52).  Returns LAPACK-style info. */
orc_int orc_dpotrf_lower(orc_int k, double* a, orc_int lda);
void orc_dpotrs_lower(orc_int k, const double* a, orc_int lda, double* b);

/* ---- ConjugateResiduals (src/conjugate_residuals.cc) --------------------- */
/* Preconditioned CR, :90-213.  lhs holds the initial iterate on entry.
 * resnorm_hist (may be NULL) receives the termination-test residual norm of
 * every pass through the loop head (at most hist_cap entries).
 * Returns errflag; *iter = # iterations. */
orc_int orc_pcr_solve(orc_int m, orc_apply_fn C, void* Cctx, orc_apply_fn P,
                      void* Pctx, const double* rhs, double tol,
                      const double* resscale, orc_int maxiter, double* lhs,
                      orc_int* iter, double* resnorm_hist, orc_int hist_cap);
/* Plain CR, :14-88 */
orc_int orc_cr_solve(orc_int m, orc_apply_fn C, void* Cctx, const double* rhs,
                     double tol, const double* resscale, orc_int maxiter,
                     double* lhs, orc_int* iter, double* resnorm_hist,
                     orc_int hist_cap);

/* ---- KKTSolverDiag (src/kkt_solver_diag.cc) ------------------------------- */
typedef struct orc_kkt_diag orc_kkt_diag;
/* The matrix arrays must outlive the object (no copy, like the reference). */
orc_kkt_diag* orc_kkt_diag_new(orc_int m, orc_int n, const orc_int* Ap,
                               const orc_int* Ai, const double* Ax,
                               orc_int nz_dense, orc_int precond_dense_cols,
                               orc_int maxiter);
/* _Factorize, :18-65.  xl == NULL means Factorize(nullptr): W = 1. */
orc_int orc_kkt_diag_factorize(orc_kkt_diag* K, const double* xl,
                               const double* xu, const double* zl,
                               const double* zu, double mu);
/* _Solve, :82-118.  Returns errflag. */
orc_int orc_kkt_diag_solve(orc_kkt_diag* K, const double* a, const double* b,
                           double tol, double* x, double* y, orc_int* iter,
                           double* resnorm_hist, orc_int hist_cap);
void orc_kkt_diag_get(const orc_kkt_diag* K, double* W, double* resscale);
void orc_kkt_diag_free(orc_kkt_diag* K);

/* ---- sparse triangular solves (src/sparse_matrix.cc:224-311) ------------- */
orc_int orc_trisolve(orc_int dim, const orc_int* Ap, const orc_int* Ai,
                     const double* Ax, double* x, char trans, char uplo,
                     orc_int unitdiag);
void orc_forward_solve(orc_int dim, const orc_int* Lp, const orc_int* Li,
                       const double* Lx, const orc_int* Up, const orc_int* Ui,
                       const double* Ux, double* x);
void orc_backward_solve(orc_int dim, const orc_int* Lp, const orc_int* Li,
                        const double* Lx, const orc_int* Up, const orc_int* Ui,
                        const double* Ux, double* x);
/* src/sparse_matrix.cc:211-222; D may be NULL */
void orc_add_normal_product(orc_int nrow, orc_int ncol, const orc_int* Ap,
                            const orc_int* Ai, const double* Ax,
                            const double* D, const double* rhs, double* lhs);

/* ---- SplittedNormalMatrix + KKTSolverBasis::_Solve ------------------------ */
/* The LU factors are INPUTS (BASICLU is not part of the reference tree):
 * B[rowperm,colperm] = (L+I)*U, L strictly lower without diagonal, U upper with
 * the diagonal entry last in each column (src/lu_update.h:43-60).
 * basis[p] = variable at basis position p; status[j] for j<n+m is one of ORC_*.
 */
typedef struct orc_split orc_split;
/* Prepare, src/splitted_normal_matrix.cc:18-66 (copies everything) */
orc_split* orc_split_prepare(orc_int m, orc_int n, const orc_int* AIp,
                             const orc_int* AIi, const double* AIx,
                             const orc_int* Lp, const orc_int* Li,
                             const double* Lx, const orc_int* Up,
                             const orc_int* Ui, const double* Ux,
                             const orc_int* rowperm, const orc_int* colperm,
                             const orc_int* basis, const orc_int* status,
                             const double* colscale);
/* _Apply, :90-117 */
void orc_split_apply(orc_split* S, const double* rhs, double* lhs, double* dot);
/* exports of the prepared state for index-parity checks; NULL skips */
orc_int orc_split_get_sizes(const orc_split* S, orc_int* nnzN, orc_int* ncolN,
                            orc_int* nfree);
void orc_split_get(const orc_split* S, orc_int* Np, orc_int* Ni, double* Nx,
                   double* Ux_scaled, orc_int* rowperm_inv,
                   orc_int* free_positions);
/* Basis::SolveDense on the fresh factors (src/basis.cc:168-170 contract,
 * src/lu_update.h:62-65): trans 'N': B*lhs = rhs; 'T': B'*lhs = rhs. */
void orc_split_solve_dense(const orc_split* S, const double* rhs, double* lhs,
                           char trans);
/* KKTSolverBasis::_Solve, src/kkt_solver_basis.cc:75-194.  AI arrays are the
 * full m x (n+m) matrix [A I].  Returns errflag. */
orc_int orc_kkt_basis_solve(orc_split* S, const double* a, const double* b,
                            double tol, orc_int maxiter, double* x, double* y,
                            orc_int* iter, double* resnorm_hist,
                            orc_int hist_cap);
void orc_split_free(orc_split* S);

/* ---- IPM::SolveNewtonSystem (src/ipm.cc:532-645); pinned through the device code, see .cc ---- */
orc_int orc_newton_solve_diag(orc_kkt_diag* K, const double* rb, const double* rc,
    const double* rl, const double* ru, const double* sl, const double* su,
    const double* xl, const double* xu, const double* zl, const double* zu,
    const unsigned char* state, double tol, double* dx, double* dxl, double* dxu,
    double* dy, double* dzl, double* dzu, orc_int* iter);
orc_int orc_newton_solve_basis(orc_split* S, const double* rb, const double* rc,
    const double* rl, const double* ru, const double* sl, const double* su,
    const double* xl, const double* xu, const double* zl, const double* zu,
    const unsigned char* state, double tol, orc_int maxiter, double* dx,
    double* dxl, double* dxu, double* dy, double* dzl, double* dzu, orc_int* iter);

/* ---- Iterate (src/iterate.cc:94-139,536-588,642-670), StepToBoundary (src/ipm.cc:320-339) ---- */
void orc_iterate_update(orc_int m, orc_int n, const unsigned char* state, double* x,
    double* xl, double* xu, double* y, double* zl, double* zu, double sp,
    const double* dx, const double* dxl, const double* dxu, double sd,
    const double* dy, const double* dzl, const double* dzu);
void orc_iterate_residuals(orc_int m, orc_int n, const orc_int* Ap, const orc_int* Ai,
    const double* Ax, const unsigned char* state, const double* b, const double* c,
    const double* lb, const double* ub, const double* x, const double* xl,
    const double* xu, const double* y, const double* zl, const double* zu,
    double* rb, double* rc, double* rl, double* ru, double* norms);
void orc_iterate_complementarity(orc_int N, const unsigned char* state, const double* xl,
    const double* xu, const double* zl, const double* zu, double* out4);
double orc_step_to_boundary(orc_int len, const double* x, const double* dx, double alpha,
                            orc_int* blocking_index);

/* IPM::Predictor + AddCorrector + StepSizes + MakeStep (src/ipm.cc:340-530); pinned through the device code (tests/dropin/ipm_main.cc).
 * The iterate is updated in place; info[7] = step_primal, step_dual, mu_before, mu_after, sigma,
 * kktiter_predictor, kktiter_corrector.  Returns the errflag of the KKT solves. */
orc_int orc_ipm_step_diag(orc_kkt_diag* K, const unsigned char* state, const double* b,
    const double* c, const double* lb, const double* ub, double* x, double* xl, double* xu,
    double* y, double* zl, double* zu, double kkt_tol, double* info);

/* Iterate::ComputeObjectives (src/iterate.cc:590-640): out3 = pobjective, dobjective, offset */
void orc_iterate_objectives(orc_int m, orc_int n, const orc_int* Ap, const orc_int* Ai, const double* Ax,
    const unsigned char* state, const double* b, const double* c, const double* lb, const double* ub,
    const double* x, const double* y, const double* zl, const double* zu, double* out3);
/* Model::ComputeNorms (src/model.cc:58-67): out2 = norm_bounds, norm_c */
void orc_model_norms(orc_int m, orc_int n, const double* b, const double* c, const double* lb,
                     const double* ub, double* out2);
/* IPM::Driver (src/ipm.cc:56-123) around KKTSolverDiag; pinned through the device code (tests/dropin/ipm_main.cc).  Returns status_ipm;
 * info[10] = iter, errflag, kktiter, pobjective, dobjective (after postprocessing), presidual, dresidual,
 * complementarity, mu, last min(step_primal, step_dual). */
orc_int orc_ipm_driver_diag(orc_kkt_diag* K, const unsigned char* state, const double* b,
    const double* c, const double* lb, const double* ub, double* x, double* xl, double* xu, double* y,
    double* zl, double* zu, double kkt_tol, double feasibility_tol, double optimality_tol,
    orc_int ipm_maxiter, double* info);

/* ---- LU factorization behind src/lu_factorization.h:21-58 (section 8f rank 1) ------------------------------
 * singleton rounds + dense bump with partial pivoting; see ipx_oracle.cc.  Returns NULL when the bump exceeds
 * bump_limit rows (bump_limit < 0: no limit).  info[8] = column singletons, row singletons, bump size, rounds,
 * dependent columns, 0, 0, 0. */
typedef struct orc_lu orc_lu;
orc_lu* orc_lu_factorize(orc_int dim, const orc_int* Bbegin, const orc_int* Bend, const orc_int* Bi, const double* Bx,
                         double pivottol, int strict_abs_pivottol, orc_int bump_limit);
/* ... with elimination rounds in place of tearing (ipx_oracle.cc): when the singleton rounds stall with more than bump_limit
 * active columns, pivots of low Markowitz cost that form a diagonal block are eliminated together, round after round, until at
 * most sparse_min columns are active (or at most bump_limit and two rounds in a row have each eliminated fewer than 1 / slow_den of the columns; 0: never); NULL as well when the current matrix exceeds fill_max x nnz(B) + 2^20 entries (0: never).  info[6] = pivots of the elimination rounds, info[7] = # elimination rounds. */
orc_lu* orc_lu_factorize_sparse(orc_int dim, const orc_int* Bbegin, const orc_int* Bend, const orc_int* Bi, const double* Bx,
                                double pivottol, int strict_abs_pivottol, orc_int bump_limit, orc_int sparse_min, orc_int slow_den, orc_int fill_max);
/* the policy of round 5 (the device's default): rounds for bumps of more than sparse_from rows, ended by density / fill / slowness once
 * at most rest_limit columns are left (see ipx_oracle.cc) */
orc_lu* orc_lu_factorize_policy(orc_int dim, const orc_int* Bbegin, const orc_int* Bend, const orc_int* Bi, const double* Bx,
                                double pivottol, int strict_abs_pivottol, orc_int sparse_from, orc_int rest_limit, orc_int sparse_min,
                                orc_int slow_den, orc_int fill_max, double dense_at);
void orc_lu_sizes(const orc_lu* F, orc_int* lnz, orc_int* unz, orc_int* ndep, orc_int* info);
void orc_lu_get(const orc_lu* F, orc_int* Lp, orc_int* Li, double* Lx, orc_int* Up, orc_int* Ui, double* Ux,
                orc_int* rowperm, orc_int* colperm, orc_int* dependent);
void orc_lu_free(orc_lu* F);

/* ---- Maxvolume (section 8f rank 2): src/maxvolume.cc:108-337 over the part of ipx::Basis it drives
 * (src/basis.cc:162-330), with product-form updates of a fixed factorization; see ipx_oracle.cc.  status: the
 * ORC_* values above.  The matrix arrays must outlive the object. */
typedef struct orc_basis orc_basis;
orc_basis* orc_basis_new(orc_int m, orc_int n, const orc_int* Ap, const orc_int* Ai, const double* Ax,
                         const orc_int* basis, const orc_int* status, orc_int max_etas, orc_int* errflag);
void orc_basis_free(orc_basis* B);
/* counts[5] = factorizations, updates, ftrans, btrans, etas since the last factorization */
void orc_basis_get(const orc_basis* B, orc_int* basis, orc_int* status, orc_int* counts);
void orc_basis_solve_dense(const orc_basis* B, const double* rhs, double* lhs, char trans);
void orc_basis_solve_for_update(orc_basis* B, orc_int j, double* lhs);
void orc_basis_tableau_row(orc_basis* B, orc_int jb, double* btran, double* row, int ignore_fixed);
orc_int orc_basis_exchange_if_stable(orc_basis* B, orc_int jb, orc_int jn, double tableau_entry, orc_int* exchanged);
/* Maxvolume::RunSequential (src/maxvolume.cc:14-106); info[8] = updates, skipped, passes, volinc, refused, errflag,
 * tblnnz and tblmax of the last pass; log: accepted exchanges (jb, jn) */
orc_int orc_maxvolume_sequential(orc_basis* B, const double* colscale, double volume_tol, orc_int maxpasses, double* info,
                                 orc_int* log, orc_int log_cap);
orc_int orc_maxvolume_heuristic(orc_basis* B, const double* colscale, double volume_tol, orc_int maxskip_updates,
                                orc_int rows_per_slice, double* info, orc_int* log, orc_int log_cap);

#ifdef __cplusplus
}
#endif
#endif /* IPX_ORACLE_H_ */
