/* ipx_kkt_hip.h -- C ABI of the MI355X (gfx950) implementation of IPX's
 * per-IPM-iteration KKT normal-equations solve.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.
 * Each entry point names the reference interface (file:line under the reference
 * root) it replaces; ipx_amd/host/ holds KKTSolver / LinearOperator subclasses
 * written against the reference's own headers that call nothing but this file
 * (see INTEGRATION.md for the three lines of lp_solver.cc that select them).
 *
 * Conventions
 *  - ipxint = int64_t (include/ipx_config.h:5); values are IEEE fp64.
 *  - Sparse matrices are CSC triples (colptr[ncol+1], rowidx[nnz], values[nnz])
 *    exactly as ipx::SparseMatrix stores them (src/sparse_matrix.h:59-65).
 *  - Vector arguments are host pointers by default (the reference's Vector is a
 *    std::valarray in host memory).  ipxk_set_pointer_mode(IPXK_POINTER_DEVICE)
 *    switches all *vector* arguments of the solve/apply calls to device
 *    pointers (resident inputs, what bench.py times); matrix/factor/permutation
 *    arguments are always host pointers (they are uploaded once per
 *    model / per Factorize).
 *  - Every function returns 0 on success or a negative IPXK_E_* code; the text
 *    of the last failure is available from ipxk_last_error().  Numerical
 *    outcomes use the reference's own channel: an `errflag` output holding 0 or
 *    an IPX_ERROR_* value of include/ipx_status.h:31-47.
 *  - A context is not thread safe; calls block until results are in the output
 *    arrays (the reference interface is blocking and single-threaded).
 */
#ifndef IPX_KKT_HIP_H_
#define IPX_KKT_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t ipxint;
typedef struct ipxk_context ipxk_context;

#define IPXK_OK 0
#define IPXK_E_HIP (-1)        /* HIP runtime / RCCL failure  -> std::runtime_error */
#define IPXK_E_ALLOC (-2)      /* device or host allocation   -> std::bad_alloc     */
#define IPXK_E_ARGUMENT (-3)   /* invalid argument / state    -> std::logic_error   */
#define IPXK_E_UNSUPPORTED (-4)

#define IPXK_POINTER_HOST 0
#define IPXK_POINTER_DEVICE 1

/* basis statuses, values of ipx::Basis::BasicStatus (src/basis.h:64) */
#define IPXK_NONBASIC_FIXED (-2)
#define IPXK_NONBASIC (-1)
#define IPXK_BASIC 0
#define IPXK_BASIC_FREE 1

/* variable states, one byte per variable: Iterate::StateOf with the barrier state
 * split by Iterate::has_barrier_lb / has_barrier_ub (src/iterate.h:99-108,295-318) */
#define IPXK_STATE_FIXED 0
#define IPXK_STATE_FREE 1          /* FREE and the IMPLIED_* states */
#define IPXK_STATE_BARRIER_LB 2
#define IPXK_STATE_BARRIER_UB 3
#define IPXK_STATE_BARRIER_BOXED 4

/* Wall-clock seconds accumulated by the last solve, measured with HIP events on
 * the context's stream; they feed ipx_info::time_cr1* / time_cr2*
 * (include/ipx_info.h:66-75, src/kkt_solver_diag.cc:100-105,
 * src/kkt_solver_basis.cc:151-156). */
typedef struct ipxk_times {
    double cr;        /* whole CR loop            -> time_cr1 / time_cr2      */
    double op;        /* operator C applications  -> time_cr1_AAt / _NNt      */
    double precond;   /* preconditioner P         -> time_cr1_pre             */
    double solve_B;   /* forward solves           -> time_cr2_B               */
    double solve_Bt;  /* backward solves          -> time_cr2_Bt              */
} ipxk_times;

/* What the reference prints through control_.Debug(3) when a CR run stops on an
 * error (src/conjugate_residuals.cc:53-56,140-152,198-202): the numbers of the
 * last CR run on this context, for the host classes to format. */
typedef struct ipxk_cr_diag {
    ipxint errflag, iter, maxiter;
    double resnorm, tol;            /* 201: residual at the last loop head, tolerance */
    double cdot;                    /* 202: rhs-dot-lhs of the last operator application */
    double infnorm_residual;        /* 202 */
    double infnorm_sresidual;       /* 202 (preconditioned CR; 0 for plain CR) */
    double rps_old, rps_new;        /* 204: resnorm_precond_system before / after the 5 iterations */
} ipxk_cr_diag;

const char* ipxk_last_error(void);
int ipxk_device_count(void);

/* ---- model upload (stand-alone, host arrays, computed on `device`) ---- */
/* Presolver::EquilibrateMatrix (src/presolver.cc:883-974): recursive row / column
 * equilibration of the m x n structural matrix by powers of two.  Ax is scaled in
 * place; colscale[n], rowscale[m] receive the accumulated factors (1 where none).
 * *rounds = number of rounds that rescaled the matrix, or -1 if every entry was in
 * range from the start (the reference then leaves its scaling vectors empty,
 * src/presolver.cc:912-924).  All arithmetic is exact: bit-identical results. */
int ipxk_equilibrate(ipxint m, ipxint n, const ipxint* Ap, const ipxint* Ai,
                     double* Ax, double* colscale, double* rowscale,
                     ipxint* rounds, int device);
/* Transpose (src/sparse_matrix.cc:120-151): row-wise copy of an m x n CSC matrix,
 * entries of a row in ascending source-column order (Model::AIt, src/model.h:61). */
int ipxk_transpose(ipxint m, ipxint n, const ipxint* Ap, const ipxint* Ai,
                   const double* Ax, ipxint* ATp, ipxint* ATi, double* ATx,
                   int device);

/* ---- context: the model's matrix, resident on one GPU -------------------- */
/* Replaces the role of ipx::Model::AI()/AIt() for the path (src/model.h:61):
 * uploads the n structural columns of AI = [A I] (the slack identity is never
 * stored) as a 32-bit-index CSC and builds the row-wise copy on the device
 * side's own index arithmetic (Transpose, src/sparse_matrix.cc:120-151 --
 * ascending source-column order within each row).  Also classifies dense
 * columns as Model::FindDenseColumns does (src/model.cc:34-56). */
int ipxk_create(ipxint m, ipxint n, const ipxint* Ap, const ipxint* Ai,
                const double* Ax, int device, ipxk_context** out);
void ipxk_destroy(ipxk_context* ctx);
int ipxk_set_pointer_mode(ipxk_context* ctx, int mode);
/* Use a caller-owned hipStream_t (e.g. torch's current stream); NULL restores
 * the context's own stream. */
int ipxk_set_stream(ipxk_context* ctx, void* hip_stream);
int ipxk_synchronize(ipxk_context* ctx);
/* on != 0: HIP events around every operator / preconditioner / triangular-solve application fill
 * ipxk_times::op, precond, solve_B, solve_Bt (the reference's per-object timers,
 * src/normal_matrix.cc:57,125, src/diagonal_precond.cc:126,158,
 * src/splitted_normal_matrix.cc:93-110).  Off by default: ~2 extra stream markers per call. */
int ipxk_set_profiling(ipxk_context* ctx, int on);
/* Control::InterruptCheck for the calls that take no callback of their own: ipxk_maxvolume and
 * ipxk_maxvolume_sequential poll it once per candidate column (src/maxvolume.cc:52,250); a nonzero
 * value ends the run with info->errflag = that value, the exchanges made so far kept.  NULL: none. */
int ipxk_set_interrupt(ipxk_context* ctx, ipxint (*interrupt)(void* user), void* interrupt_user);
/* A context handed from one solver object to the next (ipx_amd/host/hip_device.h keeps one per Model, as the
 * reference's solver objects share one Model, src/lp_solver.cc:375,386,457) starts like a new one: NormalMatrix /
 * DiagonalPrecond / KKTSolverDiag unprepared, no iterate, no operator of a basis, no LU factors, no interrupt
 * callback, host pointer mode, own stream; the pivot tolerance of Maxvolume's refactorizations -- tightened for good
 * after an unstable exchange, like Basis::TightenLuPivotTol (src/basis.cc:490-503) -- is set to lu_pivottol
 * (Control::lu_pivottol(), src/basis.cc:30; <= 0: 0.1).  The model and its layouts stay; workspaces stay allocated. */
int ipxk_reset_solver_state(ipxk_context* ctx, double lu_pivottol);
ipxint ipxk_num_dense_cols(const ipxk_context* ctx);
/* The locality-recovering renumbering of the model ("row/column reordering ... must stay a pure permutation").
 * ipxk_create looks for one -- breadth-first levels of the bipartite graph rows <-> columns from a pseudo-peripheral row, rows
 * and columns numbered by (level, index) -- builds a second copy of the matrix in that numbering with gather layouts of its own,
 * times NormalMatrix::Apply's two products on both copies and keeps the copy if it is at least 10 % faster.  Then the CR loop of
 * ipxk_kkt_diag_solve (src/kkt_solver_diag.cc:98-99) runs on the copy: right-hand side, weights, preconditioner and residual
 * scaling are permuted going in, y coming out; nothing else of the ABI sees the numbering.  Models with dense columns (the
 * Sherman-Morrison-Woodbury preconditioner), partitioned contexts and models of fewer than 2^20 entries keep the numbering as given;
 * a matrix without structure (half of its rows within 8 levels of any row) is recognised at once.  IPXK_REORDER=0: never, =1: always. */
typedef struct {
  ipxint active;             /* 1: the renumbered copy is in use */
  ipxint levels, components; /* of the breadth-first structure */
  double ms;                 /* cost inside ipxk_create */
  double us_original, us_reordered;   /* the two products of one Apply on either copy (0: not timed) */
} ipxk_reorder_info;
int ipxk_get_reorder_info(const ipxk_context* ctx, ipxk_reorder_info* info);
/* rowperm[m], colperm[n]: new index -> index as given (either may be NULL); fails if no renumbering was computed */
int ipxk_get_reordering(ipxk_context* ctx, ipxint* rowperm, ipxint* colperm);
/* Copies out the device-side row-wise matrix (for bit-exact index parity
 * tests against Transpose): ATp[m+1], ATi[nnz], ATx[nnz]; NULL skips. */
int ipxk_get_rowwise(const ipxk_context* ctx, ipxint* ATp, ipxint* ATi,
                     double* ATx);

/* ---- NormalMatrix (src/normal_matrix.h:18-43) ---------------------------- */
/* Prepare(W): W has n+m entries or is NULL (1 on structurals, 0 on slacks).
 * Unlike the reference (raw pointer kept, normal_matrix.h:24-28) W is copied
 * to the device here; in device pointer mode it is used in place. */
int ipxk_normal_prepare(ipxk_context* ctx, const double* W);
/* _Apply (src/normal_matrix.cc:45-126): lhs = AI*W*AI'*rhs, optional dot. */
int ipxk_normal_apply(ipxk_context* ctx, const double* rhs, double* lhs,
                      double* rhs_dot_lhs);

/* ---- DiagonalPrecond (src/diagonal_precond.h:25-57) ---------------------- */
/* Factorize (src/diagonal_precond.cc:17-111); *errflag = 0 or
 * IPX_ERROR_lapack_chol (401). */
int ipxk_diag_factorize(ipxk_context* ctx, const double* W,
                        int precond_dense_cols, ipxint* errflag);
/* _Apply (src/diagonal_precond.cc:121-159) */
int ipxk_diag_apply(ipxk_context* ctx, const double* rhs, double* lhs,
                    double* rhs_dot_lhs);
/* copies diagonal_[m] and the k x k column-major Cholesky factor; NULL skips */
int ipxk_diag_get(const ipxk_context* ctx, double* diagonal, double* chol);

/* ---- ConjugateResiduals (src/conjugate_residuals.h:20-70) ---------------- */
/* Preconditioned CR (src/conjugate_residuals.cc:90-213) with C = the prepared
 * NormalMatrix and P = the factorized DiagonalPrecond.  lhs: initial iterate in,
 * solution out.  resscale may be NULL.  maxiter < 0 means m+100.
 * interrupt (may be NULL) plays Control::InterruptCheck()
 * (src/control.cc:17-22): a nonzero return value stops the solve and becomes
 * *errflag.  Granularity: the reference polls after every iteration
 * (src/conjugate_residuals.cc:209); here the callback is polled once per cycle
 * of 5 iterations that the host enqueues, first after the first cycle (a
 * system whose initial residual meets tol returns errflag 0 whatever the
 * callback says, like the reference), and the host runs up to 2 cycles ahead
 * of the device: *iter is the last iteration that finished, up to 15
 * iterations after the callback first returned nonzero.  On a partitioned
 * system the flag is reduced (max) over the ranks together with the loop's
 * `done` snapshot, so every rank leaves the loop in the same cycle.
 * resnorm_hist (may be NULL): termination-test residual norm of each pass
 * through the loop head, at most hist_cap entries (always a host pointer). */
typedef ipxint (*ipxk_interrupt_fn)(void* user);
int ipxk_pcr_solve(ipxk_context* ctx, const double* rhs, double tol,
                   const double* resscale, ipxint maxiter, double* lhs,
                   ipxint* iter, ipxint* errflag, ipxk_interrupt_fn interrupt,
                   void* interrupt_user, double* resnorm_hist, ipxint hist_cap,
                   ipxk_times* times);

/* numbers of the last CR run (any of ipxk_pcr_solve, ipxk_cr_solve, the KKT solves) */
int ipxk_cr_diagnostics(ipxk_context* ctx, ipxk_cr_diag* out);

/* ---- KKTSolverDiag (src/kkt_solver_diag.h:23-49) ------------------------- */
/* _Factorize (src/kkt_solver_diag.cc:18-65).  xl == NULL is Factorize(nullptr)
 * (W = 1).  mu = iterate->mu().  Builds W, resscale, prepares the normal matrix
 * and factorizes the preconditioner. */
int ipxk_kkt_diag_factorize(ipxk_context* ctx, const double* xl,
                            const double* xu, const double* zl,
                            const double* zu, double mu,
                            int precond_dense_cols, ipxint* errflag);
/* _Solve (src/kkt_solver_diag.cc:82-118): a[n+m], b[m] -> x[n+m], y[m]. */
int ipxk_kkt_diag_solve(ipxk_context* ctx, const double* a, const double* b,
                        double tol, ipxint maxiter, double* x, double* y,
                        ipxint* iter, ipxint* errflag,
                        ipxk_interrupt_fn interrupt, void* interrupt_user,
                        ipxk_times* times);
/* copies W_[n+m] and resscale_[m] (host pointers; NULL skips) */
int ipxk_kkt_diag_get(const ipxk_context* ctx, double* W, double* resscale);

/* ---- SplittedNormalMatrix (src/splitted_normal_matrix.h:25-67) ----------- */
/* Prepare (src/splitted_normal_matrix.cc:18-66).  The hand-off from
 * Basis::GetLuFactors (src/basis.cc:162-166) is explicit: L (strictly lower,
 * no diagonal), U (upper, diagonal last in each column), rowperm, colperm with
 * B[rowperm,colperm] = (L+I)*U (src/lu_update.h:43-60); basis[p] = variable at
 * position p; status[n+m] in IPXK_*; colscale[n+m].  Row indices inside a column
 * may come in any order as long as U's diagonal is last; a dense trailing block of
 * the factors is cut out of the sweeps (and applied as a blocked solve or an
 * explicit inverse) only when the columns of U that cross it are sorted, as
 * GetLuFactors returns them. */
int ipxk_split_prepare(ipxk_context* ctx, const ipxint* Lp, const ipxint* Li,
                       const double* Lx, const ipxint* Up, const ipxint* Ui,
                       const double* Ux, const ipxint* rowperm,
                       const ipxint* colperm, const ipxint* basis,
                       const ipxint* status, const double* colscale);
/* Prepare when ONLY the scaling factors changed: KKTSolverBasis::_Factorize
 * keeps the factorization when the basis was not updated
 * (src/kkt_solver_basis.cc:59-64), and then all that differs for the operator is
 * the scaling of U's columns and of N and the free positions
 * (src/splitted_normal_matrix.cc:30-64).  Reuses the level schedule and the
 * packed factors of the last ipxk_split_prepare (same L, U, permutations and
 * basis) and rebuilds only what depends on status / colscale; the result is
 * bit-identical to a full ipxk_split_prepare with the same arguments. */
int ipxk_split_rescale(ipxk_context* ctx, const ipxint* status,
                       const double* colscale);
/* ---- basis LU factorization on the device ------------------
 * The LuFactorization contract, src/lu_factorization.h:21-58:
 *     B[rowperm,colperm] = (L+I)*U
 * with L strictly lower (no diagonal stored), U upper with the diagonal last in
 * each column, indices sorted; dependent columns are replaced by unit columns in
 * the product and their positions in colperm are listed.  Replaces the kernel
 * behind BasicLuKernel::_Factorize (src/basiclu_kernel.cc:31-82, BASICLU); call
 * sites ForrestTomlin::_Factorize (src/forrest_tomlin.cc:28-30) and, through
 * LuUpdate::Factorize, Basis::Factorize (src/basis.cc:116-156).  Method: rounds of
 * column / row singletons, then the remaining bump as a dense matrix with partial
 * pivoting (any pivottol in (0,1] is therefore met inside the bump; a row
 * singleton must pass |a| >= pivottol * max|active column|).  Absolute pivot
 * tolerance: kLuDependencyTol = 1e-3 (src/ipx_internal.h:26) if
 * strict_abs_pivottol, else 1e-14.  A bump of more than IPXK_LU_BUMP_MAX rows
 * (environment, default 8192) is torn first: whenever the rounds stall, the
 * active columns with the most active entries are set aside as spikes and the
 * rounds go on; the spikes are carried through the row singleton pivots by a
 * forward substitution and end in a dense block of one row per spike
 * (bump-and-spike ordering; ipxk_lu_info.spikes; the spikes may number up to
 * 16384, the largest dense block the panel kernels take).  If that block would
 * exceed 16384 rows, the factorization starts again and eliminates the bump sparsely:
 * rounds of pivots of low Markowitz cost that form a diagonal block, under the
 * same absolute and relative pivot thresholds, the fill-in entering the current
 * matrix (ipxk_lu_info.sparse_pivots / sparse_rounds; environment IPXK_LU_SPARSE
 * = 1: from the start, = 0: never); what is left is factorized densely.
 * IPXK_E_UNSUPPORTED is returned only if that rest still exceeds the limit or
 * the elimination fills the bump beyond IPXK_LU_SPARSE_FILL_MAX (8) x nnz(B).
 * Columns of B are Bi/Bx[Bbegin[j] .. Bend[j]-1] (4-array form, as Basis passes
 * AI's arrays); indices need not be sorted.  The factors stay on the device;
 * ipxk_lu_get_factors copies them out (array sizes from ipxk_lu_info; any
 * pointer may be NULL). */
typedef struct {
  ipxint lnz, unz;          /* entries of L (no diagonal) and of U (with it) */
  ipxint num_dependent;     /* columns replaced by unit columns */
  ipxint col_singletons, row_singletons, bump, rounds;
  double seconds_singletons, seconds_bump, seconds_assemble;
  ipxint spikes;            /* columns torn off a bump beyond the dense limit (0: the bump
                               was factorized as it stood); then bump == spikes */
  ipxint sparse_pivots, sparse_rounds; /* pivots and rounds of the sparse elimination of a bump
                               beyond the dense limit (0: none) */
  ipxint reused;            /* 1: nothing was computed -- the context already held the factors of exactly
                               this matrix (the basis ipxk_lu_factorize_basis / ipxk_maxvolume factorized
                               last, columns in any order, same tolerances: Basis::Load after Maxvolume
                               on the device, src/basis.cc:81-114) and hands them out again */
} ipxk_lu_info;
int ipxk_lu_factorize(ipxk_context* ctx, ipxint dim, const ipxint* Bbegin,
                      const ipxint* Bend, const ipxint* Bi, const double* Bx,
                      double pivottol, int strict_abs_pivottol,
                      ipxk_lu_info* info);
/* Number of LU factorizations this context has COMPUTED so far (a reused one does not count): a caller that
 * keeps track of which basis the resident factors belong to (KKTSolverBasisHip) sees with it whether
 * another factorization went through the context in between. */
ipxint ipxk_lu_generation(const ipxk_context* ctx);
int ipxk_lu_get_factors(ipxk_context* ctx, ipxint* Lp, ipxint* Li, double* Lx,
                        ipxint* Up, ipxint* Ui, double* Ux, ipxint* rowperm,
                        ipxint* colperm, ipxint* dependent_cols);
/* B = AI[:, basis[0..m-1]] taken from the matrix resident in the context (slack
 * columns j >= n are unit columns): Basis::Factorize (src/basis.cc:116-156)
 * without B crossing PCIe -- only the m basis indices do. */
int ipxk_lu_factorize_basis(ipxk_context* ctx, const ipxint* basis,
                            double pivottol, int strict_abs_pivottol,
                            ipxk_lu_info* info);
/* SplittedNormalMatrix::Prepare (src/splitted_normal_matrix.cc:18-66) on the
 * factors of the last ipxk_lu_factorize_basis, which never leave the device:
 * the GetLuFactors hand-off of src/basis.cc:162-166 without a host round trip.
 * Fails (IPXK_E_ARGUMENT) when that factorization found dependent columns: the
 * reference repairs the basis first (Basis::AdaptToSingularFactorization). */
int ipxk_split_prepare_lu(ipxk_context* ctx, const ipxint* status,
                          const double* colscale);
/* ---- Maxvolume on the device -------------------------------
 * Maxvolume::RunHeuristic (src/maxvolume.cc:108-153 with Driver :202-320,
 * ScaleFtran :322-337, FindLargest :179-200) over the part of ipx::Basis it
 * drives (SolveDense, SolveForUpdate, TableauRow, ExchangeIfStable,
 * src/basis.cc:162-330), followed by the tail of KKTSolverBasis::_Factorize
 * (src/kkt_solver_basis.cc:46-61): the split operator of the final basis -- from
 * a fresh factorization, or from the earlier factors with the last exchanges
 * behind them (kept_etas below).  Precondition: ipxk_lu_factorize_basis +
 * ipxk_split_prepare_lu for the current basis.  status / colscale: n+m entries
 * as for ipxk_split_prepare (BASIC_FREE variables never leave, NONBASIC_FIXED
 * ones never enter).  The factorization is kept current by product-form etas on
 * the resident factors and refactorized after max_etas exchanges or when an
 * exchange fails the stability test (the pivot from the tableau row against the
 * one from the tableau column, relative 1e-8).  On return basis_out[m] /
 * status_out[n+m] hold the new basis (either may be NULL), the context the
 * operator for it (by a fresh factorization, src/kkt_solver_basis.cc:56-61, or -- when that would cost more than
 * carrying the last exchanges through the solves that follow, a fixed cost model -- by the earlier factors with the
 * exchanges as a product form behind them: info.kept_etas; IPXK_MAXVOL_KEEP_ETAS=0 in the environment: always fresh), exchange_log (may be NULL) the accepted exchanges as pairs
 * (leaving variable, entering variable), at most log_cap of them.
 * info.errflag: 0, or IPX_ERROR_basis_too_ill_conditioned (306). */
typedef struct {
  double volume_tol;        /* ipx_parameters.volume_tol, default 2.0 */
  ipxint maxskip_updates;   /* default 10 */
  ipxint rows_per_slice;    /* default 10000 */
  ipxint max_etas;          /* exchanges between refactorizations, default (0) 100; < 0: as many as pay -- a
                               refactorization when the etas since the last one have cost as much as it
                               costs (a fixed cost model, not the clock: the run stays reproducible),
                               at least 100, at most 1024 or what 2 GiB of dense etas hold */
} ipxk_maxvolume_params;
typedef struct {
  ipxint updates, skipped, slices;   /* Maxvolume::updates() / skipped() / slices() */
  ipxint refused;                    /* exchanges refused as unstable (then refactorized) */
  ipxint factorizations;             /* refactorizations, the final one (if any: kept_etas) included */
  ipxint errflag;
  double volinc;                     /* Maxvolume::volinc(): log2 of the volume gained */
  double seconds;
  ipxint kept_etas;                  /* > 0: no final refactorization -- the context holds the factors of an earlier basis and
                                        this many etas (product form) behind them, which every solve and operator application
                                        of the context applies, and the next ipxk_maxvolume goes on with; any factorization
                                        in the context (ipxk_lu_factorize*, ipxk_split_prepare*) ends that state */
} ipxk_maxvolume_info;
int ipxk_maxvolume(ipxk_context* ctx, const ipxint* status, const double* colscale,
                   const ipxk_maxvolume_params* params, ipxint* basis_out,
                   ipxint* status_out, ipxk_maxvolume_info* info,
                   ipxint* exchange_log, ipxint log_cap);
/* Maxvolume::RunSequential (src/maxvolume.cc:14-106), the variant KKTSolverBasis
 * selects for update_heuristic == 0 (src/kkt_solver_basis.cc:47-51): passes over
 * the NONBASIC columns in decreasing order of their scaling factor; per candidate
 * the tableau column (SolveForUpdate) and the largest scaled entry
 * |x_p| * invscale_basic[p] * colscale[j]; an exchange when it exceeds
 * max(volume_tol, 1), checked like ipxk_maxvolume's (the BTRAN of the leaving
 * variable that ExchangeIfStable computes for sys = -1 gives the pivot from the
 * row).  maxpasses < 0: until a pass brings no update (the reference's default).
 * Same preconditions, outputs and final refactorization + Prepare as
 * ipxk_maxvolume; info->slices reports the number of passes.  A candidate costs
 * two sweeps over all m unknowns on the device (the reference's solves are
 * hypersparse): meant for moderate sizes, the heuristic is the one for 1M rows. */
int ipxk_maxvolume_sequential(ipxk_context* ctx, const ipxint* status,
                              const double* colscale, double volume_tol,
                              ipxint maxpasses, ipxint max_etas, ipxint* basis_out,
                              ipxint* status_out, ipxk_maxvolume_info* info,
                              ipxint* exchange_log, ipxint log_cap);
/* _Apply (src/splitted_normal_matrix.cc:90-117) */
int ipxk_split_apply(ipxk_context* ctx, const double* rhs, double* lhs,
                     double* rhs_dot_lhs);
/* ForwardSolve / BackwardSolve with the prepared (scaled) factors, in place
 * (src/sparse_matrix.cc:303-311). */
int ipxk_forward_solve(ipxk_context* ctx, double* x);
int ipxk_backward_solve(ipxk_context* ctx, double* x);
/* Basis::SolveDense on the fresh unscaled factors (src/basis.cc:168-170) */
int ipxk_solve_dense(ipxk_context* ctx, const double* rhs, double* lhs,
                     char trans);
/* number of level sets of the four triangular sweeps (U', L', L, U) */
int ipxk_split_levels(const ipxk_context* ctx, ipxint levels[4]);
/* Plain CR (src/conjugate_residuals.cc:14-88) on the prepared split operator */
int ipxk_cr_solve(ipxk_context* ctx, const double* rhs, double tol,
                  const double* resscale, ipxint maxiter, double* lhs,
                  ipxint* iter, ipxint* errflag, ipxk_interrupt_fn interrupt,
                  void* interrupt_user, double* resnorm_hist, ipxint hist_cap,
                  ipxk_times* times);

/* ---- KKTSolverBasis::_Solve (src/kkt_solver_basis.cc:75-194) ------------- */
int ipxk_kkt_basis_solve(ipxk_context* ctx, const double* a, const double* b,
                         double tol, ipxint maxiter, double* x, double* y,
                         ipxint* iter, ipxint* errflag,
                         ipxk_interrupt_fn interrupt, void* interrupt_user,
                         ipxk_times* times);

/* ---- IPM::SolveNewtonSystem (src/ipm.cc:532-645), SURVEY.md section 8f row 3 ----
 * Builds the KKT right-hand side from the residuals rb[m], rc/rl/ru[n+m] (any of
 * these four may be NULL = zero) and the complementarity targets sl/su[n+m],
 * solves with the factorized diag solver (use_basis = 0) or the prepared basis
 * solver (use_basis = 1) to tol, and recovers the Newton step dx, dxl, dxu, dzl,
 * dzu [n+m], dy [m].  xl, xu, zl, zu are the iterate's vectors, state[n+m] the
 * IPXK_STATE_* codes.  With device pointers nothing crosses PCIe.  On errflag != 0
 * the step vectors are undefined (ipm.cc:571-573 returns). */
int ipxk_newton_solve(ipxk_context* ctx, int use_basis, const double* rb,
                      const double* rc, const double* rl, const double* ru,
                      const double* sl, const double* su, const double* xl,
                      const double* xu, const double* zl, const double* zu,
                      const unsigned char* state, double tol, ipxint maxiter,
                      double* dx, double* dxl, double* dxu, double* dy,
                      double* dzl, double* dzu, ipxint* iter, ipxint* errflag,
                      ipxk_interrupt_fn interrupt, void* interrupt_user,
                      ipxk_times* times);

/* ---- the IPM iterate on the device -----------
 * Iterate::Initialize / accessors (src/iterate.cc:60-92): the six vectors
 * x, xl, xu, zl, zu [n+m], y [m] and one IPXK_STATE_* byte per variable are
 * copied into the context. */
int ipxk_iterate_set(ipxk_context* ctx, const double* x, const double* xl,
                     const double* xu, const double* y, const double* zl,
                     const double* zu, const unsigned char* state);
/* copies out; NULL pointers are skipped */
int ipxk_iterate_get(ipxk_context* ctx, double* x, double* xl, double* xu,
                     double* y, double* zl, double* zu);
/* Iterate::Update (src/iterate.cc:94-139): x += sp*dx on non-fixed variables,
 * xl/xu += sp*d, zl/zu += sd*d on variables with that barrier term, truncated
 * at kBarrierMin = 1e-30; y += sd*dy.  NULL step components are skipped. */
int ipxk_iterate_update(ipxk_context* ctx, double sp, const double* dx,
                        const double* dxl, const double* dxu, double sd,
                        const double* dy, const double* dzl, const double* dzu);
/* Iterate::ComputeResiduals (src/iterate.cc:536-588) for the model vectors
 * b[m], c, lb, ub [n+m]: rb = b - AI x, rc = c - AI'y - zl + zu (0 on fixed
 * variables), rl = lb - x + xl, ru = ub - x - xu (0 without that barrier
 * term); presidual = max(|rb|,|rl|,|ru|)_inf, dresidual = |rc|_inf. */
int ipxk_iterate_residuals(ipxk_context* ctx, const double* b, const double* c,
                           const double* lb, const double* ub, double* rb,
                           double* rc, double* rl, double* ru,
                           double* presidual, double* dresidual);
/* Iterate::ComputeComplementarity (src/iterate.cc:642-670):
 * out4 = {complementarity, mu, mu_min, mu_max} (host memory) */
int ipxk_iterate_complementarity(ipxk_context* ctx, double out4[4]);
/* StepToBoundary (src/ipm.cc:320-339): largest alpha <= alpha0 with
 * x + alpha*dx >= 0, damped by 1 - eps at the blocking index (-1: none).
 * Parallel minimum over the candidates; equals the reference's sequential
 * scan except when alpha lands within one ulp factor of another candidate. */
int ipxk_step_to_boundary(ipxk_context* ctx, const double* x, const double* dx,
                          ipxint len, double alpha0, double* alpha,
                          ipxint* blocking_index);

/* One predictor-corrector step on the resident iterate: IPM::Predictor,
 * AddCorrector, StepSizes and MakeStep (src/ipm.cc:340-530) with the factorized
 * diag solver (use_basis = 0) or the prepared basis solver (1); KKT tolerance
 * kkt_tol*sqrt(mu) (src/ipm.cc:572, ipx_parameters::kkt_tol = 0.3).  The caller
 * factorizes for the current iterate first, as IPM::Driver does
 * (ipxk_iterate_factorize_diag for the diag solver).  b[m], c, lb, ub [n+m]
 * are the model's vectors.  On errflag != 0 (from a KKT solve) the iterate is
 * left unchanged. */
typedef struct ipxk_ipm_step_info {
    double step_primal, step_dual;   /* IPM::step_primal_, step_dual_ */
    double mu_before, mu_after;      /* Iterate::mu() before / after the update */
    double sigma;                    /* centering parameter of the corrector */
    double presidual, dresidual;     /* of the iterate the step started from */
    ipxint kktiter_predictor, kktiter_corrector;
    ipxint errflag;
} ipxk_ipm_step_info;
int ipxk_ipm_step(ipxk_context* ctx, int use_basis, const double* b,
                  const double* c, const double* lb, const double* ub,
                  double kkt_tol, ipxint maxiter, ipxk_ipm_step_info* info,
                  ipxk_interrupt_fn interrupt, void* interrupt_user);
/* KKTSolverDiag::_Factorize (src/kkt_solver_diag.cc:18-65) for the resident
 * iterate: nothing crosses PCIe. */
int ipxk_iterate_factorize_diag(ipxk_context* ctx, int precond_dense_cols,
                                ipxint* errflag);

/* Iterate::ComputeObjectives (src/iterate.cc:590-640) of the resident iterate for
 * the model vectors: out3 = {pobjective, dobjective, offset}; pobjective + offset
 * and dobjective + offset are the objectives after postprocessing (:203-211).
 * Variable states as in ipxk_iterate_set (the implied states of the basis solver's
 * drop procedures do not exist on the device). */
int ipxk_iterate_objectives(ipxk_context* ctx, const double* b, const double* c,
                            const double* lb, const double* ub, double out3[3]);

/* IPM::Driver (src/ipm.cc:56-123) on the resident iterate with the diag solver:
 * loop of {termination test (Iterate::term_crit_reached, src/iterate.cc:221-249,
 * with crossover_start = 0), divergence / bad-iteration test (:71-93), iteration
 * limit, InterruptCheck, KKTSolverDiag::Factorize, Predictor + AddCorrector +
 * MakeStep}.  status_ipm uses the values of include/ipx_status.h
 * (IPX_STATUS_optimal 1, primal_infeas 3, dual_infeas 4, time_limit 5,
 * iter_limit 6, no_progress 7, failed 8).  A CR failure of the diag solver ends
 * the loop with IPX_STATUS_failed and the CR errflag: that is where LpSolver
 * switches to the basis solver (src/lp_solver.cc:399-418), the caller's move. */
typedef struct ipxk_ipm_params {
    double kkt_tol;            /* ipx_parameters::kkt_tol, 0.3 */
    double feasibility_tol;    /* ipm_feasibility_tol, 1e-6 */
    double optimality_tol;     /* ipm_optimality_tol, 1e-8 */
    ipxint kkt_maxiter;        /* CR iteration cap of the diag solver (src/lp_solver.cc:393) */
    ipxint ipm_maxiter;        /* ipm_maxiter, 300 */
    int precond_dense_cols;
} ipxk_ipm_params;
typedef struct ipxk_ipm_info {
    ipxint status_ipm, iter, errflag, kktiter;
    double pobjective, dobjective;       /* after postprocessing */
    double presidual, dresidual, complementarity, mu;
    double step_primal, step_dual;       /* of the last step */
    ipxint basis_updates;                /* ipxk_ipm_driver_basis: exchanges by Maxvolume (Info::updates_ipm) */
} ipxk_ipm_info;
int ipxk_ipm_driver(ipxk_context* ctx, const double* b, const double* c,
                    const double* lb, const double* ub,
                    const ipxk_ipm_params* params, ipxk_ipm_info* info,
                    ipxk_interrupt_fn interrupt, void* interrupt_user);

/* The main IPM phase (LpSolver::RunMainIPM, src/lp_solver.cc:456-462): IPM::Driver around KKTSolverBasis.
 * Every iteration's KKTSolverBasis::_Factorize (src/kkt_solver_basis.cc:20-63) runs on the device: scaling
 * factors from the resident iterate (Iterate::ScalingFactor, src/iterate.cc:183-198), Maxvolume
 * (ipxk_maxvolume), fresh factorization, Prepare; then the predictor-corrector step with the basis-
 * preconditioned solves.  The starting basis is the slack basis (ConstructBasisFromWeights with
 * crash_basis = 0, src/basis.cc:353-385); DropPrimal / DropDual (:36-43) are not taken.  Models whose
 * iterate holds free or fixed variables are refused (IPXK_E_UNSUPPORTED).  basis_out[m] / status_out[n+m]
 * (either may be NULL) return the final basis.  params->kkt_maxiter is ignored (KKTSolverBasis runs CR with
 * maxiter = -1).  Limits of the refactorizations: see ipxk_lu_factorize (dense bump). */
int ipxk_ipm_driver_basis(ipxk_context* ctx, const double* b, const double* c,
                          const double* lb, const double* ub,
                          const ipxk_ipm_params* params, ipxk_ipm_info* info,
                          ipxint* basis_out, ipxint* status_out,
                          ipxk_interrupt_fn interrupt, void* interrupt_user);

/* ---- multi-GPU: rows of AI partitioned over ranks, one RCCL all-reduce per
 *      NormalMatrix apply ---------------------------- */
/* 128-byte RCCL unique id, created on rank 0 and broadcast by the launcher. */
int ipxk_comm_unique_id(void* id128);
/* The context must have been created from this rank's row slab (m = local
 * rows, all n columns, row indices local).  After this call dot products and
 * norms of the CR loop are global and ipxk_normal_apply all-reduces A_g' y_g. */
int ipxk_comm_init(ipxk_context* ctx, const void* id128, int rank, int nranks);
/* Alternative partition ("column partition"): the context
 * was created from this rank's slab of structural COLUMNS (all m rows, n =
 * local columns).  Vector arguments then are [local structural part; all m
 * slack entries] for (n+m)-vectors and full m-vectors; every m-vector and
 * every scalar of the CR loop is replicated, the one exchange step per
 * NormalMatrix::_Apply is the all-reduce of the m partial sums A_g t_g. */
int ipxk_comm_init_columns(ipxk_context* ctx, const void* id128, int rank, int nranks);
/* What the transport itself reports about the communicator of this context:
 * transport 0 = none, 1 = RCCL (nranks / rank from ncclCommCount /
 * ncclCommUserRank), 2 = the direct exchange (its rank table).  For the records
 * of a multi-GPU run (bench.py writes it into its JSON line). */
int ipxk_comm_info(const ipxk_context* ctx, int* transport, int* nranks, int* rank);

/* ---- measurement helpers (bench.py, section 8d) --------------------------- */
/* Runs `reps` NormalMatrix applies on resident device vectors and returns the
 * total elapsed milliseconds between HIP events on the context's stream. */
int ipxk_time_normal_apply(ipxk_context* ctx, const double* rhs_dev,
                           double* lhs_dev, int reps, double* ms_total);
/* algorithmic bytes of one NormalMatrix apply: 2*nnz*(4+8) + (n+m+2)*4 +
 * 8*(3n+4m) */
ipxint ipxk_normal_apply_bytes(const ipxk_context* ctx);
/* Which device layout the two sparse products of NormalMatrix::_Apply
 * (normal_matrix.cc:45-126) use on this model: layout[0] for t = W.*(A'y),
 * layout[1] for lhs = A t; 0 = phased (time-tiled), 1 = XCD-sliced tiles,
 * 2 = fused tiles (one slice, epilogue in the tile kernel), 3 = sorted
 * sub-tiles (the sliced layout's slices with the gathers of a tile issued in
 * address order; same partial sums as 1, bit for bit), 4 = sorted fused tiles
 * (one slice, gathers in address order, epilogue in the tile kernel: bit-identical
 * to 0 and 2, for matrices whose gathers have locality), 5 = accumulated tiles
 * (the sliced layout's slices, a row block's sums kept in LDS, the entries of a
 * tile in address order and in batches that hold one entry per row; same partial
 * sums as 1 for rows stored with ascending indices; used whenever the sliced
 * layout is and IPXK_SPMV_ACC is not 0), 6 = plain rows (the matrix as it is, 8
 * lanes per row, sums in storage order: bit-identical to 0, 2 and 4; a candidate of
 * the timing for matrices of at most 4M entries), 7 = fused accumulated tiles (5
 * with one slice and the epilogue in the tile kernel, for matrices whose gathers have
 * locality and whose rows are stored sorted: bit-identical to 0, 2, 4, 6).  The sliced layouts are
 * chosen by a property of the matrix (x larger than an XCD's L2 and gathers that
 * spread over the slices), between 1 and 3 the faster at ipxk_create; otherwise a
 * timing picks the fastest of phased, fused and sorted fused, which are
 * bit-identical (IPXK_SPMV_LAYOUT=phased|sliced|fused|sorted|sortedfused overrides);
 * us[6] receives the measured microseconds {pass1 phased, sliced (or sorted, if
 * that was kept), fused (or sorted fused), pass2 likewise} (0 = not timed). */
int ipxk_spmv_layout(const ipxk_context* ctx, int layout[2], double us[6]);
/* The guard of the explicit inverses of SplittedNormalMatrix::Prepare's device
 * form (ipxk_split_prepare*): narrow first / last levels of a sweep and large
 * dense blocks of the factors are applied as inverse(T) * b; substitution is
 * backward stable whatever the condition of T, a product with a computed inverse
 * is not, and IPX's late bases are ill conditioned by construction (that is what
 * the stability loop of src/basis.cc:130-152 and the residual test of
 * src/lu_factorization.cc:87-127 are for).  Every inverse is therefore probed at
 * Prepare with two fixed vectors, || T (M z) - z ||_inf: the inverted levels of a
 * sweep must meet 1e-10, the inverse of a dense block of the factors 1e-8
 * (IPXK_INVERSE_TOL sets both).  A dense block's inverse from the matrix cores that
 * misses 1e-8 gets up to two refinement steps X += X (I - D X) first
 * (IPXK_DENSE_INVERSE_REFINE); one that still misses it but stays below 1e-5 is held
 * against the blocked solve it would replace, probed with the same two vectors, and
 * kept if it is within four times that solve's own residual (an ill-conditioned
 * block leaves neither at 1e-8).  A block that fails keeps its level-scheduled /
 * blocked solve.  Returns the number of probes, of rejected inverses and the worst
 * accepted residual seen since ipxk_create; ipxk_split_inverse_refined the number
 * of refinement steps taken. */
int ipxk_split_inverse_stats(const ipxk_context* ctx, ipxint* probes,
                             ipxint* rejected, double* worst_residual);
ipxint ipxk_split_inverse_refined(const ipxk_context* ctx);
/* Inspection of the device layouts (tests: the layouts built on the device by
 * radix sorts, layout_device.hip, against the host builders, array by array;
 * IPXK_LAYOUT_BUILD=host forces the host builders).  The reference's
 * NormalMatrix copies nothing (normal_matrix.h:20-27): what replaces its
 * "construct = store a reference" is one upload + transpose + layout build per
 * model, timed in create_ms {upload + transpose, A' layouts, A layouts, rest}.
 * which: 0 = the gather matrix of A'y (rows = columns of A), 1 = of A t.
 * info: {use_sliced, use_sorted, use_sorted_fused, nlong, sliced.built, R,
 * nslices, nrb, nrows_pad, max_tile, bits of the fullest-slice share (double),
 * sorted.built, nslices, nsub, nrb, RB, nrows_pad, max_sub, slice_elems, fused,
 * nnz, P, G, RT*1e6 + Q*1e3, use_acc, acc.built, nslices, nrb, RB, nrows_pad,
 * slice_elems, # batches, # entries that waited for a later batch, use_acc_fused,
 * accf.built, nrb, RB, # batches, use_plain}. */
int ipxk_layout_info(const ipxk_context* ctx, int which, ipxint info[40],
                     double create_ms[4]);
/* array: 0 sliced tile_ptr (u32), 1 sliced cnt (u8), 2 sliced idx (i32), 3
 * sliced val (f64), 4 sorted sub_ptr (u32), 5 sorted cnt (u8), 6 sorted pack
 * (u32), 7 sorted val (f64), 8 / 9 / 10 the row-wise copy of the model (Transpose,
 * sparse_matrix.cc:120-151) as the device holds it: ptr (i32), idx (i32), val
 * (f64); 11 acc tile_batch (u32), 12 acc bptr (u32), 13 acc pack (u32), 14 acc
 * val (f64); 15-19 the fused accumulated tiles: tile_batch, bptr, pack (u32), val
 * (f64), xmin (i32); 20 xmin of the fused sorted tiles (i32).  Copies min(cap, size)
 * bytes into out; *nbytes = the array's size. */
int ipxk_layout_array(ipxk_context* ctx, int which, int array, void* out,
                      ipxint cap, ipxint* nbytes);
/* plain device allocation helpers so that callers without torch can hold
 * resident vectors */
int ipxk_dev_alloc(ipxk_context* ctx, ipxint bytes, void** ptr);
int ipxk_dev_free(ipxk_context* ctx, void* ptr);
int ipxk_dev_upload(ipxk_context* ctx, void* dst_dev, const void* src_host,
                    ipxint bytes);
int ipxk_dev_download(ipxk_context* ctx, void* dst_host, const void* src_dev,
                      ipxint bytes);

#ifdef __cplusplus
}
#endif
#endif /* IPX_KKT_HIP_H_ */
