// The object behind ipxk_context: one model's matrix resident on one GPU plus
// every workspace of the KKT path (allocated once, reused by every solve).
#pragma once

#include "internal.hpp"
#include "device_utils.hpp"

struct ncclComm;

namespace ipxk {

constexpr int kPartialStride = kMaxPartials + 8;   // doubles per partial array

// indices of the per-workgroup partial arrays inside Context::partials
enum PartialSlot {
    kPartRes0 = 0,   // scaled residual norm (max), parity 0
    kPartRes1,       //                          , parity 1
    kPartPdot,       // Cstep' * P * Cstep  (or Cstep'*Cstep for plain CR)
    kPartCdot,       // dot from C.Apply
    kPartRsdot,      // residual' * P * residual (every 5th iteration)
    kPartScratch,
    kNumPartialSlots
};

// A second copy of the model with rows and columns renumbered for locality (layout_device.hip, reorder_model): the loop of the
// preconditioned CR method of the diag path runs on it, in permuted numbering; everything else keeps the original numbering.
struct Reordered {
    bool active = false;                 // the copy exists and is the faster one
    int levels = 0;                      // depth of the breadth-first level structure the numbering comes from
    int components = 0;
    double ms = 0.0;                     // time of the analysis + the second copy inside ipxk_create
    float us_original = 0.f, us_reordered = 0.f;      // the two products of NormalMatrix::Apply, timed on both copies
    DevBuf<int> rowperm, rowinv, colperm, colinv;     // new -> old, old -> new
    DevBuf<int> Ap, Ai, Tp, Ti;          // the renumbered matrix: CSC and the row-wise copy
    DevBuf<double> Ax, Tx;
    GatherMatrix Acols, Arows;
    DevBuf<double> W, diagonal, resscale, rhs, y, tcols;       // the solver's vectors in the new numbering
    bool in_use = false;                 // set around the CR loop of kkt_diag_solve_dev: operator and preconditioner take the copy
};

struct SplitOperator;   // trisolve.hip
struct PrepareHost;     // trisolve.hip
struct LuState;         // lu.hip
struct MaxvolState;     // maxvolume.hip
struct NMatrix;         // nmatrix.hip

struct Context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int pointer_mode = IPXK_POINTER_HOST;
    // optional per-operator timing (ipxk_set_profiling): HIP events around every operator /
    // preconditioner / triangular-solve application of a solve, summed into ipxk_times afterwards
    bool profile_ops = false;
    ipxint (*interrupt)(void*) = nullptr;   // ipxk_set_interrupt: Control::InterruptCheck for calls without a callback argument
    void* interrupt_user = nullptr;
    bool timing_active = false;
    std::vector<hipEvent_t> time_events;
    std::vector<int> time_kinds;        // per recorded event: kind (begin) or -1-kind (end)
    size_t time_used = 0;

    // ---- model ----
    int64_t m = 0, n = 0, nnz = 0;
    std::vector<ipxint> h_Ap;           // column pointers (host)
    // host copies of the entries and of the row-wise copy: filled on demand only (ensure_host_model) -- the model
    // lives on the device, the layouts are built there (layout_device.hip)
    std::vector<ipxint> h_Ai;
    std::vector<double> h_Ax;
    std::vector<ipxint> h_ATp, h_ATi;
    std::vector<double> h_ATx;
    // the model on the device in plain form, 32-bit indices: CSC and the row-wise copy (Transpose,
    // src/sparse_matrix.cc:120-151).  Shared by the layout builders, N (nmatrix.hip) and the basis LU (lu.hip).
    DevBuf<int> pl_Ap, pl_Ai, pl_Tp, pl_Ti;
    DevBuf<double> pl_Ax, pl_Tx;
    bool have_plain = false;
    double create_ms[4] = {0, 0, 0, 0};   // ipxk_create: upload + transpose, Acols layouts, Arows layouts, the rest
    GatherMatrix Acols;                 // rows = columns of A  (computes A'y)
    GatherMatrix Arows;                 // rows = rows of A     (computes A t)
    Reordered reord;                    // the same matrix renumbered for locality, if that pays (layout_device.hip)
    int64_t num_dense = 0, nz_dense = 0;
    std::vector<ipxint> dense_cols;

    // ---- NormalMatrix ----
    DevBuf<double> W_own;               // n+m
    const double* W = nullptr;          // device pointer in use (W_own or caller's)
    bool normal_prepared = false;
    DevBuf<double> tcols;               // n workspace: t = Ws .* (A'y)

    // ---- DiagonalPrecond ----
    DevBuf<double> diagonal;            // m
    bool diag_factorized = false;
    int64_t kdense = 0;                 // # dense columns treated by SMW (0: plain diagonal)
    GatherMatrix AdCols;                // k rows x m: computes Ad' u
    GatherMatrix AdRows;                // m rows x k: computes Ad w
    DevBuf<double> chol, chol_inv, smw_work, smw_u, Wnodense;   // chol_inv: inverted 64 x 64 diagonal blocks (k > 64)
    DevBuf<double> schur_panel, schur_part;     // blocked Schur assembly: dense copy of the dense columns, per-slab partial S
    DevBuf<int> schur_colptr;
    DevBuf<int> chol_info;
    DevBuf<unsigned char> dense_mask;   // n, 1 for dense columns

    // ---- CR workspaces (m-vectors) ----
    DevBuf<double> v_rhs, v_lhs, v_residual, v_sresidual, v_step, v_Cstep, v_Cres, v_pCstep;
    DevBuf<double> v_resscale_in;       // staging for a host resscale
    DevBuf<double> partials;            // kNumPartialSlots * kPartialStride
    DevBuf<CrState> state;
    CrState* h_state = nullptr;         // pinned
    DevBuf<double> hist;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    std::vector<hipEvent_t> ev_window;

    // ---- KKTSolverDiag ----
    DevBuf<double> resscale;            // m
    DevBuf<double> k_a, k_b, k_x, k_y;  // staging for host vectors (n+m, m, n+m, m)
    DevBuf<double> k_tmp;               // n+m scratch
    bool kkt_diag_factorized = false;

    // ---- IPM::SolveNewtonSystem (newton.hip) ----
    DevBuf<double> nw_rhs1, nw_rhs2;    // n+m, m
    DevBuf<double> nw_in[10], nw_out[6];   // staging for host vectors
    DevBuf<unsigned char> nw_state;

    // ---- the IPM iterate (iterate.hip) ----
    DevBuf<double> it_x, it_xl, it_xu, it_y, it_zl, it_zu, it_partials;
    DevBuf<unsigned char> it_state;
    bool it_set = false;
    DevBuf<double> ipm[12];            // residuals, complementarity targets and the step of ipxk_ipm_step

    // ---- basis path ----
    // guard of the explicit inverses (trisolve.hip): # probes and # inverses rejected since the context was created, worst residual
    struct { long inverse_probes = 0, inverse_rejected = 0, inverse_refined = 0; double worst_probe = 0.0; } split_stats;
    SplitOperator* split = nullptr;
    SplitOperator* split_spare = nullptr;  // the operator's buffers after ipxk_reset_solver_state: reused by the next Prepare, never applied
    // LU pivot tolerance of Maxvolume's refactorizations: tightened after an unstable exchange and kept for the later calls of the SAME
    // solver object, like Basis::TightenLuPivotTol changes lu_->pivottol() for good (src/basis.cc:490-503); ipxk_reset_solver_state
    // sets it again (src/basis.cc:30: every Basis starts from control.lu_pivottol())
    double maxvol_pivottol = 0.1;
    PrepareHost* prepare_host = nullptr;   // host workspaces of split_prepare (trisolve.hip)
    LuState* lu = nullptr;                 // factors of the last ipxk_lu_factorize* (lu.hip)
    DevBuf<double> dense_work;             // workspace of the dense block inverse (dense_inverse.hip), grow-only
    MaxvolState* maxvol = nullptr;         // workspaces of ipxk_maxvolume (maxvolume.hip)
    bool etas_live = false;                // the operator `split` stands for a LATER basis than its factors: Maxvolume's last exchanges are applied as etas behind them (maxvolume.hip)
    NMatrix* nmat = nullptr;               // N of the split operator as a matrix of its own (nmatrix.hip)

    // ---- multi-GPU ----
    ncclComm* comm = nullptr;
    struct DirectComm* direct = nullptr;   // hand-written exchange over hipIpc-mapped peer buffers (IPXK_COMM=direct)
    int rank = 0, nranks = 1;
    int64_t m_global = 0;               // rows of the whole system (sum over ranks)
    bool force_comm = false;
    // false: rows of AI partitioned (m local, n global; scalars and A'y need exchange steps);
    // true: structural COLUMNS partitioned (m global, n local; every m-vector and every scalar of
    // the CR loop is replicated, the only exchange is the sum of the partial products A_g t_g)
    bool col_partition = false;
    DevBuf<double> comm_scalars;        // scratch scalars (single-value reductions)
    DevBuf<double> comm_send;           // kNumPartialSlots scalars of this rank
    DevBuf<double> comm_gather;         // nranks * kNumPartialSlots, rank-major
    int* h_cycle_done = nullptr;        // mapped pinned ring: `done` at the end of each cycle
    int* d_cycle_done = nullptr;

    Context() = default;
    ~Context();
    double* part(int slot) const { return partials.get() + (size_t)slot * kPartialStride; }
};

// layout_device.hip
struct LayoutScratch;
LayoutScratch* new_layout_scratch();
void free_layout_scratch(LayoutScratch* S);
void upload_plain_model(Context* c, const ipxint* Ap, const ipxint* Ai, const double* Ax);
void ensure_host_model(Context* c, bool rowwise);
void fetch_columns(Context* c, const std::vector<ipxint>& cols, std::vector<ipxint>& Cp, std::vector<ipxint>& Ci, std::vector<double>& Cx);
int device_max_row_length(LayoutScratch& S, int nrows, const int* dptr, hipStream_t s);
// rows of more than kMaxRowLen entries out of a row-wise matrix: fills G's long-row arrays, returns the matrix without them
bool device_strip_long_rows(LayoutScratch& S, GatherMatrix& G, int nrows, const int* dptr, const int* didx, const double* dval,
                            DevBuf<int>& sptr, DevBuf<int>& sidx, DevBuf<double>& sval, int64_t* nnz_short, hipStream_t s);
bool device_build_sliced(LayoutScratch& S, SlicedMatrix& out, int nrows, int ncols, int64_t nnz, const int* dptr, const int* didx,
                         const double* dval, hipStream_t s, int ns_request = 0);
bool device_build_sorted_fused(LayoutScratch& S, SortedMatrix& out, int nrows, int ncols, int64_t nnz, const int* dptr, const int* didx,
                               const double* dval, hipStream_t s);
bool device_build_acc_fused(LayoutScratch& S, AccMatrix& out, int nrows, int ncols, int64_t nnz, const int* dptr, const int* didx,
                            const double* dval, hipStream_t s);
bool device_build_sorted(LayoutScratch& S, SortedMatrix& out, const SlicedMatrix& sliced, int nrows, int ncols, int64_t nnz, const int* dptr,
                         const int* didx, const double* dval, hipStream_t s);

int acc_rows_per_block(int nrows, int ns);
bool device_build_acc(LayoutScratch& S, AccMatrix& out, const SlicedMatrix& sliced, int nrows, int ncols, int64_t nnz, const int* dptr,
                      const int* didx, const double* dval, hipStream_t s);

// layout_device.hip: the locality-recovering renumbering (SURVEY section 7 / 8d "row/column reordering ... a pure permutation")
void reorder_model(Context* c);
float time_normal_pair(Context* c, GatherMatrix& Ac, GatherMatrix& Ar);     // spmv.hip: microseconds of the two products
void reorder_permute_rows(Context* c, const double* in_old, double* out_new);      // out_new[i'] = in_old[rowperm[i']]
void reorder_unpermute_rows(Context* c, const double* in_new, double* out_old);    // out_old[rowperm[i']] = in_new[i']
void reorder_permute_weights(Context* c, const double* W_old, double* W_new);      // n structural by colperm, m slack by rowperm

// dense_inverse.hip
void dense_lu_inverse(Context* c, int kb, const double* D, const double* invL, const double* invU, double* X, double* Xt, int refine = 0);

enum TimeKind { kTimeOp = 0, kTimePrecond = 1, kTimeB = 2, kTimeBt = 3, kNumTimeKinds = 4 };
void time_mark(Context* c, int kind, bool begin);      // no-op unless timing is active
void time_start(Context* c, ipxk_times* times);        // called by a solve before it enqueues work
void time_collect(Context* c, ipxk_times* times);      // after the solve's stream sync

// staging of vector arguments according to the pointer mode
const double* stage_in(Context* c, const double* p, size_t len, DevBuf<double>& buf);
double* stage_out(Context* c, double* p, size_t len, DevBuf<double>& buf);
void finish_out(Context* c, double* user, const double* dev, size_t len);

// ---- spmv.hip ----
void normal_apply_dev(Context* c, const double* W, const double* rhs, double* lhs, int* ndot,
                      const int* done);
void build_model(Context* c, const ipxint* Ap, const ipxint* Ai, const double* Ax);
void debug_single_pass(Context* c, int which, const double* x, double* out);

// ---- precond.hip ----
void prepare_dense_columns(Context* c);   // model-dependent part of the dense-column preconditioner (precond.hip)
void diag_factorize_dev(Context* c, const double* W, bool precond_dense_cols, ipxint* errflag);
// lhs = P rhs; partial dot rhs'lhs -> part(slot); returns # partials
int diag_apply_dev(Context* c, const double* rhs, double* lhs, int slot, const int* done);

// ---- cr.hip ----
struct CrResult { ipxint iter; ipxint errflag; };
CrResult pcr_solve_dev(Context* c, const double* rhs, double tol, const double* resscale,
                       ipxint maxiter, double* lhs, bool lhs_is_zero, ipxk_interrupt_fn interrupt,
                       void* user, double* hist_host, ipxint hist_cap, ipxk_times* times);
CrResult cr_solve_dev(Context* c, const double* rhs, double tol, const double* resscale,
                      ipxint maxiter, double* lhs, bool lhs_is_zero, ipxk_interrupt_fn interrupt,
                      void* user, double* hist_host, ipxint hist_cap, ipxk_times* times);
double reduce_partials_host(Context* c, int slot, int count, bool is_max);
void cr_diagnostics_dev(Context* c, ipxk_cr_diag* out);
// partitioned runs: finalize this rank's partials of `slot`, all-gather, return the per-rank view
struct PartRef publish_scalar(Context* c, int slot, int count, int op);
// finalize this rank's partials of `slot`, all-reduce the scalar; returns a one-element view
struct PartRef allreduce_scalar(Context* c, int slot, int count, int op);

// ---- newton.hip ----
CrResult newton_solve_dev(Context* c, bool use_basis, const double* rb, const double* rc, const double* rl,
                          const double* ru, const double* sl, const double* su, const double* xl, const double* xu,
                          const double* zl, const double* zu, const unsigned char* state, double tol,
                          ipxint maxiter, double* dx, double* dxl, double* dxu, double* dy, double* dzl,
                          double* dzu, ipxk_interrupt_fn interrupt, void* user, ipxk_times* times);

// ---- iterate.hip ----
void iterate_update_dev(Context* c, double sp, const double* dx, const double* dxl, const double* dxu, double sd,
                        const double* dy, const double* dzl, const double* dzu);
void iterate_residuals_dev(Context* c, const double* b, const double* cc, const double* lb, const double* ub,
                           double* rb, double* rc, double* rl, double* ru, double* presidual, double* dresidual);
void iterate_complementarity_dev(Context* c, double out4[4], double* num_terms = nullptr);
void iterate_objectives_dev(Context* c, const double* b, const double* cc, const double* lb, const double* ub, double out3[3]);
void model_norms_dev(Context* c, const double* b, const double* cc, const double* lb, const double* ub, double out2[2]);
double step_to_boundary_dev(Context* c, const double* x, const double* dx, int64_t len, double alpha0,
                            ipxint* blocking);

// ---- ipm_step.hip ----
void ipm_step_dev(Context* c, bool use_basis, const double* b, const double* cc, const double* lb, const double* ub,
                  double kkt_tol, ipxint maxiter, ipxk_ipm_step_info* info, ipxk_interrupt_fn interrupt, void* user);

void ipm_driver_dev(Context* c, const double* b, const double* cc, const double* lb, const double* ub,
                    const ipxk_ipm_params* prm, ipxk_ipm_info* info, ipxk_interrupt_fn interrupt, void* user,
                    bool use_basis = false, ipxint* basis_out = nullptr, ipxint* status_out = nullptr);

// ---- kkt_diag.hip ----
void kkt_diag_factorize_dev(Context* c, const double* xl, const double* xu, const double* zl,
                            const double* zu, double mu, bool precond_dense_cols, ipxint* errflag);
CrResult kkt_diag_solve_dev(Context* c, const double* a, const double* b, double tol,
                            ipxint maxiter, double* x, double* y, ipxk_interrupt_fn interrupt,
                            void* user, ipxk_times* times);

// ---- trisolve.hip / kkt_basis.hip ----
void split_prepare_host(Context* c, const ipxint* Lp, const ipxint* Li, const double* Lx,
                        const ipxint* Up, const ipxint* Ui, const double* Ux, const ipxint* rowperm,
                        const ipxint* colperm, const ipxint* basis, const ipxint* status,
                        const double* colscale);
// lhs = C rhs (device vectors), dot partials -> part(kPartCdot); returns # partials
int split_apply_dev(Context* c, const double* rhs, double* lhs, const int* done);
void split_rescale_host(Context* c, const ipxint* status, const double* colscale);
// the operator follows Maxvolume's exchanges without new factors (Context::etas_live): the basis by position (device), its statuses and scaling
void split_follow_basis(Context* c, const ipxint* basis_dev, const ipxint* status, const double* colscale);
// out = inverse(B) in / inverse(B') in on the (scaled) factors; in may be out
void forward_solve_dev(Context* c, const double* in, double* out, bool scaled, const int* done);
void backward_solve_dev(Context* c, const double* in, double* out, bool scaled, const int* done);
void solve_dense_dev(Context* c, const double* rhs, double* lhs, char trans);
CrResult kkt_basis_solve_dev(Context* c, const double* a, const double* b, double tol,
                             ipxint maxiter, double* x, double* y, ipxk_interrupt_fn interrupt,
                             void* user, ipxk_times* times);
void split_levels(const Context* c, ipxint levels[4]);
void check_sweep_abort(Context* c);
void destroy_split(SplitOperator*);
void destroy_prepare_host(PrepareHost*);

// ---- lu.hip ----
void lu_factorize_host(Context* c, int64_t dim, const ipxint* Bbegin, const ipxint* Bend, const ipxint* Bi,
                       const double* Bx, double pivottol, bool strict, ipxk_lu_info* info);
void lu_factorize_basis(Context* c, const ipxint* basis, double pivottol, bool strict, ipxk_lu_info* info);
void lu_get_factors(Context* c, ipxint* Lp, ipxint* Li, double* Lx, ipxint* Up, ipxint* Ui, double* Ux,
                    ipxint* rowperm, ipxint* colperm, ipxint* dependent);
void split_prepare_lu(Context* c, const ipxint* status, const double* colscale);
void destroy_lu(LuState*);
long lu_generation(const Context* c);
void lu_invalidate(Context* c);          // the factors of an earlier factorization are no longer handed out
// ---- maxvolume.hip ----
void maxvolume_dev(Context* c, const ipxint* status, const double* colscale, const ipxk_maxvolume_params* prm,
                   ipxint* basis_out, ipxint* status_out, ipxk_maxvolume_info* info, ipxint* log, ipxint log_cap);
void maxvolume_sequential_dev(Context* c, const ipxint* status, const double* colscale, double volume_tol, ipxint maxpasses,
                              ipxint max_etas, ipxint* basis_out, ipxint* status_out, ipxk_maxvolume_info* info, ipxint* log,
                              ipxint log_cap);
void destroy_maxvol(MaxvolState*);
void maxvol_apply_etas(Context* c, bool transposed, double* v);     // v (by basis position) <- inverse(E_K ... E_1) v  /  its transpose
const ipxint* maxvol_current_basis(Context* c);                     // device, by basis position: the basis factors + etas stand for
void maxvol_drop_etas(Context* c);                                  // a new factorization / operator: the eta file is history
void destroy_nmatrix(NMatrix*);
void nmatrix_invalidate(Context* c);
// N = the NONBASIC columns of A with nonzero weight as a pair of gather matrices built on the device (nmatrix.hip):
// prepare returns false when N is not used for this model; apply: work = W_I .* u + N (W_N .* (N' u))
bool nmatrix_prepare(Context* c, const double* W);
void nmatrix_apply(Context* c, const double* WI, const double* u, double* work, const int* done);
// ---- presolve.hip (stand-alone, no context) ----
void equilibrate_device(int device, int64_t m, int64_t n, const ipxint* Ap, const ipxint* Ai, double* Ax,
                        double* colscale, double* rowscale, ipxint* rounds);
void transpose_device(int device, int64_t m, int64_t n, const ipxint* Ap, const ipxint* Ai, const double* Ax,
                      ipxint* Tp, ipxint* Ti, double* Tx);

// ---- comm.hip ----
void comm_allreduce_sum(Context* c, double* buf, size_t count);
void comm_allreduce_max(Context* c, double* buf, size_t count);
void comm_allgather(Context* c, const double* send, double* recv, size_t count_per_rank);
bool comm_active(const Context* c);      // more than one rank (or forced for testing)
bool comm_rows(const Context* c);        // active communicator, row partition
bool comm_cols(const Context* c);        // active communicator, column partition
void comm_allreduce_min(Context* c, double* buf, size_t count);
void comm_destroy(Context* c);
void comm_check(Context* c);             // raises if a collective of the direct transport timed out
double* comm_stage(Context* c, size_t count);
void comm_allreduce_sum_staged(Context* c, double* dst, size_t count);

}  // namespace ipxk

struct ipxk_context : ipxk::Context {};
