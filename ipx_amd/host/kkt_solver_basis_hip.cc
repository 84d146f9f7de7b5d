#include "kkt_solver_basis_hip.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdlib>

#include "maxvolume.h"
#include "sparse_matrix.h"
#include "splitted_normal_matrix.h"
#include "timer.h"

// measurement aid (IPXK_TIME_CPU_PREPARE=1): seconds the reference's own SplittedNormalMatrix::Prepare would take on
// the bases of this run -- work KKTSolverBasis::_Factorize does for its CPU operator and this class no longer does
static double g_cpu_prepare_seconds = 0.0;
static long g_cpu_prepare_calls = 0;
extern "C" double ipx_hip_cpu_prepare_seconds() { return g_cpu_prepare_seconds; }
extern "C" long ipx_hip_cpu_prepare_calls() { return g_cpu_prepare_calls; }
// # Factorize calls whose Maxvolume ran on the device / on the CPU, over all solver objects of the process
static long g_device_maxvolume_calls = 0, g_cpu_maxvolume_calls = 0;
extern "C" long ipx_hip_device_maxvolume_calls() { return g_device_maxvolume_calls; }
extern "C" long ipx_hip_cpu_maxvolume_calls() { return g_cpu_maxvolume_calls; }
// # of those calls on the device that ended WITHOUT the final refactorization: the last exchanges stay behind the factors as etas
// (ipxk_maxvolume_info.kept_etas), and the etas kept over all of them
static long g_kept_eta_calls = 0, g_kept_etas = 0;
extern "C" long ipx_hip_kept_eta_calls() { return g_kept_eta_calls; }
extern "C" long ipx_hip_kept_etas() { return g_kept_etas; }
// where KKTSolverBasisHip::Factorize spends its time, over all solver objects of the process (seconds): the reference's
// DropPrimal / DropDual, the device LU + Prepare of the current basis, Maxvolume on the device, Basis::Load / Factorize of
// the reference's Basis afterwards, the CPU path (Maxvolume on Basis + hand-off)
static double g_phase_seconds[5] = {0, 0, 0, 0, 0};
extern "C" double ipx_hip_factorize_phase_seconds(int phase) { return phase >= 0 && phase < 5 ? g_phase_seconds[phase] : 0.0; }

namespace ipx {

namespace {
ipxint PollInterrupt(void* control) {
    return static_cast<const Control*>(control)->InterruptCheck();
}
}  // namespace

KKTSolverBasisHip::KKTSolverBasisHip(const Control& control, Basis& basis)
    : control_(control), model_(basis.model()), basis_(basis), device_(basis.model()), reference_(control, basis) {
    // the context may have served another solver object before (HipModel): it starts like a new one, and Maxvolume's
    // refactorizations start from this run's pivot tolerance, as every Basis does (src/basis.cc:30)
    HipCheck(ipxk_reset_solver_state(device_.get(), control_.lu_pivottol()));
    // time_cr2_NNt / _B / _Bt are only printed at debug level >= 2 (src/lp_solver.cc:107)
    HipCheck(ipxk_set_profiling(device_.get(), control_.Debug(2) ? 1 : 0));
    HipCheck(ipxk_set_interrupt(device_.get(), PollInterrupt, const_cast<Control*>(&control_)));
}

KKTSolverBasisHip::~KKTSolverBasisHip() {
    // whoever comes next (crossover, LpSolver's basic solution) reads the reference's Basis: it learns the final basis now at the
    // latest.  The device factorized exactly this basis without dependent columns, and Load's factorization is those factors handed
    // out again (LuKernelHip::SharedWithSolver), so there is no error to report from here; an exception must not leave a destructor.
    if (basis_pending_) {
        try {
            Timer timer;
            (void)SyncBasis();
            g_phase_seconds[3] += timer.Elapsed();
        } catch (...) {
        }
    }
    // the context outlives this object (HipModel registry): it must not keep a pointer to this solver's Control
    ipxk_set_interrupt(device_.get(), nullptr, nullptr);
}

const Basis* KKTSolverBasisHip::_basis() const {
    // IPM::PrintOutput asks for the Basis every iteration but reads it only at debug level 4 (src/ipm.cc:685-691)
    if (basis_pending_ && control_.Debug(4))
        (void)const_cast<KKTSolverBasisHip*>(this)->SyncBasis();
    return &basis_;
}

Int KKTSolverBasisHip::SyncBasis() {
    if (!basis_pending_)
        return 0;
    basis_pending_ = false;
    std::vector<int> basic_status(device_status_.begin(), device_status_.end());
    return basis_.Load(basic_status.data());      // loads and factorizes (src/basis.cc:81-114)
}

// Would the reference's DropPrimal / DropDual find a candidate?  Their screening conditions (src/kkt_solver_basis.cc:208-216:
// a BASIC variable whose nearer bound is closer than ipm_drop_primal and 100 times closer than its dual; :306-314: a NONBASIC
// variable whose larger dual is below ipm_drop_dual and 100 times below its primal distance) evaluated on the basis the device
// holds.  Only used to decide whether the reference's Basis must be brought up to date before its own procedures run.
bool KKTSolverBasisHip::DegenerateCandidateExists(const Iterate& iterate) const {
    const Vector &xl = iterate.xl(), &xu = iterate.xu(), &zl = iterate.zl(), &zu = iterate.zu();
    const double near_bound = control_.ipm_drop_primal(), near_zero = control_.ipm_drop_dual();
    for (std::size_t j = 0; j < device_status_.size(); j++) {
        if (device_status_[j] == Basis::BASIC) {
            const bool at_lower = xl[j] <= xu[j];
            const double dist = at_lower ? xl[j] : xu[j], dual = at_lower ? zl[j] : zu[j];
            if (dist <= near_bound && dist < 0.01 * dual)
                return true;
        } else if (device_status_[j] == Basis::NONBASIC) {
            const bool lower_active = zl[j] >= zu[j];
            const double dual = lower_active ? zl[j] : zu[j], dist = lower_active ? xl[j] : xu[j];
            if (dual <= near_zero && dual < 0.01 * dist)
                return true;
        }
    }
    return false;
}

Int KKTSolverBasisHip::_basis_changes() const { return reference_.basis_changes_; }

// The reference's two procedures for degenerate variables, run by the reference's own object on the Basis both share
// (src/kkt_solver_basis.cc:30-43 decides when; :196-387 are the procedures).  They pivot on the reference's Basis
// (TableauRow / SolveForUpdate / ExchangeIfStable), change variable states in *iterate, overwrite entries of
// reference_.colscale_ (INFINITY for implied, 0 for fixed variables) and count their exchanges in
// reference_.basis_changes_ and info->updates_ipm.
void KKTSolverBasisHip::DropDegenerateVariables(Iterate* iterate, Info* info) {
    if (iterate->pobjective() < iterate->dobjective())
        return;                      // possibly infeasible / unbounded: the reference removes nothing then (:30-35)
    if (basis_pending_) {
        // the reference's Basis still holds the basis of an earlier iteration: without a candidate its two procedures would do
        // nothing on the CURRENT basis, so they are not called on the old one; with one, the Basis is brought up to date first
        if (!DegenerateCandidateExists(*iterate))
            return;
        Timer timer;
        info->errflag = SyncBasis();
        g_phase_seconds[3] += timer.Elapsed();
        if (info->errflag)
            return;
    }
    reference_.DropPrimal(iterate, info);
    if (!info->errflag)
        reference_.DropDual(iterate, info);
}

// src/kkt_solver_basis.cc:20-67 with the Maxvolume / refactorize / Prepare tail (:45-64) on the device
void KKTSolverBasisHip::_Factorize(Iterate* iterate, Info* info) {
    static const bool on_device = !(std::getenv("IPXK_DEVICE_MAXVOLUME") && std::getenv("IPXK_DEVICE_MAXVOLUME")[0] == '0');
    Vector& colscale = reference_.colscale_;
    factorized_ = false;
    iter_ = 0;
    reference_.basis_changes_ = 0;
    info->errflag = 0;
    for (Int j = 0; j < (Int)colscale.size(); j++)
        colscale[j] = iterate->ScalingFactor(j);
    Timer timer;
    DropDegenerateVariables(iterate, info);
    g_phase_seconds[0] += timer.Elapsed();
    if (info->errflag)
        return;
    if (!(on_device && MaxvolumeOnDevice(info))) {
        timer.Reset();
        if (!info->errflag)
            info->errflag = SyncBasis();          // the reference's own path works on its Basis
        if (!info->errflag)
            MaxvolumeOnBasis(info);
        g_phase_seconds[4] += timer.Elapsed();
    }
    factorized_ = info->errflag == 0;
}

bool KKTSolverBasisHip::MaxvolumeOnDevice(Info* info) {
    const Int m = model_.rows();
    const Int n = model_.cols();
    ipxk_context* ctx = device_.get();
    std::vector<Int> status(n+m), basic(m);
    std::vector<double> colscale(n+m);
    // (the LU kernel of the reference's Basis may go through this context too -- LuKernelHip::SharedWithSolver: a refactorization
    // by DropPrimal / DropDual's ExchangeIfStable replaces the resident factors, and the generation counter tells)
    bool same_basis = device_lu_valid_ && (Int)device_member_.size() == n+m && ipxk_lu_generation(ctx) == device_lu_generation_;
    for (Int j = 0; j < n+m; j++) {
        colscale[j] = reference_.colscale_[j];
        status[j] = basis_pending_ ? device_status_[j] : static_cast<Int>(basis_.StatusOf(j));
        const bool member = status[j] == Basis::BASIC || status[j] == Basis::BASIC_FREE;
        if (same_basis && member != (device_member_[j] != 0))
            same_basis = false;
    }
    for (Int p = 0; p < m; p++)
        basic[p] = basis_pending_ ? device_basis_[p] : basis_[p];
    Timer timer;
    // the factors of the current basis on the device: what the previous call left (its final refactorization, or earlier factors
    // with its last exchanges behind them as etas) when the basis is still the same set of columns -- then only the scaling is
    // handed over --, else a fresh device LU
    if (same_basis) {
        HipCheck(ipxk_split_rescale(ctx, status.data(), colscale.data()));
    } else {
        device_lu_valid_ = false;
        ipxk_lu_info lu;
        int rc = ipxk_lu_factorize_basis(ctx, basic.data(), control_.lu_pivottol(), 0, &lu);
        if (rc == IPXK_E_UNSUPPORTED)
            return false;
        HipCheck(rc);
        if (lu.num_dependent > 0)
            return false;                       // the reference repairs the basis (Basis::Factorize): its path
        HipCheck(ipxk_split_prepare_lu(ctx, status.data(), colscale.data()));
    }
    g_phase_seconds[1] += timer.Elapsed();
    Timer timer_maxvol;
    std::vector<Int> basis_out(m), status_out(n+m);
    ipxk_maxvolume_info mv;
    int rc;
    if (control_.update_heuristic() == 0) {
        rc = ipxk_maxvolume_sequential(ctx, status.data(), colscale.data(), control_.volume_tol(), control_.maxpasses(), -1,
                                       basis_out.data(), status_out.data(), &mv, nullptr, 0);
    } else {
        ipxk_maxvolume_params prm;
        prm.volume_tol = control_.volume_tol();
        prm.maxskip_updates = control_.maxskip_updates();
        prm.rows_per_slice = control_.rows_per_slice();
        prm.max_etas = -1;          // refactorize when the etas have cost as much as a refactorization (ipx_kkt_hip.h)
        rc = ipxk_maxvolume(ctx, status.data(), colscale.data(), &prm, basis_out.data(), status_out.data(), &mv, nullptr, 0);
    }
    if (rc == IPXK_E_UNSUPPORTED) {            // a refactorization on the way hit the dense limit: nothing was handed back yet
        device_lu_valid_ = false;
        return false;
    }
    HipCheck(rc);
    g_phase_seconds[2] += timer_maxvol.Elapsed();
    device_maxvolume_calls_++;
    g_device_maxvolume_calls++;
    if (control_.update_heuristic() != 0 && mv.kept_etas > 0) {
        g_kept_eta_calls++;
        g_kept_etas += mv.kept_etas;
    }
    info->updates_ipm += mv.updates;
    info->time_maxvol += timer.Elapsed();
    reference_.basis_changes_ += mv.updates;
    info->errflag = mv.errflag;
    if (info->errflag)
        return true;
    control_.Debug()
        << " Maxvolume on the device: " << mv.updates << " updates, " << mv.skipped << " skipped, "
        << mv.factorizations << " factorizations, volume increase 2^" << sci2(mv.volinc) << '\n';
    // the device holds the operator of the final basis (fresh factors, or the earlier ones + etas: mv.kept_etas)
    device_member_.assign(n+m, 0);
    for (Int p = 0; p < m; p++)
        device_member_[basis_out[p]] = 1;
    device_lu_valid_ = true;
    device_lu_generation_ = ipxk_lu_generation(ctx);
    prepared_once_ = false;                   // (the factors of an earlier GetLuFactors hand-off are gone)
    if (mv.updates > 0) {
        // the reference's Basis learns the final basis when somebody needs it there (see basis_pending_): Basis::Load -- loads
        // and factorizes (src/basis.cc:81-114), as :57-61 would -- with the resident factors handed out again
        device_status_ = status_out;
        device_basis_ = basis_out;
        basis_pending_ = true;
        static const bool eager = std::getenv("IPXK_EAGER_BASIS_LOAD") != nullptr;     // (measurement: Load at once, as until round 5)
        if (eager) {
            Timer timer_load;
            info->errflag = SyncBasis();
            g_phase_seconds[3] += timer_load.Elapsed();
        }
    } else if (!basis_pending_ && !basis_.FactorizationIsFresh()) {
        Timer timer_load;
        info->errflag = basis_.Factorize();     // updates by DropPrimal / DropDual (:57-61)
        g_phase_seconds[3] += timer_load.Elapsed();
    }
    return true;
}

// the reference's own path: Maxvolume on its Basis, then the hand-off of the fresh factors
void KKTSolverBasisHip::MaxvolumeOnBasis(Info* info) {
    const Int m = model_.rows();
    const Int n = model_.cols();
    cpu_maxvolume_calls_++;
    g_cpu_maxvolume_calls++;
    device_lu_valid_ = false;
    Maxvolume maxvol(control_);
    if (control_.update_heuristic() == 0) {
        info->errflag = maxvol.RunSequential(&reference_.colscale_[0], basis_);
    } else {
        info->errflag = maxvol.RunHeuristic(&reference_.colscale_[0], basis_);
    }
    info->updates_ipm += maxvol.updates();
    info->time_maxvol += maxvol.time();
    reference_.basis_changes_ += maxvol.updates();
    if (info->errflag)
        return;
    if (!basis_.FactorizationIsFresh()) {
        info->errflag = basis_.Factorize();
        if (info->errflag)
            return;
    }
    // The device keeps the factors of the previous hand-off.  They are still the factors Basis holds iff no LU
    // factorization happened since (Basis::factorizations(), src/basis.h:214, counts every one -- also a
    // refactorization of an UNCHANGED basis after Basis::TightenLuPivotTol, which basis_changes() == 0 would
    // miss) and no update was applied on top of them.  Then only the scaling changed.
    const bool same_factors = prepared_once_ && basis_.FactorizationIsFresh() &&
                              basis_.factorizations() == factorizations_at_handoff_;
    factorizations_at_handoff_ = basis_.factorizations();
    std::vector<double> colscale(n + m);
    std::vector<Int> status(n + m), basic(m);
    for (Int j = 0; j < n + m; j++) {
        colscale[j] = reference_.colscale_[j];
        status[j] = basis_.StatusOf(j);
    }
    for (Int p = 0; p < m; p++)
        basic[p] = basis_[p];
    static const bool time_cpu_prepare = std::getenv("IPXK_TIME_CPU_PREPARE") != nullptr;
    if (time_cpu_prepare) {
        SplittedNormalMatrix probe(model_);
        Timer timer;
        probe.Prepare(basis_, colscale.data());
        g_cpu_prepare_seconds += timer.Elapsed();
        g_cpu_prepare_calls++;
    }
    // Hand-off of the LU factors: B[rowperm,colperm] = (L+I)*U (src/lu_update.h:43-60).
    SparseMatrix L, U;
    std::vector<Int> rowperm(m), colperm(m);
    if (!same_factors)
        basis_.GetLuFactors(&L, &U, rowperm.data(), colperm.data());
    const ipx_hip::BasisHandoff h{m, n, L.colptr(), L.rowidx(), L.values(), U.colptr(), U.rowidx(),
                                  U.values(), rowperm.data(), colperm.data(), basic.data(),
                                  status.data(), colscale.data()};
    ipx_hip::HandOffBasis(device_.get(), h, same_factors);
    prepared_once_ = true;
}

void KKTSolverBasisHip::_Solve(const Vector& a, const Vector& b, double tol,
                                Vector& x, Vector& y, Info* info) {
    assert(factorized_);
    const ipx_hip::SolveOutcome r = ipx_hip::SolveBasisOnDevice(
        device_.get(), &a[0], &b[0], tol, maxiter_, &x[0], &y[0], PollInterrupt,
        const_cast<Control*>(&control_), control_.parameters().debug >= 3);
    if (!r.debug3.empty())
        control_.Debug(3) << r.debug3;      // the reference's messages (src/conjugate_residuals.cc:53-56)
    info->errflag = r.errflag;
    info->kktiter2 += r.iter;
    info->time_cr2 += r.times.cr;
    info->time_cr2_NNt += r.times.op;
    info->time_cr2_B += r.times.solve_B;
    info->time_cr2_Bt += r.times.solve_Bt;
    iter_ += r.iter;
}

}  // namespace ipx
