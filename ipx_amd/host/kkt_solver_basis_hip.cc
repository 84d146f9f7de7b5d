#include "kkt_solver_basis_hip.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdlib>

#include "indexed_vector.h"
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

namespace ipx {

namespace {
ipxint PollInterrupt(void* control) {
    return static_cast<const Control*>(control)->InterruptCheck();
}
}  // namespace

KKTSolverBasisHip::KKTSolverBasisHip(const Control& control, Basis& basis)
    : control_(control), model_(basis.model()), basis_(basis), device_(basis.model()) {
    const Int m = model_.rows();
    const Int n = model_.cols();
    colscale_.resize(n+m);
    // time_cr2_NNt / _B / _Bt are only printed at debug level >= 2 (src/lp_solver.cc:107)
    HipCheck(ipxk_set_profiling(device_.get(), control_.Debug(2) ? 1 : 0));
    HipCheck(ipxk_set_interrupt(device_.get(), PollInterrupt, const_cast<Control*>(&control_)));
}

KKTSolverBasisHip::~KKTSolverBasisHip() {
    // the context outlives this object (HipModel registry): it must not keep a pointer to this solver's Control
    ipxk_set_interrupt(device_.get(), nullptr, nullptr);
}

// src/kkt_solver_basis.cc:20-67
void KKTSolverBasisHip::_Factorize(Iterate* iterate, Info* info) {
    const Int m = model_.rows();
    const Int n = model_.cols();
    info->errflag = 0;
    factorized_ = false;
    iter_ = 0;
    basis_changes_ = 0;

    for (Int j = 0; j < n+m; j++)
        colscale_[j] = iterate->ScalingFactor(j);

    // Remove degenerate variables unless the primal objective is smaller than the dual objective (:30-43).
    if (iterate->pobjective() >= iterate->dobjective()) {
        DropPrimal(iterate, info);
        if (info->errflag)
            return;
        DropDual(iterate, info);
        if (info->errflag)
            return;
    }

    // Maxvolume (:45-55), refactorization and the preconditioned normal matrix (:57-64)
    static const bool on_device = !(std::getenv("IPXK_DEVICE_MAXVOLUME") && std::getenv("IPXK_DEVICE_MAXVOLUME")[0] == '0');
    if (!(on_device && MaxvolumeOnDevice(info)))
        MaxvolumeOnBasis(info);
    if (info->errflag)
        return;
    factorized_ = true;
}

bool KKTSolverBasisHip::MaxvolumeOnDevice(Info* info) {
    const Int m = model_.rows();
    const Int n = model_.cols();
    ipxk_context* ctx = device_.get();
    std::vector<Int> status(n+m), basic(m);
    std::vector<double> colscale(n+m);
    bool same_basis = device_lu_valid_ && (Int)device_member_.size() == n+m;
    for (Int j = 0; j < n+m; j++) {
        colscale[j] = colscale_[j];
        status[j] = basis_.StatusOf(j);
        const bool member = status[j] == Basis::BASIC || status[j] == Basis::BASIC_FREE;
        if (same_basis && member != (device_member_[j] != 0))
            same_basis = false;
    }
    for (Int p = 0; p < m; p++)
        basic[p] = basis_[p];
    Timer timer;
    // the factors of the current basis on the device: those of the previous call's final refactorization when the
    // basis is still the same set of columns (then only the scaling is handed over), else a fresh device LU
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
    std::vector<Int> basis_out(m), status_out(n+m);
    ipxk_maxvolume_info mv;
    int rc;
    if (control_.update_heuristic() == 0) {
        rc = ipxk_maxvolume_sequential(ctx, status.data(), colscale.data(), control_.volume_tol(), control_.maxpasses(), 100,
                                       basis_out.data(), status_out.data(), &mv, nullptr, 0);
    } else {
        ipxk_maxvolume_params prm;
        prm.volume_tol = control_.volume_tol();
        prm.maxskip_updates = control_.maxskip_updates();
        prm.rows_per_slice = control_.rows_per_slice();
        prm.max_etas = 100;
        rc = ipxk_maxvolume(ctx, status.data(), colscale.data(), &prm, basis_out.data(), status_out.data(), &mv, nullptr, 0);
    }
    if (rc == IPXK_E_UNSUPPORTED) {            // a refactorization on the way hit the dense limit: nothing was handed back yet
        device_lu_valid_ = false;
        return false;
    }
    HipCheck(rc);
    device_maxvolume_calls_++;
    g_device_maxvolume_calls++;
    info->updates_ipm += mv.updates;
    info->time_maxvol += timer.Elapsed();
    basis_changes_ += mv.updates;
    info->errflag = mv.errflag;
    if (info->errflag)
        return true;
    control_.Debug()
        << " Maxvolume on the device: " << mv.updates << " updates, " << mv.skipped << " skipped, "
        << mv.factorizations << " factorizations, volume increase 2^" << sci2(mv.volinc) << '\n';
    // the device holds the fresh factorization and the operator of the final basis
    device_member_.assign(n+m, 0);
    for (Int p = 0; p < m; p++)
        device_member_[basis_out[p]] = 1;
    device_lu_valid_ = true;
    prepared_once_ = false;                   // (the factors of an earlier GetLuFactors hand-off are gone)
    if (mv.updates > 0) {
        // the reference's Basis learns the final basis: loads and factorizes (src/basis.cc:81-114), as :57-61 would
        std::vector<int> basic_status(n+m);
        for (Int j = 0; j < n+m; j++)
            basic_status[j] = static_cast<int>(status_out[j]);
        info->errflag = basis_.Load(basic_status.data());
    } else if (!basis_.FactorizationIsFresh()) {
        info->errflag = basis_.Factorize();     // updates by DropPrimal / DropDual (:57-61)
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
        info->errflag = maxvol.RunSequential(&colscale_[0], basis_);
    } else {
        info->errflag = maxvol.RunHeuristic(&colscale_[0], basis_);
    }
    info->updates_ipm += maxvol.updates();
    info->time_maxvol += maxvol.time();
    basis_changes_ += maxvol.updates();
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
        colscale[j] = colscale_[j];
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

// src/kkt_solver_basis.cc:196-290
void KKTSolverBasisHip::DropPrimal(Iterate* iterate, Info* info) {
    const Int m = model_.rows();
    const Int n = model_.cols();
    const Vector& xl = iterate->xl();
    const Vector& xu = iterate->xu();
    const Vector& zl = iterate->zl();
    const Vector& zu = iterate->zu();
    const double drop_primal = control_.ipm_drop_primal();
    const double volume_tol = 2.0;
    info->errflag = 0;

    std::vector<Int> candidates;
    for (Int p = 0; p < m; p++) {
        const Int jb = basis_[p];
        if (basis_.StatusOf(jb) != Basis::BASIC)     // free variables stay
            continue;
        const bool lower = xl[jb] <= xu[jb];         // the nearer bound
        const double xj = lower ? xl[jb] : xu[jb];
        const double zj = lower ? zl[jb] : zu[jb];
        if (xj < 0.01*zj && xj <= drop_primal)
            candidates.push_back(jb);
    }
    if (candidates.empty())
        return;

    IndexedVector btran(m), row(n+m);
    Vector invscale_basic(m);
    for (Int p = 0; p < m; p++)
        invscale_basic[p] = 1.0 / colscale_[basis_[p]];

    while (!candidates.empty()) {
        const Int jb = candidates.back();
        const Int p = basis_.PositionOf(jb);
        assert(p >= 0);
        const double s = invscale_basic[p];
        basis_.TableauRow(jb, btran, row, true);
        Int jmax = -1;
        double vmax = volume_tol;
        auto search_pivot = [&](Int j, double pivot) {
            pivot = std::abs(pivot);
            if (pivot > kPivotZeroTol) {
                const double v = pivot * colscale_[j] * s;
                if (v > vmax) {
                    vmax = v;
                    jmax = j;
                }
            }
        };
        for_each_nonzero(row, search_pivot);
        if (jmax >= 0) {
            const double pivot = row[jmax];
            if (std::abs(pivot) < 1e-3)
                control_.Debug(3)
                    << " |pivot| = " << sci2(std::abs(pivot))
                    << " (primal basic variable close to bound)\n";
            bool exchanged;
            info->errflag = basis_.ExchangeIfStable(jb, jmax, pivot, 1, &exchanged);
            if (info->errflag)
                return;
            if (!exchanged)      // the factorization was unstable and has been redone: same candidate again
                continue;
            invscale_basic[p] = 1.0 / colscale_[jmax];
            info->updates_ipm++;
            basis_changes_++;
        } else {
            // the variable becomes "implied" at a bound
            if (zl[jb]/xl[jb] > zu[jb]/xu[jb])
                iterate->make_implied_lb(jb);
            else
                iterate->make_implied_ub(jb);
            basis_.FreeBasicVariable(jb);
            invscale_basic[p] = 0.0;
            colscale_[jb] = INFINITY;
            info->primal_dropped++;
        }
        candidates.pop_back();
    }
}

// src/kkt_solver_basis.cc:292-387
void KKTSolverBasisHip::DropDual(Iterate* iterate, Info* info) {
    const Int m = model_.rows();
    const Int n = model_.cols();
    const Vector& xl = iterate->xl();
    const Vector& xu = iterate->xu();
    const Vector& zl = iterate->zl();
    const Vector& zu = iterate->zu();
    const double drop_dual = control_.ipm_drop_dual();
    const double volume_tol = 2.0;
    info->errflag = 0;

    std::vector<Int> candidates;
    for (Int jn = 0; jn < n+m; jn++) {
        if (basis_.StatusOf(jn) != Basis::NONBASIC)
            continue;
        const bool lower = zl[jn] >= zu[jn];         // the larger dual variable
        const double xj = lower ? xl[jn] : xu[jn];
        const double zj = lower ? zl[jn] : zu[jn];
        if (zj < 0.01*xj && zj <= drop_dual)
            candidates.push_back(jn);
    }
    if (candidates.empty())
        return;

    IndexedVector ftran(m);
    Vector invscale_basic(m);
    for (Int p = 0; p < m; p++)
        invscale_basic[p] = 1.0 / colscale_[basis_[p]];

    while (!candidates.empty()) {
        const Int jn = candidates.back();
        const double s = colscale_[jn];
        basis_.SolveForUpdate(jn, ftran);
        Int pmax = -1;
        double vmax = volume_tol;
        auto search_pivot = [&](Int p, double pivot) {
            pivot = std::abs(pivot);
            if (pivot > kPivotZeroTol) {
                const double v = pivot * invscale_basic[p] * s;
                if (v > vmax) {
                    vmax = v;
                    pmax = p;
                }
            }
        };
        for_each_nonzero(ftran, search_pivot);
        if (pmax >= 0) {
            const double pivot = ftran[pmax];
            if (std::abs(pivot) < 1e-3)
                control_.Debug(3)
                    << " |pivot| = " << sci2(std::abs(pivot))
                    << " (dual nonbasic variable close to zero)\n";
            const Int jb = basis_[pmax];
            bool exchanged;
            info->errflag = basis_.ExchangeIfStable(jb, jn, pivot, -1, &exchanged);
            if (info->errflag)
                return;
            if (!exchanged)
                continue;
            invscale_basic[pmax] = 1.0 / colscale_[jn];
            info->updates_ipm++;
            basis_changes_++;
        } else {
            // the variable becomes "fixed" at its current value
            iterate->make_fixed(jn);
            basis_.FixNonbasicVariable(jn);
            colscale_[jn] = 0.0;
            info->dual_dropped++;
        }
        candidates.pop_back();
    }
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
