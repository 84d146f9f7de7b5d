// Host-side glue between the reference's solver classes and the C ABI, on plain arrays only (no
// reference types): what KKTSolverDiagHip / KKTSolverBasisHip do after they have pulled the raw
// arrays out of the reference's objects.  Kept free of reference headers so that it can be built and
// run where the reference's basis code cannot (BASICLU is not part of the reference tree):
// tests/dropin/handoff_main.cc drives exactly these functions on the golden basis fixture.
#ifndef IPX_DEVICE_GLUE_H_
#define IPX_DEVICE_GLUE_H_

#include <cstdio>
#include <new>
#include <sstream>
#include <stdexcept>
#include <string>

#include "ipx_kkt_hip.h"

namespace ipx_hip {

inline void Check(int rc) {
    if (rc == IPXK_OK) return;
    if (rc == IPXK_E_ALLOC) throw std::bad_alloc();
    const std::string msg = ipxk_last_error();
    if (rc == IPXK_E_ARGUMENT) throw std::logic_error(msg);
    throw std::runtime_error(msg);
}

// Scientific(d, 0, 2) of the reference (src/control.h:129, src/utils... Format): "%.2e"
inline std::string Sci2(double d) {
    char buf[64];
    snprintf(buf, sizeof buf, "%.2e", d);
    return buf;
}

// The message ConjugateResiduals::Solve writes to control_.Debug(3) when it stops on an error
// (reference src/conjugate_residuals.cc:53-56 plain CR, :140-152 and :198-202 preconditioned CR);
// empty when the reference prints nothing (errflag 0, 203, 205, interrupts).
inline std::string CrDebugMessage(ipxk_context* ctx, bool preconditioned) {
    ipxk_cr_diag d;
    Check(ipxk_cr_diagnostics(ctx, &d));
    std::ostringstream s;
    if (d.errflag == 201) {
        s << (preconditioned ? " PCR" : " CR") << " method not converged in " << d.maxiter << " iterations."
          << " residual = " << Sci2(d.resnorm) << ',' << " tolerance = " << Sci2(d.tol) << '\n';
    } else if (d.errflag == 202) {
        if (preconditioned)
            s << " matrix in PCR method not posdef. cdot = " << Sci2(d.cdot)
              << ", infnorm(sresidual) = " << Sci2(d.infnorm_sresidual)
              << ", infnorm(residual) = " << Sci2(d.infnorm_residual) << '\n';
        else
            s << " matrix in CR method not posdef. cdot = " << Sci2(d.cdot)
              << ", infnorm(residual) = " << Sci2(d.infnorm_residual) << '\n';
    } else if (d.errflag == 204) {
        s << " resnorm_precond_system old = " << Sci2(d.rps_old) << '\n'
          << " resnorm_precond_system new = " << Sci2(d.rps_new) << '\n';
    }
    return s.str();
}

// Hand-off of KKTSolverBasis::_Factorize (reference src/kkt_solver_basis.cc:59-64) after the basis
// maintenance: the fresh LU factors of Basis::GetLuFactors (src/basis.cc:162-166), the basis, the
// variable statuses and the interior-point column scaling go to the device.  When neither the basis
// nor its factorization changed since the previous hand-off (basis_changes() == 0 on a factorization
// that was fresh before), only the scaling-dependent part is rebuilt.
struct BasisHandoff {
    ipxint m, n;
    const ipxint *Lp, *Li; const double* Lx;
    const ipxint *Up, *Ui; const double* Ux;
    const ipxint *rowperm, *colperm, *basis;
    const ipxint* status;
    const double* colscale;
};
inline void HandOffBasis(ipxk_context* ctx, const BasisHandoff& h, bool same_factors) {
    if (same_factors)
        Check(ipxk_split_rescale(ctx, h.status, h.colscale));
    else
        Check(ipxk_split_prepare(ctx, h.Lp, h.Li, h.Lx, h.Up, h.Ui, h.Ux, h.rowperm, h.colperm, h.basis,
                                 h.status, h.colscale));
}

// What KKTSolverBasis::_Solve / KKTSolverDiag::_Solve leave in ipx::Info (src/kkt_solver_basis.cc:151-156,
// src/kkt_solver_diag.cc:100-105)
struct SolveOutcome {
    ipxint iter = 0, errflag = 0;
    ipxk_times times{};
    std::string debug3;          // text for control_.Debug(3), empty if none
};
inline SolveOutcome SolveBasisOnDevice(ipxk_context* ctx, const double* a, const double* b, double tol,
                                       ipxint maxiter, double* x, double* y, ipxk_interrupt_fn interrupt,
                                       void* interrupt_user, bool want_debug3) {
    SolveOutcome r;
    Check(ipxk_kkt_basis_solve(ctx, a, b, tol, maxiter, x, y, &r.iter, &r.errflag, interrupt, interrupt_user, &r.times));
    if (want_debug3 && r.errflag) r.debug3 = CrDebugMessage(ctx, false);
    return r;
}
inline SolveOutcome SolveDiagOnDevice(ipxk_context* ctx, const double* a, const double* b, double tol,
                                      ipxint maxiter, double* x, double* y, ipxk_interrupt_fn interrupt,
                                      void* interrupt_user, bool want_debug3) {
    SolveOutcome r;
    Check(ipxk_kkt_diag_solve(ctx, a, b, tol, maxiter, x, y, &r.iter, &r.errflag, interrupt, interrupt_user, &r.times));
    if (want_debug3 && r.errflag) r.debug3 = CrDebugMessage(ctx, true);
    return r;
}

}  // namespace ipx_hip

#endif  // IPX_DEVICE_GLUE_H_
