// TEST HARNESS ONLY -- pins NOTHING for parity and is never part of oracle/_ref/libipx_ref.so.
//
// BASICLU (the reference's LU library, install.txt:1-3) is not in this image, so the two reference
// units that wrap it (src/basiclu_kernel.cc, src/basiclu_wrapper.cc) cannot be compiled, and with them
// ipx::Basis -- hence ipx::KKTSolverBasis, IPM's main phase and LpSolver -- could never be linked.
// This file supplies the symbols those two units would have defined, so that the rest of the reference
// links and the PRODUCT can be run under it:
//
//   * ipx::BasicLuKernel::_Factorize (src/basiclu_kernel.h:10-18) forwards to ipx::LuKernelHip, i.e. it
//     performs the one-line change of src/basis.cc:27 that INTEGRATION.md describes.  Every LU
//     factorization of a run built with this file is therefore computed on the MI355X by lu.hip; the
//     reference's ForrestTomlin / LuFactorization (its own code) judge and update those factors.
//   * ipx::BasicLu (src/basiclu_wrapper.h:11-47) is only constructed for lu_kernel <= 0
//     (src/basis.cc:24-25); the programs built with this file set lu_kernel = 1, and every member
//     throws, so a run can never silently use it.
//
// No arithmetic of BASICLU is imitated here.
#include <memory>
#include <stdexcept>

#include "basiclu_kernel.h"
#include "basiclu_wrapper.h"
#include "device_glue.h"
#include "lu_kernel_hip.h"

namespace {

[[noreturn]] void Absent() {
    throw std::logic_error("BASICLU is not part of this build: run with parameters.lu_kernel = 1");
}

// one process-wide context that only names the device (the LU entry points take their matrix as arguments)
ipxk_context* LuContext() {
    static ipxk_context* ctx = [] {
        const ipxint p[2] = {0, 1}, i[1] = {0};
        const double x[1] = {1.0};
        ipxk_context* c = nullptr;
        ipx_hip::Check(ipxk_create(1, 1, p, i, x, 0, &c));
        return c;
    }();
    return ctx;
}

long g_factorizations = 0;
long g_reused = 0;
long g_max_bump = 0;
double g_seconds = 0.0;

}  // namespace

// counters for the test programs (tests/dropin/lp_main.cc)
extern "C" long dropin_lu_factorizations() { return g_factorizations; }
extern "C" long dropin_lu_reused() { return g_reused; }
extern "C" long dropin_lu_max_bump() { return g_max_bump; }
extern "C" double dropin_lu_seconds() { return g_seconds; }

namespace ipx {

void BasicLuKernel::_Factorize(Int dim, const Int* Bbegin, const Int* Bend, const Int* Bi, const double* Bx,
                               double pivottol, bool strict_abs_pivottol, SparseMatrix* L, SparseMatrix* U,
                               std::vector<Int>* rowperm, std::vector<Int>* colperm,
                               std::vector<Int>* dependent_cols) {
    // no fallback: a basis the device LU declines ends the run loudly.  A factorization requested while a KKT solver object is at
    // work goes through that object's context (LuKernelHip::SharedWithSolver), where the resident factors of the same basis are
    // handed out again.
    LuKernelHip lu(LuKernelHip::SharedWithSolver{}, LuContext());
    lu.Factorize(dim, Bbegin, Bend, Bi, Bx, pivottol, strict_abs_pivottol, L, U, rowperm, colperm, dependent_cols);
    g_factorizations++;
    g_reused += lu.reused();
    if (lu.info().bump > g_max_bump) g_max_bump = lu.info().bump;
    g_seconds += lu.info().seconds_singletons + lu.info().seconds_bump + lu.info().seconds_assemble;
}

BasicLu::BasicLu(const Control& control, Int dim) : control_(control), dim_(dim) { Absent(); }
Int BasicLu::_Factorize(const Int*, const Int*, const Int*, const double*, bool) { Absent(); }
void BasicLu::_GetFactors(SparseMatrix*, SparseMatrix*, Int*, Int*, std::vector<Int>*) { Absent(); }
void BasicLu::_SolveDense(const Vector&, Vector&, char) { Absent(); }
void BasicLu::_FtranForUpdate(Int, const Int*, const double*) { Absent(); }
void BasicLu::_FtranForUpdate(Int, const Int*, const double*, IndexedVector&) { Absent(); }
void BasicLu::_BtranForUpdate(Int) { Absent(); }
void BasicLu::_BtranForUpdate(Int, IndexedVector&) { Absent(); }
Int BasicLu::_Update(double) { Absent(); }
bool BasicLu::_NeedFreshFactorization() { Absent(); }
double BasicLu::_fill_factor() const { Absent(); }
double BasicLu::_pivottol() const { Absent(); }
void BasicLu::_pivottol(double) { Absent(); }
void BasicLu::Reallocate() { Absent(); }

}  // namespace ipx
