"""ctypes binding of the C ABI in include/ipx_kkt_hip.h (libipx_kkt_hip.so).

This is plumbing for tests and bench.py; the product is the shared library.  There
is no CPU fallback: if the library is missing or no GPU is present, construction
raises.  Names mirror the reference's classes on the path (NormalMatrix,
DiagonalPrecond, ConjugateResiduals, KKTSolverDiag, SplittedNormalMatrix,
KKTSolverBasis) -- see the file:line citations in the header.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libipx_kkt_hip.so")

i64, f64 = np.int64, np.float64
c_i64, c_f64 = C.c_int64, C.c_double
P_i64, P_f64 = C.POINTER(C.c_int64), C.POINTER(C.c_double)
INTERRUPT_FN = C.CFUNCTYPE(c_i64, C.c_void_p)

POINTER_HOST, POINTER_DEVICE = 0, 1
NONBASIC_FIXED, NONBASIC, BASIC, BASIC_FREE = -2, -1, 0, 1

EXPORTS = [
    "ipxk_last_error", "ipxk_device_count", "ipxk_create", "ipxk_destroy", "ipxk_set_pointer_mode",
    "ipxk_set_stream", "ipxk_synchronize", "ipxk_set_profiling", "ipxk_set_interrupt", "ipxk_reset_solver_state", "ipxk_get_reorder_info", "ipxk_get_reordering", "ipxk_num_dense_cols", "ipxk_get_rowwise",
    "ipxk_normal_prepare", "ipxk_normal_apply", "ipxk_diag_factorize", "ipxk_diag_apply",
    "ipxk_diag_get", "ipxk_pcr_solve", "ipxk_cr_diagnostics", "ipxk_kkt_diag_factorize", "ipxk_kkt_diag_solve",
    "ipxk_kkt_diag_get", "ipxk_split_prepare", "ipxk_split_rescale", "ipxk_split_apply", "ipxk_forward_solve",
    "ipxk_backward_solve", "ipxk_solve_dense", "ipxk_split_levels", "ipxk_cr_solve",
    "ipxk_kkt_basis_solve", "ipxk_newton_solve", "ipxk_iterate_set", "ipxk_iterate_get", "ipxk_iterate_update",
    "ipxk_iterate_residuals", "ipxk_iterate_complementarity", "ipxk_step_to_boundary", "ipxk_ipm_step", "ipxk_iterate_objectives", "ipxk_ipm_driver", "ipxk_iterate_factorize_diag", "ipxk_comm_unique_id", "ipxk_comm_init", "ipxk_comm_init_columns", "ipxk_comm_info", "ipxk_maxvolume_sequential",
    "ipxk_time_normal_apply", "ipxk_equilibrate", "ipxk_transpose", "ipxk_lu_factorize", "ipxk_lu_factorize_basis",
    "ipxk_lu_get_factors", "ipxk_lu_generation", "ipxk_split_prepare_lu", "ipxk_maxvolume", "ipxk_ipm_driver_basis",
    "ipxk_normal_apply_bytes", "ipxk_spmv_layout", "ipxk_layout_info", "ipxk_layout_array", "ipxk_split_inverse_stats", "ipxk_split_inverse_refined", "ipxk_dev_alloc", "ipxk_dev_free", "ipxk_dev_upload",
    "ipxk_dev_download",
]


class IpmStepInfo(C.Structure):
    _fields_ = [("step_primal", c_f64), ("step_dual", c_f64), ("mu_before", c_f64), ("mu_after", c_f64),
                ("sigma", c_f64), ("presidual", c_f64), ("dresidual", c_f64), ("kktiter_predictor", c_i64),
                ("kktiter_corrector", c_i64), ("errflag", c_i64)]


class CrDiag(C.Structure):
    _fields_ = [("errflag", c_i64), ("iter", c_i64), ("maxiter", c_i64), ("resnorm", c_f64), ("tol", c_f64),
                ("cdot", c_f64), ("infnorm_residual", c_f64), ("infnorm_sresidual", c_f64), ("rps_old", c_f64),
                ("rps_new", c_f64)]


class LuInfo(C.Structure):
    _fields_ = [("lnz", c_i64), ("unz", c_i64), ("num_dependent", c_i64), ("col_singletons", c_i64),
                ("row_singletons", c_i64), ("bump", c_i64), ("rounds", c_i64), ("seconds_singletons", c_f64),
                ("seconds_bump", c_f64), ("seconds_assemble", c_f64), ("spikes", c_i64), ("sparse_pivots", c_i64),
                ("sparse_rounds", c_i64), ("reused", c_i64)]


class ReorderInfo(C.Structure):
    _fields_ = [("active", c_i64), ("levels", c_i64), ("components", c_i64), ("ms", c_f64), ("us_original", c_f64), ("us_reordered", c_f64)]


class MaxvolumeParams(C.Structure):
    _fields_ = [("volume_tol", c_f64), ("maxskip_updates", c_i64), ("rows_per_slice", c_i64), ("max_etas", c_i64)]


class MaxvolumeInfo(C.Structure):
    _fields_ = [("updates", c_i64), ("skipped", c_i64), ("slices", c_i64), ("refused", c_i64), ("factorizations", c_i64),
                ("errflag", c_i64), ("volinc", c_f64), ("seconds", c_f64), ("kept_etas", c_i64)]


class IpmParams(C.Structure):
    _fields_ = [("kkt_tol", c_f64), ("feasibility_tol", c_f64), ("optimality_tol", c_f64), ("kkt_maxiter", c_i64),
                ("ipm_maxiter", c_i64), ("precond_dense_cols", C.c_int)]


class IpmInfo(C.Structure):
    _fields_ = [("status_ipm", c_i64), ("iter", c_i64), ("errflag", c_i64), ("kktiter", c_i64), ("pobjective", c_f64),
                ("dobjective", c_f64), ("presidual", c_f64), ("dresidual", c_f64), ("complementarity", c_f64),
                ("mu", c_f64), ("step_primal", c_f64), ("step_dual", c_f64), ("basis_updates", c_i64)]


class Times(C.Structure):
    _fields_ = [("cr", c_f64), ("op", c_f64), ("precond", c_f64), ("solve_B", c_f64),
                ("solve_Bt", c_f64)]


class KktError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ipxk error %d: %s" % (code, msg))
        self.code = code


def build_library():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(HERE, "csrc")])


_lib = None


def load_library():
    """Loads libipx_kkt_hip.so (raises if it has not been built: no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s is missing: build it with `make -C ipx_amd/csrc` (hipcc, gfx950)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.ipxk_last_error.restype = C.c_char_p
        L.ipxk_num_dense_cols.restype = c_i64
        L.ipxk_lu_generation.restype = c_i64
        L.ipxk_split_inverse_refined.restype = c_i64
        L.ipxk_normal_apply_bytes.restype = c_i64
        _lib = L
    return _lib


def equilibrate(A, device=0):
    """Presolver::EquilibrateMatrix on the device: (scaled values, colscale, rowscale, rounds)."""
    lib = load_library()
    x = _F(A.x).copy()
    cs, rs = np.zeros(A.ncol, f64), np.zeros(A.nrow, f64)
    rounds = c_i64(0)
    rc = lib.ipxk_equilibrate(c_i64(A.nrow), c_i64(A.ncol), _ip(_I(A.p)), _ip(_I(A.i)), _fp(x), _fp(cs), _fp(rs),
                              C.byref(rounds), C.c_int(device))
    if rc != 0:
        raise KktError(rc, lib.ipxk_last_error().decode())
    return x, cs, rs, int(rounds.value)


def transpose(A, device=0):
    """Transpose on the device: (ATp, ATi, ATx) of the row-wise copy."""
    lib = load_library()
    p, i, x = np.zeros(A.nrow + 1, i64), np.zeros(A.nnz, i64), np.zeros(A.nnz, f64)
    rc = lib.ipxk_transpose(c_i64(A.nrow), c_i64(A.ncol), _ip(_I(A.p)), _ip(_I(A.i)), _fp(_F(A.x)), _ip(p), _ip(i), _fp(x),
                            C.c_int(device))
    if rc != 0:
        raise KktError(rc, lib.ipxk_last_error().decode())
    return p, i, x


def _ip(a):
    return None if a is None else a.ctypes.data_as(P_i64)


def _fp(a):
    return None if a is None else a.ctypes.data_as(P_f64)


def _I(a):
    return None if a is None else np.ascontiguousarray(a, dtype=i64)


def _F(a):
    return None if a is None else np.ascontiguousarray(a, dtype=f64)


class DeviceVector:
    """fp64 vector resident in HBM (hipMalloc through the C ABI)."""

    def __init__(self, ctx, n, host=None):
        self.ctx, self.n = ctx, int(n)
        p = C.c_void_p()
        ctx._check(ctx.lib.ipxk_dev_alloc(ctx.h, c_i64(8 * self.n), C.byref(p)))
        self.ptr = p
        if host is not None:
            self.upload(host)

    def upload(self, host):
        host = _F(host)
        assert host.size == self.n
        self.ctx._check(self.ctx.lib.ipxk_dev_upload(self.ctx.h, self.ptr, _fp(host),
                                                     c_i64(8 * self.n)))

    def download(self):
        out = np.empty(self.n, f64)
        self.ctx._check(self.ctx.lib.ipxk_dev_download(self.ctx.h, _fp(out), self.ptr,
                                                       c_i64(8 * self.n)))
        return out

    def as_arg(self):
        return C.cast(self.ptr, P_f64)

    def free(self):
        if self.ptr:
            self.ctx.lib.ipxk_dev_free(self.ctx.h, self.ptr)
            self.ptr = None


class KktContext:
    """One model's constraint matrix resident on one MI355X.

    A: object with .nrow, .ncol, .p, .i, .x (int64 CSC of the n structural columns).
    """

    def __init__(self, A, device=0):
        self.lib = load_library()
        self.m, self.n = int(A.nrow), int(A.ncol)
        self._keep = (_I(A.p), _I(A.i), _F(A.x))
        h = C.c_void_p()
        rc = self.lib.ipxk_create(c_i64(self.m), c_i64(self.n), _ip(self._keep[0]),
                                  _ip(self._keep[1]), _fp(self._keep[2]), C.c_int(device),
                                  C.byref(h))
        if rc != 0:
            raise KktError(rc, self.lib.ipxk_last_error().decode())
        self.h = h
        self.device_mode = False

    def _check(self, rc):
        if rc != 0:
            raise KktError(rc, self.lib.ipxk_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.ipxk_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # -- plumbing ---------------------------------------------------------------
    def set_pointer_mode(self, device):
        self._check(self.lib.ipxk_set_pointer_mode(self.h, C.c_int(1 if device else 0)))
        self.device_mode = bool(device)

    def set_stream(self, stream_handle):
        self._check(self.lib.ipxk_set_stream(self.h, C.c_void_p(stream_handle)))

    def set_profiling(self, on):
        self._check(self.lib.ipxk_set_profiling(self.h, C.c_int(1 if on else 0)))

    def set_interrupt(self, interrupt):
        """Control::InterruptCheck for maxvolume / maxvolume_sequential (polled once per candidate); None removes it"""
        self._interrupt_cb = INTERRUPT_FN(lambda _u: int(interrupt())) if interrupt else C.cast(None, INTERRUPT_FN)
        self._check(self.lib.ipxk_set_interrupt(self.h, self._interrupt_cb, None))

    def synchronize(self):
        self._check(self.lib.ipxk_synchronize(self.h))

    def reorder_info(self):
        """the locality-recovering renumbering ipxk_create found (or not): dict(active, levels, components, ms, us_original, us_reordered)"""
        info = ReorderInfo()
        self._check(self.lib.ipxk_get_reorder_info(self.h, C.byref(info)))
        return {name: getattr(info, name) for name, _ in ReorderInfo._fields_}

    def reordering(self):
        """(rowperm, colperm): new index -> index as given"""
        r, q = np.zeros(self.m, i64), np.zeros(self.n, i64)
        self._check(self.lib.ipxk_get_reordering(self.h, _ip(r), _ip(q)))
        return r, q

    def reset_solver_state(self, lu_pivottol=0.0):
        """the context as a new solver object finds it (HipModel hands a cached context out through this): nothing prepared /
        factorized, Maxvolume's refactorizations start from lu_pivottol (<= 0: 0.1)"""
        self._check(self.lib.ipxk_reset_solver_state(self.h, c_f64(lu_pivottol)))

    def vector(self, n, host=None):
        return DeviceVector(self, n, host)

    @property
    def num_dense_cols(self):
        return int(self.lib.ipxk_num_dense_cols(self.h))

    @property
    def normal_apply_bytes(self):
        return int(self.lib.ipxk_normal_apply_bytes(self.h))

    def comm_info(self):
        """(transport, nranks, rank) as the transport reports them: 'none' | 'rccl' (ncclCommCount / ncclCommUserRank) | 'direct'"""
        t, n, r = C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.ipxk_comm_info(self.h, C.byref(t), C.byref(n), C.byref(r)))
        return ("none", "rccl", "direct")[t.value], n.value, r.value

    def spmv_layout(self):
        """(layout of A'y, layout of A t) as 'phased'/'sliced', and the build-time timings in us."""
        lay = (C.c_int * 2)()
        us = (C.c_double * 6)()
        self._check(self.lib.ipxk_spmv_layout(self.h, lay, us))
        names = ("phased", "sliced", "fused", "sorted", "sortedfused", "acc", "plain", "accfused")
        return (names[lay[0]], names[lay[1]]), [float(v) for v in us]

    def split_inverse_stats(self):
        """(# explicit inverses probed, # rejected, worst probe residual) since the context was created."""
        a, b, w = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
        self._check(self.lib.ipxk_split_inverse_stats(self.h, C.byref(a), C.byref(b), C.byref(w)))
        return a.value, b.value, w.value

    def split_inverse_refined(self):
        """# refinement steps of dense-block inverses since the context was created"""
        return int(self.lib.ipxk_split_inverse_refined(self.h))

    def layout_info(self, which):
        """Scalars of the device layouts of gather matrix `which` (0: A'y, 1: A t) and the create timings (ms)."""
        info = (C.c_int64 * 40)()
        ms = (C.c_double * 4)()
        self._check(self.lib.ipxk_layout_info(self.h, which, info, ms))
        keys = ("use_sliced", "use_sorted", "use_sorted_fused", "nlong", "sliced_built", "R", "nslices", "nrb", "nrows_pad",
                "max_tile", "dominant_bits", "sorted_built", "so_nslices", "so_nsub", "so_nrb", "so_RB", "so_nrows_pad",
                "so_max_sub", "so_slice_elems", "so_fused", "nnz", "P", "G", "RTQ", "use_acc", "acc_built", "acc_nslices", "acc_nrb",
                "acc_RB", "acc_nrows_pad", "acc_slice_elems", "acc_nbatches", "acc_deferred", "use_acc_fused", "accf_built", "accf_nrb", "accf_RB",
                "accf_nbatches", "use_plain")
        d = {k: int(v) for k, v in zip(keys, info)}
        d["dominant_fraction"] = float(np.array([d.pop("dominant_bits")], dtype=np.int64).view(np.float64)[0])
        return d, [float(v) for v in ms]

    def layout_array(self, which, array):
        """One array of the device layouts as numpy (see ipxk_layout_array)."""
        dt = (np.uint32, np.uint8, np.int32, np.float64, np.uint32, np.uint8, np.uint32, np.float64, np.int32, np.int32, np.float64,
              np.uint32, np.uint32, np.uint32, np.float64, np.uint32, np.uint32, np.uint32, np.float64, np.int32, np.int32)[array]
        nb = C.c_int64(0)
        self._check(self.lib.ipxk_layout_array(self.h, which, array, None, 0, C.byref(nb)))
        out = np.zeros(nb.value // np.dtype(dt).itemsize, dtype=dt)
        if nb.value:
            self._check(self.lib.ipxk_layout_array(self.h, which, array, out.ctypes.data_as(C.c_void_p), nb.value, C.byref(nb)))
            self._check(self.lib.ipxk_synchronize(self.h))
        return out

    def get_rowwise(self):
        nnz = int(self._keep[0][-1])
        p, i, x = np.zeros(self.m + 1, i64), np.zeros(nnz, i64), np.zeros(nnz, f64)
        self._check(self.lib.ipxk_get_rowwise(self.h, _ip(p), _ip(i), _fp(x)))
        return p, i, x

    # -- NormalMatrix -------------------------------------------------------------
    def normal_prepare(self, W):
        self._W = _F(W)
        self._check(self.lib.ipxk_normal_prepare(self.h, _fp(self._W)))

    def normal_apply(self, rhs, want_dot=True):
        rhs = _F(rhs)
        lhs = np.zeros(self.m, f64)
        dot = c_f64(0.0)
        self._check(self.lib.ipxk_normal_apply(self.h, _fp(rhs), _fp(lhs),
                                               C.byref(dot) if want_dot else None))
        return lhs, dot.value

    def time_normal_apply(self, rhs_dev, lhs_dev, reps):
        ms = c_f64(0.0)
        self._check(self.lib.ipxk_time_normal_apply(self.h, rhs_dev.as_arg(), lhs_dev.as_arg(),
                                                    C.c_int(reps), C.byref(ms)))
        return ms.value

    # -- DiagonalPrecond ------------------------------------------------------------
    def diag_factorize(self, W, precond_dense_cols=True):
        err = c_i64(0)
        self._check(self.lib.ipxk_diag_factorize(self.h, _fp(_F(W)),
                                                 C.c_int(1 if precond_dense_cols else 0),
                                                 C.byref(err)))
        return int(err.value)

    def diag_apply(self, rhs):
        rhs = _F(rhs)
        lhs = np.zeros(self.m, f64)
        dot = c_f64(0.0)
        self._check(self.lib.ipxk_diag_apply(self.h, _fp(rhs), _fp(lhs), C.byref(dot)))
        return lhs, dot.value

    def diag_get(self, k=0):
        d = np.zeros(self.m, f64)
        ch = np.zeros(k * k, f64)
        self._check(self.lib.ipxk_diag_get(self.h, _fp(d), _fp(ch) if k else None))
        return d, ch.reshape(k, k).T

    # -- ConjugateResiduals -----------------------------------------------------------
    def _cr(self, fn, rhs, tol, resscale, maxiter, lhs0, hist_cap, interrupt):
        rhs = _F(rhs)
        lhs = np.zeros(self.m, f64) if lhs0 is None else _F(lhs0).copy()
        it, err = c_i64(0), c_i64(0)
        hist = np.full(max(hist_cap, 1), np.nan, f64)
        times = Times()
        cb = INTERRUPT_FN(lambda _u: int(interrupt())) if interrupt else C.cast(None, INTERRUPT_FN)
        self._check(fn(self.h, _fp(rhs), c_f64(tol), _fp(_F(resscale)), c_i64(maxiter), _fp(lhs),
                       C.byref(it), C.byref(err), cb, None, _fp(hist) if hist_cap else None,
                       c_i64(hist_cap), C.byref(times)))
        hist = hist[:hist_cap]
        return lhs, int(it.value), int(err.value), hist[~np.isnan(hist)], times

    def pcr_solve(self, rhs, tol, resscale, maxiter, lhs0=None, hist_cap=0, interrupt=None):
        return self._cr(self.lib.ipxk_pcr_solve, rhs, tol, resscale, maxiter, lhs0, hist_cap, interrupt)

    def cr_diagnostics(self):
        d = CrDiag()
        self._check(self.lib.ipxk_cr_diagnostics(self.h, C.byref(d)))
        return {k: getattr(d, k) for k, _ in CrDiag._fields_}

    def cr_solve(self, rhs, tol, resscale, maxiter, lhs0=None, hist_cap=0, interrupt=None):
        return self._cr(self.lib.ipxk_cr_solve, rhs, tol, resscale, maxiter, lhs0, hist_cap, interrupt)

    # -- KKTSolverDiag ------------------------------------------------------------------
    def kkt_diag_factorize(self, xl=None, xu=None, zl=None, zu=None, mu=0.0,
                           precond_dense_cols=True):
        err = c_i64(0)
        self._check(self.lib.ipxk_kkt_diag_factorize(
            self.h, _fp(_F(xl)), _fp(_F(xu)), _fp(_F(zl)), _fp(_F(zu)), c_f64(mu),
            C.c_int(1 if precond_dense_cols else 0), C.byref(err)))
        return int(err.value)

    def kkt_diag_solve(self, a, b, tol, maxiter=-1, interrupt=None):
        """Host vectors in/out (the reference's Vector boundary)."""
        a, b = _F(a), _F(b)
        x, y = np.zeros(self.n + self.m, f64), np.zeros(self.m, f64)
        it, err, times = c_i64(0), c_i64(0), Times()
        cb = INTERRUPT_FN(lambda _u: int(interrupt())) if interrupt else C.cast(None, INTERRUPT_FN)
        self._check(self.lib.ipxk_kkt_diag_solve(self.h, _fp(a), _fp(b), c_f64(tol), c_i64(maxiter),
                                                 _fp(x), _fp(y), C.byref(it), C.byref(err), cb, None,
                                                 C.byref(times)))
        return x, y, int(it.value), int(err.value), times

    def kkt_diag_solve_resident(self, a_dev, b_dev, x_dev, y_dev, tol, maxiter=-1):
        """Device-resident vectors (pointer mode must be device): what bench.py times."""
        it, err, times = c_i64(0), c_i64(0), Times()
        self._check(self.lib.ipxk_kkt_diag_solve(self.h, a_dev.as_arg(), b_dev.as_arg(), c_f64(tol),
                                                 c_i64(maxiter), x_dev.as_arg(), y_dev.as_arg(),
                                                 C.byref(it), C.byref(err),
                                                 C.cast(None, INTERRUPT_FN), None, C.byref(times)))
        return int(it.value), int(err.value), times

    # -- IPM::SolveNewtonSystem ------------------------------------------------------------
    STATE_FIXED, STATE_FREE, STATE_BARRIER_LB, STATE_BARRIER_UB, STATE_BARRIER_BOXED = range(5)

    def newton_solve(self, use_basis, rb, rc, rl, ru, sl, su, xl, xu, zl, zu, state, tol, maxiter=-1):
        """Host vectors; rb, rc, rl, ru may be None (zero).  Returns a dict with the six step components."""
        N = self.n + self.m
        ins = [_F(v) for v in (rb, rc, rl, ru, sl, su, xl, xu, zl, zu)]
        st = np.ascontiguousarray(state, dtype=np.uint8)
        assert st.shape == (N,)
        out = {k: np.zeros(self.m if k == "dy" else N, f64) for k in ("dx", "dxl", "dxu", "dy", "dzl", "dzu")}
        it, err, times = c_i64(0), c_i64(0), Times()
        self._check(self.lib.ipxk_newton_solve(
            self.h, C.c_int(1 if use_basis else 0), *[_fp(v) for v in ins],
            st.ctypes.data_as(C.POINTER(C.c_ubyte)), c_f64(tol), c_i64(maxiter),
            _fp(out["dx"]), _fp(out["dxl"]), _fp(out["dxu"]), _fp(out["dy"]), _fp(out["dzl"]), _fp(out["dzu"]),
            C.byref(it), C.byref(err), C.cast(None, INTERRUPT_FN), None, C.byref(times)))
        out.update(iter=int(it.value), errflag=int(err.value), times=times)
        return out

    def newton_solve_resident(self, use_basis, dev_in, state_dev, tol, maxiter, dev_out):
        """Device-resident call (pointer mode must be device).  dev_in: 10 DeviceVectors or None in the order
        rb, rc, rl, ru, sl, su, xl, xu, zl, zu; state_dev: DeviceVector holding the state BYTES (see
        state_vector); dev_out: dx, dxl, dxu, dy, dzl, dzu."""
        it, err, times = c_i64(0), c_i64(0), Times()
        self._check(self.lib.ipxk_newton_solve(
            self.h, C.c_int(1 if use_basis else 0), *[(v.as_arg() if v is not None else None) for v in dev_in],
            C.cast(state_dev.ptr, C.POINTER(C.c_ubyte)), c_f64(tol), c_i64(maxiter),
            *[v.as_arg() for v in dev_out], C.byref(it), C.byref(err), C.cast(None, INTERRUPT_FN), None,
            C.byref(times)))
        return int(it.value), int(err.value), times

    def state_vector(self, state):
        """Uploads one state byte per variable into a device buffer (padded to whole doubles)."""
        st = np.ascontiguousarray(state, dtype=np.uint8)
        padded = np.zeros((st.size + 7) // 8 * 8, np.uint8)
        padded[:st.size] = st
        return DeviceVector(self, padded.size // 8, padded.view(f64))

    # -- the IPM iterate (host vectors; pointer mode host) -----------------------------------
    IT_KEYS = ("x", "xl", "xu", "y", "zl", "zu")

    def iterate_set(self, it, state):
        st = np.ascontiguousarray(state, dtype=np.uint8)
        vecs = [_F(it[key]) for key in self.IT_KEYS]
        self._check(self.lib.ipxk_iterate_set(self.h, *[_fp(v) for v in vecs],
                                              st.ctypes.data_as(C.POINTER(C.c_ubyte))))

    def iterate_get(self):
        N = self.n + self.m
        out = {key: np.zeros(self.m if key == "y" else N, f64) for key in self.IT_KEYS}
        self._check(self.lib.ipxk_iterate_get(self.h, *[_fp(out[key]) for key in self.IT_KEYS]))
        return out

    def iterate_update(self, sp, dx, dxl, dxu, sd, dy, dzl, dzu):
        self._check(self.lib.ipxk_iterate_update(self.h, c_f64(sp), _fp(_F(dx)), _fp(_F(dxl)), _fp(_F(dxu)),
                                                 c_f64(sd), _fp(_F(dy)), _fp(_F(dzl)), _fp(_F(dzu))))

    def iterate_residuals(self, b, c, lb, ub):
        N = self.n + self.m
        rb, rc, rl, ru = np.zeros(self.m, f64), np.zeros(N, f64), np.zeros(N, f64), np.zeros(N, f64)
        pres, dres = c_f64(0.0), c_f64(0.0)
        self._check(self.lib.ipxk_iterate_residuals(self.h, _fp(_F(b)), _fp(_F(c)), _fp(_F(lb)), _fp(_F(ub)),
                                                    _fp(rb), _fp(rc), _fp(rl), _fp(ru), C.byref(pres),
                                                    C.byref(dres)))
        return dict(rb=rb, rc=rc, rl=rl, ru=ru, presidual=pres.value, dresidual=dres.value)

    def iterate_complementarity(self):
        out = (C.c_double * 4)()
        self._check(self.lib.ipxk_iterate_complementarity(self.h, out))
        return dict(complementarity=out[0], mu=out[1], mu_min=out[2], mu_max=out[3])

    def step_to_boundary(self, x, dx, alpha0=1.0):
        x, dx = _F(x), _F(dx)
        alpha, blk = c_f64(0.0), c_i64(-1)
        self._check(self.lib.ipxk_step_to_boundary(self.h, _fp(x), _fp(dx), c_i64(x.size), c_f64(alpha0),
                                                   C.byref(alpha), C.byref(blk)))
        return alpha.value, int(blk.value)

    def iterate_factorize_diag(self, precond_dense_cols=True):
        err = c_i64(0)
        self._check(self.lib.ipxk_iterate_factorize_diag(self.h, C.c_int(1 if precond_dense_cols else 0),
                                                         C.byref(err)))
        return int(err.value)

    def ipm_step(self, use_basis, b, c, lb, ub, kkt_tol=0.3, maxiter=-1):
        """One predictor-corrector step on the resident iterate (host model vectors)."""
        info = IpmStepInfo()
        self._check(self.lib.ipxk_ipm_step(self.h, C.c_int(1 if use_basis else 0), _fp(_F(b)), _fp(_F(c)),
                                           _fp(_F(lb)), _fp(_F(ub)), c_f64(kkt_tol), c_i64(maxiter),
                                           C.byref(info), C.cast(None, INTERRUPT_FN), None))
        return {name: getattr(info, name) for name, _ in IpmStepInfo._fields_}

    def ipm_step_resident(self, use_basis, b_dev, c_dev, lb_dev, ub_dev, kkt_tol=0.3, maxiter=-1):
        info = IpmStepInfo()
        self._check(self.lib.ipxk_ipm_step(self.h, C.c_int(1 if use_basis else 0), b_dev.as_arg(), c_dev.as_arg(),
                                           lb_dev.as_arg(), ub_dev.as_arg(), c_f64(kkt_tol), c_i64(maxiter),
                                           C.byref(info), C.cast(None, INTERRUPT_FN), None))
        return {name: getattr(info, name) for name, _ in IpmStepInfo._fields_}

    def iterate_objectives(self, b, c, lb, ub):
        """Iterate::ComputeObjectives of the resident iterate: (pobjective, dobjective, offset)"""
        out = (C.c_double * 3)()
        self._check(self.lib.ipxk_iterate_objectives(self.h, _fp(_F(b)), _fp(_F(c)), _fp(_F(lb)), _fp(_F(ub)), out))
        return out[0], out[1], out[2]

    def ipm_driver(self, b, c, lb, ub, kkt_tol=0.3, feasibility_tol=1e-6, optimality_tol=1e-8, kkt_maxiter=-1,
                   ipm_maxiter=300, precond_dense_cols=True, interrupt=None):
        """IPM::Driver on the resident iterate with the diag solver (host model vectors)."""
        prm = IpmParams(kkt_tol, feasibility_tol, optimality_tol, kkt_maxiter, ipm_maxiter, 1 if precond_dense_cols else 0)
        info = IpmInfo()
        cb = INTERRUPT_FN(lambda _u: int(interrupt())) if interrupt else C.cast(None, INTERRUPT_FN)
        self._check(self.lib.ipxk_ipm_driver(self.h, _fp(_F(b)), _fp(_F(c)), _fp(_F(lb)), _fp(_F(ub)), C.byref(prm),
                                             C.byref(info), cb, None))
        return {name: getattr(info, name) for name, _ in IpmInfo._fields_}

    def ipm_driver_basis(self, b, c, lb, ub, kkt_tol=0.3, feasibility_tol=1e-6, optimality_tol=1e-8, ipm_maxiter=300,
                         interrupt=None):
        """IPM::Driver on the resident iterate with the basis solver (LpSolver::RunMainIPM); returns the info fields
        plus the final basis and statuses"""
        prm = IpmParams(kkt_tol, feasibility_tol, optimality_tol, -1, ipm_maxiter, 1)
        info = IpmInfo()
        basis, status = np.zeros(self.m, i64), np.zeros(self.n + self.m, i64)
        cb = INTERRUPT_FN(lambda _u: int(interrupt())) if interrupt else C.cast(None, INTERRUPT_FN)
        self._check(self.lib.ipxk_ipm_driver_basis(self.h, _fp(_F(b)), _fp(_F(c)), _fp(_F(lb)), _fp(_F(ub)), C.byref(prm),
                                                   C.byref(info), _ip(basis), _ip(status), cb, None))
        out = {name: getattr(info, name) for name, _ in IpmInfo._fields_}
        out.update(basis=basis, status=status)
        return out

    def kkt_diag_get(self):
        W, rs = np.zeros(self.n + self.m, f64), np.zeros(self.m, f64)
        self._check(self.lib.ipxk_kkt_diag_get(self.h, _fp(W), _fp(rs)))
        return W, rs

    # -- SplittedNormalMatrix / KKTSolverBasis -------------------------------------------
    def split_prepare(self, L, U, rowperm, colperm, basis, status, colscale):
        args = [_I(L.p), _I(L.i), _F(L.x), _I(U.p), _I(U.i), _F(U.x), _I(rowperm), _I(colperm),
                _I(basis), _I(status), _F(colscale)]
        self._check(self.lib.ipxk_split_prepare(
            self.h, _ip(args[0]), _ip(args[1]), _fp(args[2]), _ip(args[3]), _ip(args[4]),
            _fp(args[5]), _ip(args[6]), _ip(args[7]), _ip(args[8]), _ip(args[9]), _fp(args[10])))

    # ---- LU factorization (LuFactorization contract, reference src/lu_factorization.h:21-58) ----
    def _lu_result(self, dim, info, download):
        out = {name: getattr(info, name) for name, _ in LuInfo._fields_}
        if not download:
            return out
        Lp, Up = np.zeros(dim + 1, i64), np.zeros(dim + 1, i64)
        Li, Lx = np.zeros(info.lnz, i64), np.zeros(info.lnz, f64)
        Ui, Ux = np.zeros(info.unz, i64), np.zeros(info.unz, f64)
        rowperm, colperm, dep = np.zeros(dim, i64), np.zeros(dim, i64), np.zeros(info.num_dependent, i64)
        self._check(self.lib.ipxk_lu_get_factors(self.h, _ip(Lp), _ip(Li), _fp(Lx), _ip(Up), _ip(Ui), _fp(Ux),
                                                 _ip(rowperm), _ip(colperm), _ip(dep)))
        from .synth import CscMatrix
        out.update(L=CscMatrix(dim, dim, Lp, Li, Lx), U=CscMatrix(dim, dim, Up, Ui, Ux), rowperm=rowperm,
                   colperm=colperm, dependent=dep)
        return out

    def lu_factorize(self, dim, Bbegin, Bend, Bi, Bx, pivottol=0.1, strict=False, download=True):
        Bbegin, Bend, Bi, Bx = _I(Bbegin), _I(Bend), _I(Bi), _F(Bx)
        info = LuInfo()
        self._check(self.lib.ipxk_lu_factorize(self.h, c_i64(dim), _ip(Bbegin), _ip(Bend), _ip(Bi), _fp(Bx),
                                               c_f64(pivottol), C.c_int(1 if strict else 0), C.byref(info)))
        return self._lu_result(dim, info, download)

    def lu_factorize_basis(self, basis, pivottol=0.1, strict=False, download=True):
        basis = _I(basis)
        assert basis.size == self.m
        info = LuInfo()
        self._check(self.lib.ipxk_lu_factorize_basis(self.h, _ip(basis), c_f64(pivottol), C.c_int(1 if strict else 0),
                                                     C.byref(info)))
        return self._lu_result(self.m, info, download)

    def lu_generation(self):
        """number of LU factorizations this context has computed (a reused one does not count)"""
        return int(self.lib.ipxk_lu_generation(self.h))

    def split_prepare_lu(self, status, colscale):
        status, colscale = _I(status), _F(colscale)
        self._check(self.lib.ipxk_split_prepare_lu(self.h, _ip(status), _fp(colscale)))

    def maxvolume(self, status, colscale, volume_tol=2.0, maxskip_updates=10, rows_per_slice=10000, max_etas=100,
                  log_cap=100000):
        """Maxvolume::RunHeuristic + refactorization + Prepare on the resident basis; returns dict(basis, status,
        exchanges, info fields)"""
        status, colscale = _I(status), _F(colscale)
        prm = MaxvolumeParams(volume_tol, maxskip_updates, rows_per_slice, max_etas)
        info = MaxvolumeInfo()
        basis_out, status_out = np.zeros(self.m, i64), np.zeros(self.n + self.m, i64)
        log = np.zeros(2 * max(log_cap, 1), i64)
        self._check(self.lib.ipxk_maxvolume(self.h, _ip(status), _fp(colscale), C.byref(prm), _ip(basis_out), _ip(status_out),
                                            C.byref(info), _ip(log), c_i64(log_cap)))
        out = {name: getattr(info, name) for name, _ in MaxvolumeInfo._fields_}
        out.update(basis=basis_out, status=status_out, exchanges=log[: 2 * min(info.updates, log_cap)].reshape(-1, 2))
        return out

    def maxvolume_sequential(self, status, colscale, volume_tol=2.0, maxpasses=-1, max_etas=100, log_cap=100000):
        """Maxvolume::RunSequential (update_heuristic == 0) + refactorization + Prepare on the resident basis"""
        status, colscale = _I(status), _F(colscale)
        info = MaxvolumeInfo()
        basis_out, status_out = np.zeros(self.m, i64), np.zeros(self.n + self.m, i64)
        log = np.zeros(2 * max(log_cap, 1), i64)
        self._check(self.lib.ipxk_maxvolume_sequential(self.h, _ip(status), _fp(colscale), c_f64(volume_tol), c_i64(maxpasses),
                                                       c_i64(max_etas), _ip(basis_out), _ip(status_out), C.byref(info), _ip(log),
                                                       c_i64(log_cap)))
        out = {name: getattr(info, name) for name, _ in MaxvolumeInfo._fields_}
        out["passes"] = out["slices"]
        out.update(basis=basis_out, status=status_out, exchanges=log[: 2 * min(info.updates, log_cap)].reshape(-1, 2))
        return out

    def split_rescale(self, status, colscale):
        status, colscale = _I(status), _F(colscale)
        self._check(self.lib.ipxk_split_rescale(self.h, _ip(status), _fp(colscale)))

    def split_apply(self, rhs, want_dot=True):
        rhs = _F(rhs)
        lhs = np.zeros(self.m, f64)
        dot = c_f64(0.0)
        self._check(self.lib.ipxk_split_apply(self.h, _fp(rhs), _fp(lhs),
                                              C.byref(dot) if want_dot else None))
        return lhs, dot.value

    def forward_solve(self, x):
        x = _F(x).copy()
        self._check(self.lib.ipxk_forward_solve(self.h, _fp(x)))
        return x

    def backward_solve(self, x):
        x = _F(x).copy()
        self._check(self.lib.ipxk_backward_solve(self.h, _fp(x)))
        return x

    def solve_dense(self, rhs, trans):
        rhs = _F(rhs)
        lhs = np.zeros(self.m, f64)
        self._check(self.lib.ipxk_solve_dense(self.h, _fp(rhs), _fp(lhs), C.c_char(trans.encode())))
        return lhs

    def split_levels(self):
        lv = np.zeros(4, i64)
        self._check(self.lib.ipxk_split_levels(self.h, _ip(lv)))
        return [int(v) for v in lv]

    def kkt_basis_solve(self, a, b, tol, maxiter=-1):
        a, b = _F(a), _F(b)
        x, y = np.zeros(self.n + self.m, f64), np.zeros(self.m, f64)
        it, err, times = c_i64(0), c_i64(0), Times()
        self._check(self.lib.ipxk_kkt_basis_solve(self.h, _fp(a), _fp(b), c_f64(tol), c_i64(maxiter),
                                                  _fp(x), _fp(y), C.byref(it), C.byref(err),
                                                  C.cast(None, INTERRUPT_FN), None, C.byref(times)))
        return x, y, int(it.value), int(err.value), times

    def kkt_basis_solve_resident(self, a_dev, b_dev, x_dev, y_dev, tol, maxiter=-1):
        it, err, times = c_i64(0), c_i64(0), Times()
        self._check(self.lib.ipxk_kkt_basis_solve(self.h, a_dev.as_arg(), b_dev.as_arg(), c_f64(tol),
                                                  c_i64(maxiter), x_dev.as_arg(), y_dev.as_arg(),
                                                  C.byref(it), C.byref(err),
                                                  C.cast(None, INTERRUPT_FN), None, C.byref(times)))
        return int(it.value), int(err.value), times

    # -- multi-GPU -------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = (C.c_char * 128)()
        self._check(self.lib.ipxk_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, nranks, columns=False):
        """columns=False: the context holds this rank's rows; True: its structural columns."""
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        fn = self.lib.ipxk_comm_init_columns if columns else self.lib.ipxk_comm_init
        self._check(fn(self.h, buf, C.c_int(rank), C.c_int(nranks)))
