"""TEST INFRASTRUCTURE ONLY: ctypes bindings for the CPU oracle
(oracle/libipx_oracle.so, this repo's restatement) and, when it has been built,
for the reference's own objects (oracle/_ref/libipx_ref.so).

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package ipx_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libipx_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libipx_ref.so")

i64 = np.int64
f64 = np.float64
c_i64 = C.c_int64
c_f64 = C.c_double
P_i64 = C.POINTER(C.c_int64)
P_f64 = C.POINTER(C.c_double)
APPLY_FN = C.CFUNCTYPE(None, C.c_void_p, P_f64, P_f64, P_f64)

NONBASIC_FIXED, NONBASIC, BASIC, BASIC_FREE = -2, -1, 0, 1


def build(ref=True):
    """Compile the oracle (and the reference build when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _ip(a):
    if a is None:
        return None
    assert a.dtype == i64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(P_i64)


def _fp(a):
    if a is None:
        return None
    assert a.dtype == f64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(P_f64)


def _I(a):
    return None if a is None else np.ascontiguousarray(a, dtype=i64)


def _F(a):
    return None if a is None else np.ascontiguousarray(a, dtype=f64)


class Csc:
    """CSC matrix with int64 indices (the reference's SparseMatrix layout)."""

    def __init__(self, nrow, ncol, p, i, x):
        self.nrow, self.ncol = int(nrow), int(ncol)
        self.p, self.i, self.x = _I(p), _I(i), _F(x)
        assert self.p.shape == (self.ncol + 1,)

    @property
    def nnz(self):
        return int(self.p[-1])

    @staticmethod
    def from_scipy(A):
        A = A.tocsc()
        A.sort_indices()
        return Csc(A.shape[0], A.shape[1], A.indptr, A.indices, A.data)

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.x, self.i, self.p), shape=(self.nrow, self.ncol))


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        self.lib = L = C.CDLL(path)
        L.orc_dot.restype = c_f64
        L.orc_infnorm.restype = c_f64
        for name in ("orc_find_dense_columns", "orc_diag_num_dense", "orc_dpotrf_lower",
                     "orc_pcr_solve", "orc_cr_solve", "orc_kkt_diag_factorize",
                     "orc_kkt_diag_solve", "orc_trisolve", "orc_split_get_sizes",
                     "orc_kkt_basis_solve", "orc_newton_solve_diag", "orc_newton_solve_basis", "orc_ipm_step_diag"):
            getattr(L, name).restype = c_i64
        for name in ("orc_diag_factorize", "orc_kkt_diag_new", "orc_split_prepare", "orc_lu_factorize", "orc_lu_factorize_sparse", "orc_lu_factorize_policy", "orc_basis_new"):
            getattr(L, name).restype = C.c_void_p

    # ---- Iterate / StepToBoundary ---------------------------------------------
    def iterate_update(self, m, n, state, it, sp, dx, dxl, dxu, sd, dy, dzl, dzu):
        """it: dict with x, xl, xu, y, zl, zu (modified copies are returned); None steps are skipped."""
        out = {k: _F(it[k]).copy() for k in ("x", "xl", "xu", "y", "zl", "zu")}
        st = np.ascontiguousarray(state, dtype=np.uint8)
        self.lib.orc_iterate_update(c_i64(m), c_i64(n), st.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                    *[_fp(out[k]) for k in ("x", "xl", "xu", "y", "zl", "zu")], c_f64(sp),
                                    _fp(_F(dx)), _fp(_F(dxl)), _fp(_F(dxu)), c_f64(sd), _fp(_F(dy)),
                                    _fp(_F(dzl)), _fp(_F(dzu)))
        return out

    def iterate_residuals(self, A, state, b, c, lb, ub, it):
        m, n = A.nrow, A.ncol
        st = np.ascontiguousarray(state, dtype=np.uint8)
        rb, rc, rl, ru = np.zeros(m, f64), np.zeros(n + m, f64), np.zeros(n + m, f64), np.zeros(n + m, f64)
        norms = np.zeros(2, f64)
        self.lib.orc_iterate_residuals(c_i64(m), c_i64(n), _ip(A.p), _ip(A.i), _fp(A.x),
                                       st.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(_F(b)), _fp(_F(c)),
                                       _fp(_F(lb)), _fp(_F(ub)),
                                       *[_fp(_F(it[k])) for k in ("x", "xl", "xu", "y", "zl", "zu")],
                                       _fp(rb), _fp(rc), _fp(rl), _fp(ru), _fp(norms))
        return dict(rb=rb, rc=rc, rl=rl, ru=ru, presidual=float(norms[0]), dresidual=float(norms[1]))

    def iterate_complementarity(self, state, it):
        st = np.ascontiguousarray(state, dtype=np.uint8)
        out = np.zeros(4, f64)
        self.lib.orc_iterate_complementarity(c_i64(st.size), st.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                             *[_fp(_F(it[k])) for k in ("xl", "xu", "zl", "zu")], _fp(out))
        return dict(complementarity=float(out[0]), mu=float(out[1]), mu_min=float(out[2]), mu_max=float(out[3]))

    def step_to_boundary(self, x, dx, alpha=1.0):
        x, dx = _F(x), _F(dx)
        blk = c_i64(-1)
        self.lib.orc_step_to_boundary.restype = c_f64
        a = self.lib.orc_step_to_boundary(c_i64(x.size), _fp(x), _fp(dx), c_f64(alpha), C.byref(blk))
        return float(a), int(blk.value)

    # ---- vector / index kernels ------------------------------------------
    def dot(self, x, y):
        x, y = _F(x), _F(y)
        return self.lib.orc_dot(c_i64(x.size), _fp(x), _fp(y))

    def infnorm(self, x):
        x = _F(x)
        return self.lib.orc_infnorm(c_i64(x.size), _fp(x))

    def transpose(self, A):
        ATp = np.zeros(A.nrow + 1, i64)
        ATi = np.zeros(A.nnz, i64)
        ATx = np.zeros(A.nnz, f64)
        self.lib.orc_transpose(c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), _ip(A.i), _fp(A.x),
                               _ip(ATp), _ip(ATi), _fp(ATx))
        return Csc(A.ncol, A.nrow, ATp, ATi, ATx)

    def lu_factorize(self, dim, Bbegin, Bend, Bi, Bx, pivottol=0.1, strict=False, bump_limit=-1, sparse_min=None, slow_den=256, fill_max=8,
                     policy=None):
        """LuFactorization contract (src/lu_factorization.h:21-58): returns dict(L, U, rowperm, colperm, dependent,
        info) with L, U as Csc, or None when the bump exceeds bump_limit.  sparse_min: elimination rounds (not tearing)
        while more than that many columns are active.  policy = dict(sparse_from, rest_limit, sparse_min, slow_den, fill_max,
        dense_at): the device's default policy since round 5 (orc_lu_factorize_policy)."""
        Bbegin, Bend, Bi, Bx = _I(Bbegin), _I(Bend), _I(Bi), _F(Bx)
        if policy is not None:
            q = dict(sparse_from=1024, rest_limit=32768, sparse_min=512, slow_den=2048, fill_max=8, dense_at=0.2)
            q.update(policy)
            h = self.lib.orc_lu_factorize_policy(c_i64(dim), _ip(Bbegin), _ip(Bend), _ip(Bi), _fp(Bx), c_f64(pivottol), C.c_int(1 if strict else 0),
                                                 c_i64(q["sparse_from"]), c_i64(q["rest_limit"]), c_i64(q["sparse_min"]), c_i64(q["slow_den"]),
                                                 c_i64(q["fill_max"]), c_f64(q["dense_at"]))
        elif sparse_min is None:
            h = self.lib.orc_lu_factorize(c_i64(dim), _ip(Bbegin), _ip(Bend), _ip(Bi), _fp(Bx), c_f64(pivottol),
                                          C.c_int(1 if strict else 0), c_i64(bump_limit))
        else:
            h = self.lib.orc_lu_factorize_sparse(c_i64(dim), _ip(Bbegin), _ip(Bend), _ip(Bi), _fp(Bx), c_f64(pivottol),
                                                 C.c_int(1 if strict else 0), c_i64(bump_limit), c_i64(sparse_min), c_i64(slow_den), c_i64(fill_max))
        if not h:
            return None
        h = C.c_void_p(h)
        lnz, unz, ndep = c_i64(), c_i64(), c_i64()
        info = np.zeros(8, i64)
        self.lib.orc_lu_sizes(h, C.byref(lnz), C.byref(unz), C.byref(ndep), _ip(info))
        Lp, Up = np.zeros(dim + 1, i64), np.zeros(dim + 1, i64)
        Li, Lx = np.zeros(lnz.value, i64), np.zeros(lnz.value, f64)
        Ui, Ux = np.zeros(unz.value, i64), np.zeros(unz.value, f64)
        rowperm, colperm, dep = np.zeros(dim, i64), np.zeros(dim, i64), np.zeros(ndep.value, i64)
        self.lib.orc_lu_get(h, _ip(Lp), _ip(Li), _fp(Lx), _ip(Up), _ip(Ui), _fp(Ux), _ip(rowperm), _ip(colperm), _ip(dep))
        self.lib.orc_lu_free(h)
        return dict(L=Csc(dim, dim, Lp, Li, Lx), U=Csc(dim, dim, Up, Ui, Ux), rowperm=rowperm, colperm=colperm,
                    dependent=dep, info=dict(col_singletons=int(info[0]), row_singletons=int(info[1]),
                                             bump=int(info[2]), rounds=int(info[3]), dependent=int(info[4]),
                                             spikes=int(info[5]), sparse_pivots=int(info[6]), sparse_rounds=int(info[7])))

    def basis(self, A, basis, status, max_etas=100):
        """ipx::Basis as far as Maxvolume needs it, over [A I] (A: Csc m x n)"""
        return OracleBasis(self, A, basis, status, max_etas)

    def equilibrate(self, A):
        """Presolver::EquilibrateMatrix: (scaled values, colscale, rowscale, rounds); rounds = -1: untouched"""
        x = _F(A.x).copy()
        cs, rs = np.zeros(A.ncol, f64), np.zeros(A.nrow, f64)
        self.lib.orc_equilibrate.restype = c_i64
        r = self.lib.orc_equilibrate(c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), _ip(A.i), _fp(x), _fp(cs), _fp(rs))
        return x, cs, rs, int(r)

    def iterate_objectives(self, A, state, b, c, lb, ub, it):
        """Iterate::ComputeObjectives: (pobjective, dobjective, offset)"""
        out = np.zeros(3, f64)
        st = np.ascontiguousarray(state, dtype=np.uint8)
        self.lib.orc_iterate_objectives(c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), _ip(A.i), _fp(A.x),
                                        st.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(_F(b)), _fp(_F(c)), _fp(_F(lb)),
                                        _fp(_F(ub)), _fp(_F(it["x"])), _fp(_F(it["y"])), _fp(_F(it["zl"])),
                                        _fp(_F(it["zu"])), _fp(out))
        return tuple(float(v) for v in out)

    def model_norms(self, m, n, b, c, lb, ub):
        out = np.zeros(2, f64)
        self.lib.orc_model_norms(c_i64(m), c_i64(n), _fp(_F(b)), _fp(_F(c)), _fp(_F(lb)), _fp(_F(ub)), _fp(out))
        return float(out[0]), float(out[1])

    def inverse_perm(self, perm):
        perm = _I(perm)
        inv = np.zeros_like(perm)
        self.lib.orc_inverse_perm(c_i64(perm.size), _ip(perm), _ip(inv))
        return inv

    def copy_permute_scale(self, A, cols, perm=None, scale=None):
        cols, perm, scale = _I(cols), _I(perm), _F(scale)
        nnz = int((A.p[cols + 1] - A.p[cols]).sum())
        Np = np.zeros(cols.size + 1, i64)
        Ni = np.zeros(nnz, i64)
        Nx = np.zeros(nnz, f64)
        self.lib.orc_copy_permute_scale(c_i64(A.nrow), _ip(A.p), _ip(A.i), _fp(A.x),
                                        c_i64(cols.size), _ip(cols), _ip(perm), _fp(scale),
                                        _ip(Np), _ip(Ni), _fp(Nx))
        return Csc(A.nrow, cols.size, Np, Ni, Nx)

    def find_dense_columns(self, A):
        nz = c_i64(0)
        k = self.lib.orc_find_dense_columns(c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), C.byref(nz))
        return int(k), int(nz.value)

    # ---- operators ----------------------------------------------------------
    def normal_apply(self, A, W, rhs, want_dot=True):
        rhs, W = _F(rhs), _F(W)
        lhs = np.zeros(A.nrow, f64)
        dot = c_f64(0.0)
        self.lib.orc_normal_apply(c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), _ip(A.i), _fp(A.x),
                                  _fp(W), _fp(rhs), _fp(lhs), C.byref(dot) if want_dot else None)
        return lhs, dot.value

    def diag_factorize(self, A, W, nz_dense, precond_dense_cols=True):
        err = c_i64(0)
        W = _F(W)
        h = self.lib.orc_diag_factorize(c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), _ip(A.i),
                                        _fp(A.x), _fp(W), c_i64(nz_dense),
                                        c_i64(1 if precond_dense_cols else 0), C.byref(err))
        return (OracleDiagPrecond(self, h, A.nrow) if h else None), int(err.value)

    def dpotrf_lower(self, S):
        a = np.asfortranarray(S, dtype=f64).copy(order="F")
        k = a.shape[0]
        info = self.lib.orc_dpotrf_lower(c_i64(k), a.ctypes.data_as(P_f64), c_i64(k))
        return np.tril(a), int(info)

    def pcr_solve(self, applyC, applyP, rhs, tol, resscale, maxiter, lhs0=None, hist_cap=0):
        """applyC/applyP: python callables (rhs ndarray) -> (lhs ndarray, dot)."""
        rhs = _F(rhs)
        m = rhs.size
        lhs = np.zeros(m, f64) if lhs0 is None else _F(lhs0).copy()
        cbC, cbP = _wrap_apply(applyC, m), _wrap_apply(applyP, m)
        hist = np.zeros(max(hist_cap, 1), f64)
        it = c_i64(0)
        err = self.lib.orc_pcr_solve(c_i64(m), cbC, None, cbP, None, _fp(rhs), c_f64(tol),
                                     _fp(_F(resscale)), c_i64(maxiter), _fp(lhs), C.byref(it),
                                     _fp(hist), c_i64(hist_cap))
        return lhs, int(it.value), int(err), hist[:min(hist_cap, it.value + 1)]

    def cr_solve(self, applyC, rhs, tol, resscale, maxiter, lhs0=None, hist_cap=0):
        rhs = _F(rhs)
        m = rhs.size
        lhs = np.zeros(m, f64) if lhs0 is None else _F(lhs0).copy()
        cbC = _wrap_apply(applyC, m)
        hist = np.zeros(max(hist_cap, 1), f64)
        it = c_i64(0)
        err = self.lib.orc_cr_solve(c_i64(m), cbC, None, _fp(rhs), c_f64(tol),
                                    _fp(_F(resscale)), c_i64(maxiter), _fp(lhs), C.byref(it),
                                    _fp(hist), c_i64(hist_cap))
        return lhs, int(it.value), int(err), hist[:min(hist_cap, it.value + 1)]

    def kkt_diag(self, A, nz_dense=None, precond_dense_cols=True, maxiter=-1):
        return OracleKktDiag(self, A, nz_dense, precond_dense_cols, maxiter)

    # ---- triangular solves --------------------------------------------------
    def trisolve(self, T, x, trans, uplo, unitdiag):
        x = _F(x).copy()
        nz = self.lib.orc_trisolve(c_i64(T.ncol), _ip(T.p), _ip(T.i), _fp(T.x), _fp(x),
                                   C.c_char(trans.encode()), C.c_char(uplo.encode()),
                                   c_i64(unitdiag))
        return x, int(nz)

    def forward_solve(self, L, U, x):
        x = _F(x).copy()
        self.lib.orc_forward_solve(c_i64(L.ncol), _ip(L.p), _ip(L.i), _fp(L.x), _ip(U.p),
                                   _ip(U.i), _fp(U.x), _fp(x))
        return x

    def backward_solve(self, L, U, x):
        x = _F(x).copy()
        self.lib.orc_backward_solve(c_i64(L.ncol), _ip(L.p), _ip(L.i), _fp(L.x), _ip(U.p),
                                    _ip(U.i), _fp(U.x), _fp(x))
        return x

    def add_normal_product(self, A, D, rhs, lhs):
        lhs = _F(lhs).copy()
        self.lib.orc_add_normal_product(c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), _ip(A.i),
                                        _fp(A.x), _fp(_F(D)), _fp(_F(rhs)), _fp(lhs))
        return lhs

    def split_prepare(self, AI, n, L, U, rowperm, colperm, basis, status, colscale):
        return OracleSplit(self, AI, n, L, U, rowperm, colperm, basis, status, colscale)


def _wrap_apply(fn, m):
    def cb(_ctx, rhs_p, lhs_p, dot_p):
        rhs = np.ctypeslib.as_array(rhs_p, shape=(m,))
        lhs, dot = fn(rhs.copy())
        np.ctypeslib.as_array(lhs_p, shape=(m,))[:] = lhs
        if dot_p:
            dot_p[0] = dot
    return APPLY_FN(cb)


class OracleBasis:
    def __init__(self, orc, A, basis, status, max_etas=100):
        self.lib, self.m, self.n = orc.lib, A.nrow, A.ncol
        self.keep = (_I(A.p), _I(A.i), _F(A.x))
        err = c_i64(0)
        self.lib.orc_basis_exchange_if_stable.restype = c_i64
        self.lib.orc_maxvolume_heuristic.restype = c_i64
        self.lib.orc_maxvolume_sequential.restype = c_i64
        self.h = C.c_void_p(self.lib.orc_basis_new(c_i64(self.m), c_i64(self.n), _ip(self.keep[0]), _ip(self.keep[1]),
                                                   _fp(self.keep[2]), _ip(_I(basis)), _ip(_I(status)), c_i64(max_etas),
                                                   C.byref(err)))
        self.errflag = err.value

    def get(self):
        basis, status, counts = np.zeros(self.m, i64), np.zeros(self.n + self.m, i64), np.zeros(5, i64)
        self.lib.orc_basis_get(self.h, _ip(basis), _ip(status), _ip(counts))
        return basis, status, dict(zip(("factorizations", "updates", "ftran", "btran", "etas"), counts.tolist()))

    def solve_dense(self, rhs, trans):
        lhs = np.zeros(self.m, f64)
        self.lib.orc_basis_solve_dense(self.h, _fp(_F(rhs)), _fp(lhs), C.c_char(trans.encode()))
        return lhs

    def solve_for_update(self, j):
        lhs = np.zeros(self.m, f64)
        self.lib.orc_basis_solve_for_update(self.h, c_i64(j), _fp(lhs))
        return lhs

    def tableau_row(self, jb, ignore_fixed=True):
        btran, row = np.zeros(self.m, f64), np.zeros(self.n + self.m, f64)
        self.lib.orc_basis_tableau_row(self.h, c_i64(jb), _fp(btran), _fp(row), C.c_int(1 if ignore_fixed else 0))
        return btran, row

    def exchange_if_stable(self, jb, jn, tableau_entry):
        ex = c_i64(0)
        err = self.lib.orc_basis_exchange_if_stable(self.h, c_i64(jb), c_i64(jn), c_f64(tableau_entry), C.byref(ex))
        return int(err), bool(ex.value)

    def maxvolume(self, colscale, volume_tol=2.0, maxskip_updates=10, rows_per_slice=10000, log_cap=100000):
        info = np.zeros(8, f64)
        log = np.zeros(2 * log_cap, i64)
        err = self.lib.orc_maxvolume_heuristic(self.h, _fp(_F(colscale)), c_f64(volume_tol), c_i64(maxskip_updates),
                                               c_i64(rows_per_slice), _fp(info), _ip(log), c_i64(log_cap))
        k = int(info[0])
        return dict(errflag=int(err), updates=k, skipped=int(info[1]), slices=int(info[2]), volinc=float(info[3]),
                    refused=int(info[4]), exchanges=log[: 2 * min(k, log_cap)].reshape(-1, 2))

    def maxvolume_sequential(self, colscale, volume_tol=2.0, maxpasses=-1, log_cap=100000):
        info = np.zeros(8, f64)
        log = np.zeros(2 * log_cap, i64)
        err = self.lib.orc_maxvolume_sequential(self.h, _fp(_F(colscale)), c_f64(volume_tol), c_i64(maxpasses), _fp(info), _ip(log),
                                                c_i64(log_cap))
        k = int(info[0])
        return dict(errflag=int(err), updates=k, skipped=int(info[1]), passes=int(info[2]), volinc=float(info[3]),
                    refused=int(info[4]), tblnnz=int(info[6]), tblmax=float(info[7]), exchanges=log[: 2 * min(k, log_cap)].reshape(-1, 2))

    def close(self):
        if self.h:
            self.lib.orc_basis_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class OracleDiagPrecond:
    def __init__(self, orc, h, m):
        self.orc, self.h, self.m = orc, C.c_void_p(h), m

    @property
    def num_dense(self):
        return int(self.orc.lib.orc_diag_num_dense(self.h))

    def apply(self, rhs):
        rhs = _F(rhs)
        lhs = np.zeros(self.m, f64)
        dot = c_f64(0.0)
        self.orc.lib.orc_diag_apply(self.h, _fp(rhs), _fp(lhs), C.byref(dot))
        return lhs, dot.value

    def get(self):
        k = self.num_dense
        d = np.zeros(self.m, f64)
        ch = np.zeros(k * k, f64)
        self.orc.lib.orc_diag_get(self.h, _fp(d), _fp(ch) if k else None)
        return d, ch.reshape(k, k).T  # column-major -> [row, col]

    def __del__(self):
        if self.h:
            self.orc.lib.orc_diag_free(self.h)
            self.h = None


class OracleKktDiag:
    def __init__(self, orc, A, nz_dense, precond_dense_cols, maxiter):
        self.orc, self.A = orc, A
        self.m, self.n = A.nrow, A.ncol
        if nz_dense is None:
            nz_dense = orc.find_dense_columns(A)[1]
        self.nz_dense = nz_dense
        self.h = C.c_void_p(orc.lib.orc_kkt_diag_new(
            c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), _ip(A.i), _fp(A.x), c_i64(nz_dense),
            c_i64(1 if precond_dense_cols else 0), c_i64(maxiter)))

    def factorize(self, xl=None, xu=None, zl=None, zu=None, mu=0.0):
        xl, xu, zl, zu = _F(xl), _F(xu), _F(zl), _F(zu)
        return int(self.orc.lib.orc_kkt_diag_factorize(self.h, _fp(xl), _fp(xu), _fp(zl),
                                                       _fp(zu), c_f64(mu)))

    def solve(self, a, b, tol, hist_cap=0):
        a, b = _F(a), _F(b)
        x = np.zeros(self.n + self.m, f64)
        y = np.zeros(self.m, f64)
        it = c_i64(0)
        hist = np.zeros(max(hist_cap, 1), f64)
        err = self.orc.lib.orc_kkt_diag_solve(self.h, _fp(a), _fp(b), c_f64(tol), _fp(x),
                                              _fp(y), C.byref(it), _fp(hist), c_i64(hist_cap))
        return x, y, int(it.value), int(err), hist[:min(hist_cap, it.value + 1)]

    def newton_solve(self, rb, rc, rl, ru, sl, su, xl, xu, zl, zu, state, tol):
        """IPM::SolveNewtonSystem (src/ipm.cc:532-645) around this solver; None = zero vector."""
        N = self.n + self.m
        ins = [_F(v) for v in (rb, rc, rl, ru, sl, su, xl, xu, zl, zu)]
        st = np.ascontiguousarray(state, dtype=np.uint8)
        out = {k: np.zeros(self.m if k == "dy" else N, f64) for k in ("dx", "dxl", "dxu", "dy", "dzl", "dzu")}
        it = c_i64(0)
        err = self.orc.lib.orc_newton_solve_diag(
            self.h, *[_fp(v) for v in ins], st.ctypes.data_as(C.POINTER(C.c_ubyte)), c_f64(tol),
            _fp(out["dx"]), _fp(out["dxl"]), _fp(out["dxu"]), _fp(out["dy"]), _fp(out["dzl"]), _fp(out["dzu"]),
            C.byref(it))
        out.update(iter=int(it.value), errflag=int(err))
        return out

    def ipm_step(self, state, b, c, lb, ub, it, kkt_tol=0.3):
        """IPM::Predictor/AddCorrector/StepSizes/MakeStep (src/ipm.cc:340-530); returns (new iterate, info)."""
        out = {k: _F(it[k]).copy() for k in ("x", "xl", "xu", "y", "zl", "zu")}
        st = np.ascontiguousarray(state, dtype=np.uint8)
        info = np.zeros(7, f64)
        err = self.orc.lib.orc_ipm_step_diag(self.h, st.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(_F(b)), _fp(_F(c)),
                                             _fp(_F(lb)), _fp(_F(ub)),
                                             *[_fp(out[k]) for k in ("x", "xl", "xu", "y", "zl", "zu")],
                                             c_f64(kkt_tol), _fp(info))
        keys = ("step_primal", "step_dual", "mu_before", "mu_after", "sigma", "kktiter_predictor", "kktiter_corrector")
        d = dict(zip(keys, (float(v) for v in info)))
        d["errflag"] = int(err)
        return out, d

    def ipm_driver(self, state, b, c, lb, ub, it, kkt_tol=0.3, feasibility_tol=1e-6, optimality_tol=1e-8,
                   ipm_maxiter=300):
        """IPM::Driver (src/ipm.cc:56-123) around this KKTSolverDiag; returns (final iterate, info)."""
        out = {k: _F(it[k]).copy() for k in ("x", "xl", "xu", "y", "zl", "zu")}
        st = np.ascontiguousarray(state, dtype=np.uint8)
        info = np.zeros(10, f64)
        self.orc.lib.orc_ipm_driver_diag.restype = c_i64
        status = self.orc.lib.orc_ipm_driver_diag(
            self.h, st.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(_F(b)), _fp(_F(c)), _fp(_F(lb)), _fp(_F(ub)),
            *[_fp(out[k]) for k in ("x", "xl", "xu", "y", "zl", "zu")], c_f64(kkt_tol), c_f64(feasibility_tol),
            c_f64(optimality_tol), c_i64(ipm_maxiter), _fp(info))
        keys = ("iter", "errflag", "kktiter", "pobjective", "dobjective", "presidual", "dresidual", "complementarity",
                "mu", "last_step")
        d = dict(zip(keys, (float(v) for v in info)))
        d["status_ipm"] = int(status)
        return out, d

    def get(self):
        W = np.zeros(self.n + self.m, f64)
        rs = np.zeros(self.m, f64)
        self.orc.lib.orc_kkt_diag_get(self.h, _fp(W), _fp(rs))
        return W, rs

    def __del__(self):
        if self.h:
            self.orc.lib.orc_kkt_diag_free(self.h)
            self.h = None


class OracleSplit:
    def __init__(self, orc, AI, n, L, U, rowperm, colperm, basis, status, colscale):
        self.orc, self.m, self.n = orc, AI.nrow, n
        self._keep = (AI, L, U)
        self.h = C.c_void_p(orc.lib.orc_split_prepare(
            c_i64(AI.nrow), c_i64(n), _ip(AI.p), _ip(AI.i), _fp(AI.x), _ip(L.p), _ip(L.i),
            _fp(L.x), _ip(U.p), _ip(U.i), _fp(U.x), _ip(_I(rowperm)), _ip(_I(colperm)),
            _ip(_I(basis)), _ip(_I(status)), _fp(_F(colscale))))
        self.nnzU = U.nnz

    def apply(self, rhs):
        rhs = _F(rhs)
        lhs = np.zeros(self.m, f64)
        dot = c_f64(0.0)
        self.orc.lib.orc_split_apply(self.h, _fp(rhs), _fp(lhs), C.byref(dot))
        return lhs, dot.value

    def get(self):
        nnzN, ncolN, nfree = c_i64(0), c_i64(0), c_i64(0)
        self.orc.lib.orc_split_get_sizes(self.h, C.byref(nnzN), C.byref(ncolN), C.byref(nfree))
        Np = np.zeros(ncolN.value + 1, i64)
        Ni = np.zeros(nnzN.value, i64)
        Nx = np.zeros(nnzN.value, f64)
        Ux = np.zeros(self.nnzU, f64)
        rpi = np.zeros(self.m, i64)
        fp = np.zeros(nfree.value, i64)
        self.orc.lib.orc_split_get(self.h, _ip(Np), _ip(Ni), _fp(Nx), _fp(Ux), _ip(rpi), _ip(fp))
        return dict(N=Csc(self.m, ncolN.value, Np, Ni, Nx), Ux=Ux, rowperm_inv=rpi,
                    free_positions=fp)

    def solve_dense(self, rhs, trans):
        rhs = _F(rhs)
        lhs = np.zeros(self.m, f64)
        self.orc.lib.orc_split_solve_dense(self.h, _fp(rhs), _fp(lhs), C.c_char(trans.encode()))
        return lhs

    def kkt_solve(self, a, b, tol, maxiter=-1, hist_cap=0):
        a, b = _F(a), _F(b)
        x = np.zeros(self.n + self.m, f64)
        y = np.zeros(self.m, f64)
        it = c_i64(0)
        hist = np.zeros(max(hist_cap, 1), f64)
        err = self.orc.lib.orc_kkt_basis_solve(self.h, _fp(a), _fp(b), c_f64(tol),
                                               c_i64(maxiter), _fp(x), _fp(y), C.byref(it),
                                               _fp(hist), c_i64(hist_cap))
        return x, y, int(it.value), int(err), hist[:min(hist_cap, it.value + 1)]

    def newton_solve(self, rb, rc, rl, ru, sl, su, xl, xu, zl, zu, state, tol, maxiter=-1):
        N = self.n + self.m
        ins = [_F(v) for v in (rb, rc, rl, ru, sl, su, xl, xu, zl, zu)]
        st = np.ascontiguousarray(state, dtype=np.uint8)
        out = {k: np.zeros(self.m if k == "dy" else N, f64) for k in ("dx", "dxl", "dxu", "dy", "dzl", "dzu")}
        it = c_i64(0)
        err = self.orc.lib.orc_newton_solve_basis(
            self.h, *[_fp(v) for v in ins], st.ctypes.data_as(C.POINTER(C.c_ubyte)), c_f64(tol), c_i64(maxiter),
            _fp(out["dx"]), _fp(out["dxl"]), _fp(out["dxu"]), _fp(out["dy"]), _fp(out["dzl"]), _fp(out["dzu"]),
            C.byref(it))
        out.update(iter=int(it.value), errflag=int(err))
        return out

    def __del__(self):
        if self.h:
            self.orc.lib.orc_split_free(self.h)
            self.h = None


# ---------------------------------------------------------------------------
# reference objects (oracle/_ref) -- only where the reference could be built
# ---------------------------------------------------------------------------
def ref_available():
    return os.path.exists(REF_SO)


class Ref:
    def __init__(self, path=REF_SO):
        self.lib = L = C.CDLL(path)
        for name in ("ref_model_new", "ref_kktdiag_new", "ref_split_new", "ref_lu_new"):
            getattr(L, name).restype = C.c_void_p
        for name in ("ref_model_is_dense", "ref_diagprec_apply", "ref_pcr_solve",
                     "ref_kktdiag_factorize", "ref_kktdiag_solve", "ref_trisolve",
                     "ref_split_cr_solve", "ref_lu_update", "ref_lu_updates"):
            getattr(L, name).restype = c_i64
        L.ref_dot.restype = c_f64
        L.ref_infnorm.restype = c_f64

    def model(self, A, rhs, constr_type, obj, lb, ub):
        return RefModel(self, A, rhs, constr_type, obj, lb, ub)

    def trisolve(self, T, x, trans, uplo, unitdiag):
        x = _F(x).copy()
        nz = self.lib.ref_trisolve(c_i64(T.ncol), _ip(T.p), _ip(T.i), _fp(T.x), _fp(x),
                                   C.c_char(trans.encode()), C.c_char(uplo.encode()),
                                   c_i64(unitdiag))
        return x, int(nz)

    def forward_solve(self, L, U, x):
        x = _F(x).copy()
        self.lib.ref_forward_solve(c_i64(L.ncol), _ip(L.p), _ip(L.i), _fp(L.x), _ip(U.p),
                                   _ip(U.i), _fp(U.x), _fp(x))
        return x

    def backward_solve(self, L, U, x):
        x = _F(x).copy()
        self.lib.ref_backward_solve(c_i64(L.ncol), _ip(L.p), _ip(L.i), _fp(L.x), _ip(U.p),
                                    _ip(U.i), _fp(U.x), _fp(x))
        return x

    def add_normal_product(self, A, D, rhs, lhs):
        lhs = _F(lhs).copy()
        self.lib.ref_add_normal_product(c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), _ip(A.i),
                                        _fp(A.x), _fp(_F(D)), _fp(_F(rhs)), _fp(lhs))
        return lhs

    def transpose(self, A):
        ATp = np.zeros(A.nrow + 1, i64)
        ATi = np.zeros(A.nnz, i64)
        ATx = np.zeros(A.nnz, f64)
        self.lib.ref_transpose(c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), _ip(A.i), _fp(A.x),
                               _ip(ATp), _ip(ATi), _fp(ATx))
        return Csc(A.ncol, A.nrow, ATp, ATi, ATx)

    def copy_permute_scale(self, A, cols, perm=None, scale=None):
        cols, perm, scale = _I(cols), _I(perm), _F(scale)
        nnz = int((A.p[cols + 1] - A.p[cols]).sum())
        Np = np.zeros(cols.size + 1, i64)
        Ni = np.zeros(nnz, i64)
        Nx = np.zeros(nnz, f64)
        self.lib.ref_copy_permute_scale(c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), _ip(A.i),
                                        _fp(A.x), c_i64(cols.size), _ip(cols), _ip(perm),
                                        _fp(scale), _ip(Np), _ip(Ni), _fp(Nx))
        return Csc(A.nrow, cols.size, Np, Ni, Nx)

    def inverse_perm(self, perm):
        perm = _I(perm)
        inv = np.zeros_like(perm)
        self.lib.ref_inverse_perm(c_i64(perm.size), _ip(perm), _ip(inv))
        return inv

    def dot(self, x, y):
        x, y = _F(x), _F(y)
        return self.lib.ref_dot(c_i64(x.size), _fp(x), _fp(y))

    def split(self, L, U, N, free_positions):
        return RefSplit(self, L, U, N, free_positions)

    def lu(self, dim, Bbegin, Bend, Bi, Bx, F):
        """The reference's LuFactorization::Factorize (stability estimate) + ForrestTomlin on the factors F =
        dict(L, U, rowperm, colperm, dependent) computed by the code under test."""
        return RefLu(self, dim, Bbegin, Bend, Bi, Bx, F)


class RefLu:
    def __init__(self, ref, dim, Bbegin, Bend, Bi, Bx, F):
        self.lib, self.dim = ref.lib, dim
        self.keep = [_I(Bbegin), _I(Bend), _I(Bi), _F(Bx)]
        L, U = F["L"], F["U"]
        dep = _I(F["dependent"])
        self.h = C.c_void_p(self.lib.ref_lu_new(c_i64(dim), *[_ip(a) for a in self.keep[:3]], _fp(self.keep[3]),
                                                _ip(L.p), _ip(L.i), _fp(L.x), _ip(U.p), _ip(U.i), _fp(U.x),
                                                _ip(_I(F["rowperm"])), _ip(_I(F["colperm"])), c_i64(dep.size), _ip(dep)))
        out = np.zeros(3, f64)
        self.lib.ref_lu_info(self.h, _fp(out))
        self.flag, self.stability, self.fill_factor = int(out[0]), float(out[1]), float(out[2])

    def solve_dense(self, rhs, trans=False):
        lhs = np.zeros(self.dim, f64)
        self.lib.ref_lu_solve_dense(self.h, _fp(_F(rhs)), _fp(lhs), c_i64(1 if trans else 0))
        return lhs

    def ftran(self, bi, bx):
        bi, bx = _I(bi), _F(bx)
        lhs = np.zeros(self.dim, f64)
        self.lib.ref_lu_ftran(self.h, c_i64(bi.size), _ip(bi), _fp(bx), _fp(lhs))
        return lhs

    def btran(self, p):
        lhs = np.zeros(self.dim, f64)
        self.lib.ref_lu_btran(self.h, c_i64(p), _fp(lhs))
        return lhs

    def update(self, pivot):
        return int(self.lib.ref_lu_update(self.h, c_f64(pivot)))

    def updates(self):
        return int(self.lib.ref_lu_updates(self.h))

    def close(self):
        if self.h:
            self.lib.ref_lu_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class RefIterate:
    """The reference's ipx::Iterate on a RefModel."""

    KEYS = ("x", "xl", "xu", "y", "zl", "zu")

    def __init__(self, model):
        self.model, self.lib = model, model.ref.lib
        self.lib.ref_iterate_new.restype = C.c_void_p
        self.h = C.c_void_p(self.lib.ref_iterate_new(model.h))
        self.m, self.n = model.m, model.n

    def initialize(self, it):
        self.lib.ref_iterate_initialize(self.h, *[_fp(_F(it[k])) for k in self.KEYS])

    def update(self, sp, dx, dxl, dxu, sd, dy, dzl, dzu):
        self.lib.ref_iterate_update(self.h, c_f64(sp), _fp(_F(dx)), _fp(_F(dxl)), _fp(_F(dxu)), c_f64(sd),
                                    _fp(_F(dy)), _fp(_F(dzl)), _fp(_F(dzu)))

    def get(self):
        N = self.n + self.m
        out = {k: np.zeros(self.m if k == "y" else N, f64) for k in self.KEYS}
        self.lib.ref_iterate_get(self.h, *[_fp(out[k]) for k in self.KEYS])
        return out

    def states(self):
        st = np.zeros(self.n + self.m, np.uint8)
        self.lib.ref_iterate_states(self.h, st.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return st

    def residuals(self):
        N = self.n + self.m
        rb, rc, rl, ru = np.zeros(self.m, f64), np.zeros(N, f64), np.zeros(N, f64), np.zeros(N, f64)
        norms = np.zeros(2, f64)
        self.lib.ref_iterate_residuals(self.h, _fp(rb), _fp(rc), _fp(rl), _fp(ru), _fp(norms))
        return dict(rb=rb, rc=rc, rl=rl, ru=ru, presidual=float(norms[0]), dresidual=float(norms[1]))

    def objectives(self):
        """pobjective, dobjective, and both after postprocessing"""
        out = np.zeros(4, f64)
        self.lib.ref_iterate_objectives(self.h, _fp(out))
        return tuple(float(v) for v in out)

    def termination(self, feasibility_tol=1e-6, optimality_tol=1e-8):
        """(feasible, optimal, term_crit_reached) with crossover_start = 0"""
        out = np.zeros(3, i64)
        self.lib.ref_iterate_termination(self.h, c_f64(feasibility_tol), c_f64(optimality_tol), _ip(out))
        return tuple(bool(v) for v in out)

    def complementarity(self):
        out = np.zeros(4, f64)
        self.lib.ref_iterate_complementarity(self.h, _fp(out))
        return dict(complementarity=float(out[0]), mu=float(out[1]), mu_min=float(out[2]), mu_max=float(out[3]))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_iterate_free(self.h)
            self.h = None


class RefModel:
    """Solver-form model built by the reference's UserModel::Load + Presolver."""

    def __init__(self, ref, A, rhs, constr_type, obj, lb, ub):
        self.ref = ref
        dims = np.zeros(5, i64)
        ct = np.frombuffer(constr_type.encode() if isinstance(constr_type, str)
                           else bytes(constr_type), dtype=np.uint8).copy()
        h = ref.lib.ref_model_new(c_i64(A.nrow), c_i64(A.ncol), _ip(A.p), _ip(A.i), _fp(A.x),
                                  _fp(_F(rhs)), ct.ctypes.data_as(C.c_char_p), _fp(_F(obj)),
                                  _fp(_F(lb)), _fp(_F(ub)), _ip(dims))
        if not h:
            raise ValueError("reference rejected the model: errflag %d" % -dims[0])
        self.h = C.c_void_p(h)
        self.m, self.n, self.nnz_AI, self.dualized, self.num_dense = (int(v) for v in dims)

    def AI(self):
        p = np.zeros(self.n + self.m + 1, i64)
        i = np.zeros(self.nnz_AI, i64)
        x = np.zeros(self.nnz_AI, f64)
        self.ref.lib.ref_model_get_AI(self.h, _ip(p), _ip(i), _fp(x))
        return Csc(self.m, self.n + self.m, p, i, x)

    def AIt(self):
        p = np.zeros(self.m + 1, i64)
        i = np.zeros(self.nnz_AI, i64)
        x = np.zeros(self.nnz_AI, f64)
        self.ref.lib.ref_model_get_AIt(self.h, _ip(p), _ip(i), _fp(x))
        return Csc(self.n + self.m, self.m, p, i, x)

    def vectors(self):
        b = np.zeros(self.m, f64)
        c = np.zeros(self.n + self.m, f64)
        lb = np.zeros(self.n + self.m, f64)
        ub = np.zeros(self.n + self.m, f64)
        self.ref.lib.ref_model_get_vectors(self.h, _fp(b), _fp(c), _fp(lb), _fp(ub))
        return b, c, lb, ub

    def is_dense(self, j):
        return bool(self.ref.lib.ref_model_is_dense(self.h, c_i64(j)))

    def iterate(self):
        return RefIterate(self)

    def normal_apply(self, W, rhs, want_dot=True):
        lhs = np.zeros(self.m, f64)
        dot = c_f64(0.0)
        self.ref.lib.ref_normal_apply(self.h, _fp(_F(W)), _fp(_F(rhs)), _fp(lhs),
                                      C.byref(dot) if want_dot else None)
        return lhs, dot.value

    def diagprec_apply(self, W, precond_dense_cols, rhs):
        lhs = np.zeros(self.m, f64)
        dot = c_f64(0.0)
        err = self.ref.lib.ref_diagprec_apply(self.h, _fp(_F(W)),
                                              c_i64(1 if precond_dense_cols else 0),
                                              _fp(_F(rhs)), _fp(lhs), C.byref(dot))
        return lhs, dot.value, int(err)

    def pcr_solve(self, W, precond_dense_cols, rhs, tol, resscale, maxiter, lhs0=None,
                  hist_cap=4096):
        lhs = np.zeros(self.m, f64) if lhs0 is None else _F(lhs0).copy()
        it = c_i64(0)
        ch = np.zeros(hist_cap, f64)
        ph = np.zeros(hist_cap, f64)
        nc = np.zeros(2, i64)
        err = self.ref.lib.ref_pcr_solve(self.h, _fp(_F(W)), c_i64(1 if precond_dense_cols else 0),
                                         _fp(_F(rhs)), c_f64(tol), _fp(_F(resscale)),
                                         c_i64(maxiter), _fp(lhs), C.byref(it), _fp(ch), _fp(ph),
                                         c_i64(hist_cap), _ip(nc))
        return lhs, int(it.value), int(err), ch[:min(hist_cap, nc[0])], ph[:min(hist_cap, nc[1])]

    def kkt_diag(self, maxiter=-1, precond_dense_cols=True):
        return RefKktDiag(self, maxiter, precond_dense_cols)

    def __del__(self):
        if getattr(self, "h", None):
            self.ref.lib.ref_model_free(self.h)
            self.h = None


class RefKktDiag:
    def __init__(self, model, maxiter, precond_dense_cols):
        self.model = model
        self.lib = model.ref.lib
        self.h = C.c_void_p(self.lib.ref_kktdiag_new(model.h, c_i64(maxiter),
                                                     c_i64(1 if precond_dense_cols else 0)))

    def factorize(self, x=None, xl=None, xu=None, y=None, zl=None, zu=None):
        mu = c_f64(0.0)
        err = self.lib.ref_kktdiag_factorize(self.h, _fp(_F(x)), _fp(_F(xl)), _fp(_F(xu)),
                                             _fp(_F(y)), _fp(_F(zl)), _fp(_F(zu)), C.byref(mu))
        return int(err), mu.value

    def solve(self, a, b, tol):
        m, n = self.model.m, self.model.n
        x = np.zeros(n + m, f64)
        y = np.zeros(m, f64)
        it = c_i64(0)
        err = self.lib.ref_kktdiag_solve(self.h, _fp(_F(a)), _fp(_F(b)), c_f64(tol), _fp(x),
                                         _fp(y), C.byref(it))
        return x, y, int(it.value), int(err)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_kktdiag_free(self.h)
            self.h = None


class RefSplit:
    def __init__(self, ref, L, U, N, free_positions):
        self.lib, self.m = ref.lib, L.ncol
        fp = _I(free_positions)
        self.h = C.c_void_p(self.lib.ref_split_new(
            c_i64(self.m), _ip(L.p), _ip(L.i), _fp(L.x), _ip(U.p), _ip(U.i), _fp(U.x),
            c_i64(N.ncol), _ip(N.p), _ip(N.i), _fp(N.x), c_i64(fp.size), _ip(fp)))

    def apply(self, rhs):
        lhs = np.zeros(self.m, f64)
        dot = c_f64(0.0)
        self.lib.ref_split_apply(self.h, c_i64(self.m), _fp(_F(rhs)), _fp(lhs), C.byref(dot))
        return lhs, dot.value

    def cr_solve(self, rhs, tol, maxiter, resscale=None, hist_cap=4096):
        lhs = np.zeros(self.m, f64)
        it = c_i64(0)
        ch = np.zeros(hist_cap, f64)
        err = self.lib.ref_split_cr_solve(self.h, c_i64(self.m), _fp(_F(rhs)), c_f64(tol),
                                          _fp(_F(resscale)), c_i64(maxiter), _fp(lhs),
                                          C.byref(it), _fp(ch), c_i64(hist_cap))
        return lhs, int(it.value), int(err), ch[:min(hist_cap, it.value + 1)]

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_split_free(self.h)
            self.h = None
