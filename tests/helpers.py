"""Shared builders for the test-suite (inputs only; no oracle or product code here)."""
import numpy as np

from ipx_amd import synth


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    d = np.abs(b).max()
    return float(np.abs(a - b).max() / (d if d > 0 else 1.0))


def diag_problem(m, n, seed=1, spread=1.0, num_dense=0, k=8):
    A = synth.synthetic_lp(m, n, k, seed, num_dense=num_dense)
    st = synth.synthetic_ipm_state(m, n, spread, seed)
    return A, st


def basis_problem(m, n, seed=3, num_free=0, num_fixed=0, spread=1.0, offdiag=3, band=None):
    A0 = synth.synthetic_lp(m, n, 8, seed)
    B = synth.planted_lu_basis(A0, offdiag=offdiag, seed=seed, band=band, num_free=num_free,
                               num_fixed=num_fixed)
    st = synth.synthetic_ipm_state(m, n, 1.0, seed)
    colscale = synth.synthetic_basis_state(B["status"], spread, seed)
    return B, st, colscale


def kkt_residual_diag(A, W, a, b, x, y):
    """Residual of (2) in reference src/kkt_solver.h:21-27 for G = inv(W): returns
    (res1 = G x + AI'y - a, res2 = AI x - b) with AI = [A I]."""
    S = A.to_scipy()
    n = A.ncol
    aty = np.concatenate([S.T @ y, y])
    res1 = x / W + aty - a
    res2 = S @ x[:n] + x[n:] - b
    return res1, res2
