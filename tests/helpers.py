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


def amax(v):
    return float(np.abs(v).max()) if v.size else 0.0


def check_newton_equations(AI, st, step, tol_kkt):
    """AI = [A I] (scipy).  The six block equations of the Newton system."""
    state = st["state"]
    lb, ub = (state == 2) | (state == 4), (state == 3) | (state == 4)
    bar = state >= 2
    dx, dxl, dxu, dy, dzl, dzu = (step[k] for k in ("dx", "dxl", "dxu", "dy", "dzl", "dzu"))
    scale = 1.0 + max(np.abs(dx).max(), np.abs(dy).max())
    assert np.abs(AI @ dx - st["rb"]).max() <= 1e-9 * scale                       # A dx = rb
    assert amax(dx[lb] - dxl[lb] - st["rl"][lb]) <= 1e-12 * scale          # dx - dxl = rl
    assert amax(dx[ub] + dxu[ub] - st["ru"][ub]) <= 1e-12 * scale          # dx + dxu = ru
    # A'dy + dzl - dzu = rc holds exactly on barrier variables (the residual was shifted away from it)
    dual = AI.T @ dy + dzl - dzu - st["rc"]
    assert np.abs(dual[bar]).max() <= 1e-10 * (1.0 + np.abs(dzl).max() + np.abs(dzu).max())
    # complementarity rows: exact on the side that was not overwritten by the shift, within the KKT
    # tolerance (scaled) on the other
    cl = st["zl"][lb] * dxl[lb] + st["xl"][lb] * dzl[lb] - st["sl"][lb]
    cu = st["zu"][ub] * dxu[ub] + st["xu"][ub] * dzu[ub] - st["su"][ub]
    G = np.zeros(len(state))
    G[lb] += st["zl"][lb] / st["xl"][lb]
    G[ub] += st["zu"][ub] / st["xu"][ub]
    # residual of the first KKT block row, D = G^{-1/2} scaling (src/kkt_solver.h:21-27)
    rl0, ru0 = np.where(lb, st["rl"], 0.0), np.where(ub, st["ru"], 0.0)
    assert np.isfinite(cl).all() and np.isfinite(cu).all()
    res_l = np.zeros(len(state)); res_l[lb] = cl / st["xl"][lb]
    res_u = np.zeros(len(state)); res_u[ub] = cu / st["xu"][ub]
    assert np.abs((res_l - res_u)[bar] / np.sqrt(G[bar])).max() <= tol_kkt * (1 + 1e-6) + 1e-9 * scale
    # free and fixed variables carry no bound steps
    nb = ~bar
    assert not dxl[nb].any() and not dxu[nb].any() and not dzl[nb].any() and not dzu[nb].any()
    assert not dx[state == 0].any()
