"""GPU: ipxk_newton_solve = IPM::SolveNewtonSystem (reference src/ipm.cc:532-645) around the resident KKT
solvers, against the oracle's restatement and against the Newton equations the step must satisfy.

The reference function itself cannot be linked here (ipm.cc needs ipx::Basis, hence BASICLU): the
oracle's restatement is parity-unpinned for this row; the equation checks below do not depend on it.
Tolerances: quantities that only involve elementwise arithmetic of the solve's outputs are compared
at 1e-12 of their scale; the solve's outputs dx, dy carry the CR tolerance (1e-6 relative between
two implementations that stop at the same criterion)."""
import numpy as np
import pytest

from helpers import basis_problem, check_newton_equations, relerr
from ipx_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kkt():
    from ipx_amd import kkt as k
    k.load_library()
    return k


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


def test_newton_solve_diag(kkt, po):
    m, n = 900, 2000
    A = synth.synthetic_lp(m, n, 8, 21)
    st = synth.synthetic_newton_state(m, n, 21, num_free=6, num_ub=40, num_boxed=60)
    tol = 0.3 * np.sqrt(st["mu"])
    ctx = kkt.KktContext(A)
    assert ctx.kkt_diag_factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"]) == 0
    args = [st[k] for k in ("rb", "rc", "rl", "ru", "sl", "su", "xl", "xu", "zl", "zu", "state")]
    g = ctx.newton_solve(False, *args, tol, 500)
    orc = po.Oracle()
    k = orc.kkt_diag(po.Csc(m, n, A.p, A.i, A.x), maxiter=500)
    assert k.factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"]) == 0
    o = k.newton_solve(*args, tol)
    assert g["errflag"] == o["errflag"] == 0 and abs(g["iter"] - o["iter"]) <= 2
    for key in ("dx", "dxl", "dxu", "dy", "dzl", "dzu"):
        assert relerr(g[key], o[key]) < 1e-6, key
    AI = A.with_identity().to_scipy()
    check_newton_equations(AI, st, g, tol)
    check_newton_equations(AI, st, o, tol)
    # NULL residuals (the corrector call passes rb = rc = rl = ru = nullptr, ipm.cc:418-420)
    g0 = ctx.newton_solve(False, None, None, None, None, *args[4:], tol, 500)
    o0 = k.newton_solve(None, None, None, None, *args[4:], tol)
    assert g0["errflag"] == 0
    for key in ("dx", "dy", "dzl", "dzu"):
        assert relerr(g0[key], o0[key]) < 1e-6, key
    ctx.close()


def test_newton_solve_basis(kkt, po):
    m, n = 700, 1500
    B, _, colscale = basis_problem(m, n, seed=23, num_free=3, num_fixed=4)
    A = B["A"]
    N = n + m
    rng = np.random.default_rng(23)
    # iterate vectors consistent with colscale = 1/sqrt(zl/xl) (barrier-lb), inf (free), 0 (fixed)
    state = np.full(N, 2, dtype=np.uint8)
    state[np.isinf(colscale)] = 1
    state[colscale == 0.0] = 0
    zl = 10.0 ** rng.uniform(-1, 1, N)
    xl = colscale ** 2 * zl
    bar = state == 2
    xl[~bar] = np.inf; zl[~bar] = 0.0
    xu, zu = np.full(N, np.inf), np.zeros(N)
    fx = state == 0
    xl[fx] = xu[fx] = zl[fx] = zu[fx] = 0.0
    mu = float((xl[bar] * zl[bar]).mean())
    U = lambda k: rng.uniform(-0.5, 0.5, k)
    st = dict(state=state, xl=xl, xu=xu, zl=zl, zu=zu, mu=mu, rb=U(m), rc=U(N), rl=np.where(bar, U(N), 0.0),
              ru=np.zeros(N), sl=np.where(bar, mu - synth.xl_safe(xl) * zl, 0.0), su=np.zeros(N))
    tol = 1e-8
    ctx = kkt.KktContext(A)
    ctx.split_prepare(B["L"], B["U"], B["rowperm"], B["colperm"], B["basis"], B["status"], colscale)
    args = [st[k] for k in ("rb", "rc", "rl", "ru", "sl", "su", "xl", "xu", "zl", "zu", "state")]
    g = ctx.newton_solve(True, *args, tol, 500)
    AIm = A.with_identity()
    orc = po.Oracle()
    cs = lambda M: po.Csc(M.nrow, M.ncol, M.p, M.i, M.x)
    S = orc.split_prepare(cs(AIm), n, cs(B["L"]), cs(B["U"]), B["rowperm"], B["colperm"], B["basis"],
                          B["status"], colscale)
    o = S.newton_solve(*args, tol, 500)
    assert g["errflag"] == o["errflag"] == 0 and abs(g["iter"] - o["iter"]) <= 2
    for key in ("dx", "dxl", "dxu", "dy", "dzl", "dzu"):
        assert relerr(g[key], o[key]) < 1e-6, key
    check_newton_equations(AIm.to_scipy(), st, g, 1e-6)
    ctx.close()
