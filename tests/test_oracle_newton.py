"""CPU: the oracle's restatement of IPM::SolveNewtonSystem (reference src/ipm.cc:532-645) satisfies the
Newton equations.  The reference function cannot be linked here (ipm.cc needs ipx::Basis / BASICLU), so
this row of the oracle is parity-unpinned; the equations are what pins it."""
import numpy as np

from helpers import check_newton_equations
from ipx_amd import synth


def test_oracle_newton_equations(oracle):
    from oracle import pyoracle as po
    m, n = 300, 700
    A = synth.synthetic_lp(m, n, 8, 31)
    st = synth.synthetic_newton_state(m, n, 31, num_free=4, num_ub=25, num_boxed=30)
    tol = 0.3 * np.sqrt(st["mu"])
    k = oracle.kkt_diag(po.Csc(m, n, A.p, A.i, A.x), maxiter=500)
    assert k.factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"]) == 0
    args = [st[key] for key in ("rb", "rc", "rl", "ru", "sl", "su", "xl", "xu", "zl", "zu", "state")]
    o = k.newton_solve(*args, tol)
    assert o["errflag"] == 0 and o["iter"] > 0
    check_newton_equations(A.with_identity().to_scipy(), st, o, tol)
    # zero residuals, only complementarity targets (the corrector's call, ipm.cc:418-420)
    o0 = k.newton_solve(None, None, None, None, *args[4:], tol)
    st0 = dict(st, rb=np.zeros(m), rc=np.zeros(n + m), rl=np.zeros(n + m), ru=np.zeros(n + m))
    check_newton_equations(A.with_identity().to_scipy(), st0, o0, tol)


def test_oracle_ipm_step_properties(oracle):
    """IPM::Predictor/AddCorrector/StepSizes/MakeStep restated (src/ipm.cc:340-530): a few steps from an
    infeasible interior start keep the barrier variables positive, take steps in (0, 1) and reduce both
    residual norms."""
    from oracle import pyoracle as po
    m, n = 200, 480
    P = synth.synthetic_iterate(m, n, 45)
    A, state = P["A"], P["state"]
    Ao = po.Csc(m, n, A.p, A.i, A.x)
    b, c = P["rhs"], np.concatenate([P["obj"], np.zeros(m)])
    it = P["it"]
    k = oracle.kkt_diag(Ao, maxiter=2000)
    r0 = oracle.iterate_residuals(Ao, state, b, c, P["lbs"], P["ubs"], it)
    for _ in range(4):
        comp = oracle.iterate_complementarity(state, it)
        assert k.factorize(it["xl"], it["xu"], it["zl"], it["zu"], comp["mu"]) == 0
        it, info = k.ipm_step(state, b, c, P["lbs"], P["ubs"], it)
        assert info["errflag"] == 0 and 0.0 < info["step_primal"] < 1.0 and 0.0 < info["step_dual"] < 1.0
        assert 0.0 < info["sigma"] <= 1.0 + 1e-12
    lbm, ubm = (state == 2) | (state == 4), (state == 3) | (state == 4)
    assert (it["xl"][lbm] > 0).all() and (it["zl"][lbm] > 0).all() and (it["xu"][ubm] > 0).all() and (it["zu"][ubm] > 0).all()
    r1 = oracle.iterate_residuals(Ao, state, b, c, P["lbs"], P["ubs"], it)
    assert r1["presidual"] < r0["presidual"] and r1["dresidual"] < r0["dresidual"]
