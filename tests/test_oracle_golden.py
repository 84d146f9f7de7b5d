"""CPU: the oracle (this repo's restatement) against the committed golden vectors that were
produced by the reference's own objects (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from helpers import relerr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def csc(po, d, prefix, nrow, ncol):
    return po.Csc(nrow, ncol, d[prefix + "p"], d[prefix + "i"], d[prefix + "x"])


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


def test_afiro_kkt_diag(oracle, po):
    d = load("afiro")
    m, n = int(d["m"]), int(d["n"])
    AI = csc(po, d, "AI", m, n + m)
    A = po.Csc(m, n, AI.p[: n + 1], AI.i[: AI.p[n]], AI.x[: AI.p[n]])
    # model upload format: the row-wise copy is Transpose(AI) -- bit-exact index arithmetic
    AIt = oracle.transpose(AI)
    assert np.array_equal(AIt.p, d["AItp"]) and np.array_equal(AIt.i, d["AIti"])
    assert np.array_equal(AIt.x, d["AItx"])
    k = oracle.kkt_diag(A, maxiter=-1)
    assert k.factorize() == int(d["fact_err"])
    x, y, it, err, _ = k.solve(d["a"], d["b"], float(d["tol"]))
    assert (it, err) == (int(d["iter"]), int(d["errflag"]))
    assert np.array_equal(x, d["x"]) and np.array_equal(y, d["y"])
    k.factorize(d["xl"], d["xu"], d["zl"], d["zu"], float(d["mu"]))
    x, y, it, err, _ = k.solve(d["a"], d["b"], float(d["tol2"]))
    assert (it, err) == (int(d["iter2"]), int(d["errflag2"]))
    assert np.array_equal(x, d["x2"]) and np.array_equal(y, d["y2"])


@pytest.mark.parametrize("name,exact", [("diag_200", True), ("dense_300", False)])
def test_diag_path(oracle, po, name, exact):
    d = load(name)
    m, n = int(d["m"]), int(d["n"])
    A = csc(po, d, "A", m, n)
    # dense-column classification (src/model.cc:34-56): bit-exact
    ndense, nz_dense = oracle.find_dense_columns(A)
    assert ndense == int(d["num_dense"])
    assert np.array_equal((np.diff(A.p) >= nz_dense).astype(np.int8), d["is_dense"])
    AIt = oracle.transpose(po.Csc(m, n, A.p, A.i, A.x))
    # reference AIt holds the identity entry last in each row (src/model.h:61); compare A's part
    rows = np.repeat(np.arange(m), np.diff(d["AItp"]))
    keep = d["AIti"] < n
    assert np.array_equal(AIt.i, d["AIti"][keep]) and np.array_equal(AIt.x, d["AItx"][keep])
    assert np.array_equal(np.bincount(rows[keep], minlength=m), np.diff(AIt.p))

    W, rhs = d["W"], d["rhs"]
    lhs, dot = oracle.normal_apply(A, W, rhs)
    assert np.array_equal(lhs, d["normal_lhs"]) and dot == float(d["normal_dot"])
    lhs, dot = oracle.normal_apply(A, None, rhs)
    assert np.array_equal(lhs, d["normal0_lhs"]) and dot == float(d["normal0_dot"])

    P, err = oracle.diag_factorize(A, W, nz_dense, True)
    assert err == int(d["prec_err"]) and P.num_dense == ndense
    pl, pd = P.apply(rhs)
    if exact:
        assert np.array_equal(pl, d["prec_lhs"]) and pd == float(d["prec_dot"])
    else:   # the in-repo Cholesky is not LAPACK's blocked dpotrf: rounding differs
        assert relerr(pl, d["prec_lhs"]) < 1e-12 and abs(pd - float(d["prec_dot"])) < 1e-12 * abs(pd)

    C = lambda v: oracle.normal_apply(A, W, v)
    y, it, e, hist = oracle.pcr_solve(C, P.apply, rhs, float(d["pcr_tol"]), d["resscale"], 500, hist_cap=600)
    assert e == int(d["pcr_err"])
    if exact:
        assert it == int(d["pcr_iter"]) and np.array_equal(y, d["pcr_y"])
    else:
        assert abs(it - int(d["pcr_iter"])) <= 2 and relerr(y, d["pcr_y"]) < 1e-6
    assert hist[-1] <= float(d["pcr_tol"]) and len(hist) == it + 1

    y, it, e, _ = oracle.pcr_solve(C, P.apply, rhs, 1e-30, d["resscale"], 7)
    assert (it, e) == (int(d["lim_iter"]), int(d["lim_err"])) == (7, 201)
    if exact:
        assert np.array_equal(y, d["lim_y"])

    Wn = d["Wneg"]
    Pn, _ = oracle.diag_factorize(A, Wn, nz_dense, False)
    _, it, e, _ = oracle.pcr_solve(lambda v: oracle.normal_apply(A, Wn, v), Pn.apply, rhs, 1e-12, None, 200)
    assert (it, e) == (int(d["neg_iter"]), int(d["neg_err"]))

    k = oracle.kkt_diag(A, maxiter=500)
    assert k.factorize(d["xl"], d["xu"], d["zl"], d["zu"], float(d["mu"])) == int(d["kkt_fact_err"])
    x, yk, it, e, _ = k.solve(d["a"], d["b"], float(d["kkt_tol"]))
    assert e == int(d["kkt_err"])
    if exact:
        assert it == int(d["kkt_iter"])
        assert np.array_equal(x, d["kkt_x"]) and np.array_equal(yk, d["kkt_y"])
    else:
        assert abs(it - int(d["kkt_iter"])) <= 2
        assert relerr(x, d["kkt_x"]) < 1e-6 and relerr(yk, d["kkt_y"]) < 1e-6


def test_basis_path(oracle, po):
    d = load("basis_200")
    m, n = int(d["m"]), int(d["n"])
    A = csc(po, d, "A", m, n)
    L, U = csc(po, d, "L", m, m), csc(po, d, "U", m, m)
    x0 = d["x0"]
    assert np.array_equal(oracle.inverse_perm(d["rowperm"]), d["rowperm_inv"])
    LT = oracle.transpose(L)
    assert np.array_equal(LT.p, d["LTp"]) and np.array_equal(LT.i, d["LTi"]) and np.array_equal(LT.x, d["LTx"])
    for trans, uplo, unit, key, T in (("t", "u", 0, "Ut", U), ("t", "l", 1, "Lt", L),
                                      ("n", "l", 1, "Lf", L), ("n", "u", 0, "Uf", U)):
        xs, _ = oracle.trisolve(T, x0, trans, uplo, unit)
        assert np.array_equal(xs, d["tri_" + key]), key
    assert np.array_equal(oracle.forward_solve(L, U, x0), d["fwd"])
    assert np.array_equal(oracle.backward_solve(L, U, x0), d["bwd"])

    AI = po.Csc(m, n + m, np.concatenate([A.p, A.p[-1] + 1 + np.arange(m)]),
                np.concatenate([A.i, np.arange(m)]), np.concatenate([A.x, np.ones(m)]))
    S = oracle.split_prepare(AI, n, L, U, d["rowperm"], d["colperm"], d["basis"], d["status"], d["colscale"])
    pre = S.get()
    # Prepare's outputs (splitted_normal_matrix.cc:18-66): index maps bit-exact, scaled values equal
    assert np.array_equal(pre["rowperm_inv"], d["rowperm_inv"])
    assert np.array_equal(pre["free_positions"], d["free_positions"])
    assert np.array_equal(pre["N"].p, d["Np"]) and np.array_equal(pre["N"].i, d["Ni"])
    assert np.array_equal(pre["N"].x, d["Nx"]) and np.array_equal(pre["Ux"], d["Ux_scaled"])
    lhs, dot = S.apply(x0)
    assert np.array_equal(lhs, d["split_lhs"]) and dot == float(d["split_dot"])
    y, it, e, hist = oracle.cr_solve(S.apply, d["cr_rhs"], float(d["cr_tol"]), None, -1, hist_cap=300)
    assert (it, e) == (int(d["cr_iter"]), int(d["cr_err"])) and np.array_equal(y, d["cr_y"])
    _, it, e, _ = oracle.cr_solve(S.apply, d["cr_rhs"], 1e-30, None, 5)
    assert (it, e) == (int(d["cr_lim_iter"]), int(d["cr_lim_err"])) == (5, 201)


def test_kkt_basis_solve_property(oracle, po):
    """a14 (KKTSolverBasis::_Solve) cannot be run in the reference here (needs BASICLU): pin the
    restatement through the KKT system it must solve (src/kkt_solver_basis.cc:69-74)."""
    d = load("basis_200")
    m, n = int(d["m"]), int(d["n"])
    A = csc(po, d, "A", m, n)
    L, U = csc(po, d, "L", m, m), csc(po, d, "U", m, m)
    AI = po.Csc(m, n + m, np.concatenate([A.p, A.p[-1] + 1 + np.arange(m)]),
                np.concatenate([A.i, np.arange(m)]), np.concatenate([A.x, np.ones(m)]))
    status, colscale = d["status"], d["colscale"]
    S = oracle.split_prepare(AI, n, L, U, d["rowperm"], d["colperm"], d["basis"], status, colscale)
    # SolveDense really inverts B = AI[:, basis]
    Bm = AI.to_scipy()[:, d["basis"]]
    r = np.random.default_rng(5).standard_normal(m)
    assert relerr(Bm @ S.solve_dense(r, "N"), r) < 1e-10
    assert relerr(Bm.T @ S.solve_dense(r, "T"), r) < 1e-10
    a, b = d["a"], d["b"]
    x, y, it, e, _ = S.kkt_solve(a, b, 1e-10)
    assert e == 0
    AIs = AI.to_scipy()
    fixed, free = status == -2, status == 1
    assert np.all(x[fixed] == 0.0)
    # primal equation AI x = b holds exactly up to round-off (x_B = inv(B)(b - N x_N))
    assert relerr(AIs @ x, b) < 1e-9
    # dual equation on barrier variables: x_j/d_j^2 + AI_j'y = a_j + res_j, res small; free: AI_j'y = a_j
    g = AIs.T @ y
    bar = ~fixed & ~free
    res = x[bar] / colscale[bar] ** 2 + g[bar] - a[bar]
    assert np.abs(res * colscale[bar]).max() < 1e-8
    assert np.abs(g[free] - a[free]).max() < 1e-9


def test_golden_iterate(oracle):
    """ipx::Iterate of the reference (states, ComputeResiduals, ComputeComplementarity, Update incl. the
    kBarrierMin truncation) -- the oracle reproduces the committed outputs bit for bit."""
    from oracle import pyoracle as po
    g = np.load(os.path.join(GOLD, "iterate_150.npz"))
    m, n = int(g["m"]), int(g["n"])
    A = po.Csc(m, n, g["Ap"], g["Ai"], g["Ax"])
    it = {k: g["it_" + k] for k in ("x", "xl", "xu", "y", "zl", "zu")}
    st = {k: g["step_" + k] for k in ("dx", "dxl", "dxu", "dy", "dzl", "dzu")}
    r = oracle.iterate_residuals(A, g["state"], g["b"], g["c"], g["lbs"], g["ubs"], it)
    for key in ("rb", "rc", "rl", "ru"):
        assert np.array_equal(r[key], g[key]), key
    assert r["presidual"] == float(g["presidual"]) and r["dresidual"] == float(g["dresidual"])
    c = oracle.iterate_complementarity(g["state"], it)
    assert (c["complementarity"], c["mu"], c["mu_min"], c["mu_max"]) == tuple(
        float(g[k]) for k in ("complementarity", "mu", "mu_min", "mu_max"))
    for tag in "ab":
        got = oracle.iterate_update(m, n, g["state"], it, float(g["upd_%s_sp" % tag]), st["dx"], st["dxl"],
                                    st["dxu"], float(g["upd_%s_sd" % tag]), st["dy"], st["dzl"], st["dzu"])
        for key in got:
            assert np.array_equal(got[key], g["upd_%s_%s" % (tag, key)]), (tag, key)
    assert (g["upd_b_xl"] == 1e-30).any()          # the fixture does exercise the truncation
