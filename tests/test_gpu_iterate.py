"""GPU: the IPM iterate on the device (ipxk_iterate_*, ipxk_step_to_boundary) against the oracle, which
tests/test_oracle_vs_ref.py::test_iterate_bitwise pins bit for bit to the reference's ipx::Iterate, and
against the reference's Iterate itself where oracle/_ref is present.

Elementwise results (Update, rl, ru, the slack part of rc) and max/min reductions are bit-exact; rb and
rc are bit-exact in the phased SpMV layout (used at these sizes) and within 1e-12 otherwise; the
complementarity SUM is a parallel reduction (1e-13 relative)."""
import numpy as np
import pytest

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


@pytest.mark.parametrize("seed,m,n", [(301, 120, 300), (302, 2100, 5000)])
def test_iterate_vs_oracle(kkt, po, oracle, seed, m, n):
    P = synth.synthetic_iterate(m, n, seed)
    A, st, state = P["A"], P["step"], P["state"]
    Ao = po.Csc(m, n, A.p, A.i, A.x)
    b, c = P["rhs"], np.concatenate([P["obj"], np.zeros(m)])
    ctx = kkt.KktContext(A)
    ctx.iterate_set(P["it"], state)
    got = ctx.iterate_get()
    for key in got:
        assert np.array_equal(got[key], P["it"][key]), key
    r1 = ctx.iterate_residuals(b, c, P["lbs"], P["ubs"])
    r2 = oracle.iterate_residuals(Ao, state, b, c, P["lbs"], P["ubs"], P["it"])
    for key in ("rb", "rc", "rl", "ru"):
        assert np.array_equal(r1[key], r2[key]), key
    assert r1["presidual"] == r2["presidual"] and r1["dresidual"] == r2["dresidual"]
    c1, c2 = ctx.iterate_complementarity(), oracle.iterate_complementarity(state, P["it"])
    assert c1["mu_min"] == c2["mu_min"] and c1["mu_max"] == c2["mu_max"]
    assert abs(c1["complementarity"] - c2["complementarity"]) <= 1e-13 * c2["complementarity"]
    assert abs(c1["mu"] - c2["mu"]) <= 1e-13 * c2["mu"]
    for sp_, sd_, skip in ((0.7, 0.4, ()), (1.0, 1.0, ("dxu", "dzl")), (3.0, 5.0, ())):
        args = {k: (None if k in skip else st[k]) for k in ("dx", "dxl", "dxu", "dy", "dzl", "dzu")}
        ctx.iterate_set(P["it"], state)
        ctx.iterate_update(sp_, args["dx"], args["dxl"], args["dxu"], sd_, args["dy"], args["dzl"], args["dzu"])
        want = oracle.iterate_update(m, n, state, P["it"], sp_, args["dx"], args["dxl"], args["dxu"], sd_,
                                     args["dy"], args["dzl"], args["dzu"])
        got = ctx.iterate_get()
        for key in want:
            assert np.array_equal(want[key], got[key]), (key, sp_)
    # StepToBoundary on each of the four barrier vectors
    for xs, ds, mask in (("xl", "dxl", (state == 2) | (state == 4)), ("zu", "dzu", (state == 3) | (state == 4))):
        x, dx = P["it"][xs][mask], st[ds][mask]
        a1, b1 = ctx.step_to_boundary(x, dx)
        a2, b2 = oracle.step_to_boundary(x, dx)
        assert a1 == a2 and b1 == b2 and np.all(x + a1 * dx >= 0.0)
    assert ctx.step_to_boundary(np.ones(5), np.ones(5)) == (1.0, -1)     # nothing blocks
    ctx.close()


def test_iterate_vs_reference_object(kkt, po, ref):
    """the same through the reference's own ipx::Iterate (oracle/_ref)"""
    m, n = 400, 950
    P = synth.synthetic_iterate(m, n, 303)
    A = P["A"]
    rm = ref.model(po.Csc(m, n, A.p, A.i, A.x), P["rhs"], P["constr_type"], P["obj"], P["lb"], P["ub"])
    b, c, lbs, ubs = rm.vectors()
    ri = rm.iterate()
    ri.initialize(P["it"])
    state = ri.states()
    ctx = kkt.KktContext(A)
    ctx.iterate_set(P["it"], state)
    r1, r2 = ctx.iterate_residuals(b, c, lbs, ubs), ri.residuals()
    for key in ("rb", "rc", "rl", "ru"):
        assert np.array_equal(r1[key], r2[key]), key
    assert (r1["presidual"], r1["dresidual"]) == (r2["presidual"], r2["dresidual"])
    st = P["step"]
    ri.update(0.9, st["dx"], st["dxl"], st["dxu"], 0.8, st["dy"], st["dzl"], st["dzu"])
    ctx.iterate_update(0.9, st["dx"], st["dxl"], st["dxu"], 0.8, st["dy"], st["dzl"], st["dzu"])
    want, got = ri.get(), ctx.iterate_get()
    for key in want:
        assert np.array_equal(want[key], got[key]), key
    c1, c2 = ctx.iterate_complementarity(), ri.complementarity()
    assert c1["mu_min"] == c2["mu_min"] and c1["mu_max"] == c2["mu_max"] and abs(c1["mu"] - c2["mu"]) <= 1e-13 * c2["mu"]
    ctx.close()


def test_iterate_error_paths(kkt, monkeypatch):
    A = synth.synthetic_lp(60, 150, 8, 9)
    ctx = kkt.KktContext(A)
    with pytest.raises(kkt.KktError):
        ctx.iterate_update(1.0, None, None, None, 1.0, None, None, None)      # no iterate set
    with pytest.raises(kkt.KktError):
        ctx.iterate_complementarity()
    st = synth.synthetic_newton_state(60, 150, 9)
    with pytest.raises(kkt.KktError):                                         # solver not factorized
        ctx.newton_solve(False, *[st[k] for k in ("rb", "rc", "rl", "ru", "sl", "su", "xl", "xu", "zl", "zu",
                                                   "state")], 1e-3)
    # partitioned systems: the device iterate / Newton step are single-GPU offers
    monkeypatch.setenv("IPXK_FORCE_COMM", "1")
    ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    assert ctx.kkt_diag_factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"], precond_dense_cols=False) == 0
    with pytest.raises(kkt.KktError):
        ctx.newton_solve(False, *[st[k] for k in ("rb", "rc", "rl", "ru", "sl", "su", "xl", "xu", "zl", "zu",
                                                   "state")], 1e-3)
    ctx.close()


def test_golden_iterate_gpu(kkt):
    """the committed outputs of the reference's ipx::Iterate (tests/golden/iterate_150.npz)"""
    import os
    from ipx_amd.synth import CscMatrix
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "iterate_150.npz"))
    m, n = int(g["m"]), int(g["n"])
    ctx = kkt.KktContext(CscMatrix(m, n, g["Ap"], g["Ai"], g["Ax"]))
    it = {k: g["it_" + k] for k in ("x", "xl", "xu", "y", "zl", "zu")}
    st = {k: g["step_" + k] for k in ("dx", "dxl", "dxu", "dy", "dzl", "dzu")}
    ctx.iterate_set(it, g["state"])
    r = ctx.iterate_residuals(g["b"], g["c"], g["lbs"], g["ubs"])
    for key in ("rb", "rc", "rl", "ru"):
        assert np.array_equal(r[key], g[key]), key
    assert r["presidual"] == float(g["presidual"]) and r["dresidual"] == float(g["dresidual"])
    c = ctx.iterate_complementarity()
    assert c["mu_min"] == float(g["mu_min"]) and c["mu_max"] == float(g["mu_max"])
    assert abs(c["mu"] - float(g["mu"])) <= 1e-13 * float(g["mu"])
    for tag in "ab":
        ctx.iterate_set(it, g["state"])
        ctx.iterate_update(float(g["upd_%s_sp" % tag]), st["dx"], st["dxl"], st["dxu"], float(g["upd_%s_sd" % tag]),
                           st["dy"], st["dzl"], st["dzu"])
        got = ctx.iterate_get()
        for key in got:
            assert np.array_equal(got[key], g["upd_%s_%s" % (tag, key)]), (tag, key)
    ctx.close()
