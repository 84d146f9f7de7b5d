"""CPU: the oracle against the reference's OWN objects (oracle/_ref), live, on fresh seeds.
Skipped where the reference build is absent (the GPU box only has the prebuilt files, which is
enough to run; a checkout without /root/reference and without oracle/_ref skips)."""
import numpy as np
import pytest

from helpers import basis_problem, diag_problem, relerr


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


@pytest.mark.parametrize("seed,m,n,num_dense", [(101, 120, 260, 0), (102, 333, 700, 0), (103, 400, 900, 3)])
def test_diag_path_bitwise(oracle, ref, po, seed, m, n, num_dense):
    from ipx_amd import synth
    A, st = diag_problem(m, n, seed=seed, num_dense=num_dense)
    v = synth.lp_vectors(m, n)
    Ao = po.Csc(m, n, A.p, A.i, A.x)
    rm = ref.model(Ao, v["rhs"], v["constr_type"], v["obj"], v["lb"], v["ub"])
    assert (rm.m, rm.n, rm.dualized) == (m, n, 0)
    AI = rm.AI()
    assert np.array_equal(AI.p[: n + 1], A.p) and np.array_equal(AI.x[: A.nnz], A.x)   # no rescaling
    nd, nzd = oracle.find_dense_columns(Ao)
    assert nd == rm.num_dense
    assert [rm.is_dense(j) for j in range(n)] == list(np.diff(A.p) >= nzd)
    AT, RT = oracle.transpose(po.Csc(m, n + m, AI.p, AI.i, AI.x)), rm.AIt()
    assert np.array_equal(AT.p, RT.p) and np.array_equal(AT.i, RT.i) and np.array_equal(AT.x, RT.x)
    W = st["xl"] / st["zl"]
    rhs = np.random.default_rng(seed).standard_normal(m)
    for Wv in (W, None):
        l1, d1 = rm.normal_apply(Wv, rhs)
        l2, d2 = oracle.normal_apply(Ao, Wv, rhs)
        assert np.array_equal(l1, l2) and d1 == d2
    k = rm.kkt_diag(maxiter=400)
    err, mu = k.factorize(np.ones(n + m), st["xl"], st["xu"], np.zeros(m), st["zl"], st["zu"])
    ko = oracle.kkt_diag(Ao, nzd, True, 400)
    assert ko.factorize(st["xl"], st["xu"], st["zl"], st["zu"], mu) == err == 0
    for tol in (0.3 * np.sqrt(mu), 1e-9):
        x1, y1, it1, e1 = k.solve(st["a"], st["b"], tol)
        x2, y2, it2, e2, _ = ko.solve(st["a"], st["b"], tol)
        assert e1 == e2
        if num_dense == 0:      # no LAPACK involved: bit for bit
            assert it1 == it2 and np.array_equal(x1, x2) and np.array_equal(y1, y2)
        else:                   # Cholesky rounding differs (in-repo dpotrf vs OpenBLAS)
            assert abs(it1 - it2) <= 2 and relerr(y1, y2) < 1e-6


def test_sparse_kernels_bitwise(oracle, ref, po):
    B, st, colscale = basis_problem(180, 400, seed=77, num_free=2, num_fixed=3)
    cs = lambda M: po.Csc(M.nrow, M.ncol, M.p, M.i, M.x)
    L, U, AI = cs(B["L"]), cs(B["U"]), cs(B["A"].with_identity())
    x0 = np.random.default_rng(1).standard_normal(180)
    for trans, uplo, unit, T in (("t", "u", 0, U), ("t", "l", 1, L), ("n", "l", 1, L), ("n", "u", 0, U)):
        a, na = ref.trisolve(T, x0, trans, uplo, unit)
        b, nb = oracle.trisolve(T, x0, trans, uplo, unit)
        assert np.array_equal(a, b) and na == nb
    assert np.array_equal(ref.forward_solve(L, U, x0), oracle.forward_solve(L, U, x0))
    assert np.array_equal(ref.backward_solve(L, U, x0), oracle.backward_solve(L, U, x0))
    assert np.array_equal(ref.inverse_perm(B["rowperm"]), oracle.inverse_perm(B["rowperm"]))
    nb_cols = np.nonzero(B["status"] == -1)[0]
    rpi = oracle.inverse_perm(B["rowperm"])
    N1 = ref.copy_permute_scale(AI, nb_cols, rpi, colscale[nb_cols])
    N2 = oracle.copy_permute_scale(AI, nb_cols, rpi, colscale[nb_cols])
    assert np.array_equal(N1.p, N2.p) and np.array_equal(N1.i, N2.i) and np.array_equal(N1.x, N2.x)
    z = np.zeros(180)
    assert np.array_equal(ref.add_normal_product(N1, None, x0, z), oracle.add_normal_product(N2, None, x0, z))
    assert ref.dot(x0, x0[::-1].copy()) == oracle.dot(x0, x0[::-1].copy())
    # the split operator + plain CR from explicit factors
    S = oracle.split_prepare(AI, 400, L, U, B["rowperm"], B["colperm"], B["basis"], B["status"], colscale)
    pre = S.get()
    Us = po.Csc(180, 180, U.p, U.i, pre["Ux"])
    R = ref.split(L, Us, pre["N"], pre["free_positions"])
    l1, d1 = R.apply(x0)
    l2, d2 = S.apply(x0)
    assert np.array_equal(l1, l2) and d1 == d2
    rhs = x0.copy()
    rhs[pre["free_positions"]] = 0.0
    y1, it1, e1, _ = R.cr_solve(rhs, 1e-9, -1)
    y2, it2, e2, _ = oracle.cr_solve(S.apply, rhs, 1e-9, None, -1)
    assert (it1, e1) == (it2, e2) and np.array_equal(y1, y2)


@pytest.mark.parametrize("seed,m,n", [(201, 90, 200), (202, 257, 610)])
def test_iterate_bitwise(oracle, ref, po, seed, m, n):
    """Iterate::Initialize states, Update, ComputeResiduals, ComputeComplementarity and StepToBoundary:
    the oracle against the reference's ipx::Iterate (bitwise)."""
    from ipx_amd import synth
    P = synth.synthetic_iterate(m, n, seed)
    A = P["A"]
    Ao = po.Csc(m, n, A.p, A.i, A.x)
    rm = ref.model(Ao, P["rhs"], P["constr_type"], P["obj"], P["lb"], P["ub"])
    assert (rm.m, rm.n, rm.dualized) == (m, n, 0)
    b, c, lbs, ubs = rm.vectors()
    assert np.array_equal(lbs, P["lbs"]) and np.array_equal(ubs, P["ubs"]) and np.array_equal(b, P["rhs"])
    ri = rm.iterate()
    ri.initialize(P["it"])
    assert np.array_equal(ri.states(), P["state"])
    r1 = ri.residuals()
    r2 = oracle.iterate_residuals(Ao, P["state"], b, c, lbs, ubs, P["it"])
    for key in ("rb", "rc", "rl", "ru"):
        assert np.array_equal(r1[key], r2[key]), key
    assert r1["presidual"] == r2["presidual"] and r1["dresidual"] == r2["dresidual"]
    assert ri.complementarity() == oracle.iterate_complementarity(P["state"], P["it"])
    # ComputeObjectives and the termination tests of IPM::Driver
    pobj, dobj, pobj_pp, dobj_pp = ri.objectives()
    po_, do_, off_ = oracle.iterate_objectives(Ao, P["state"], b, c, lbs, ubs, P["it"])
    assert (pobj, dobj) == (po_, do_) and (pobj_pp, dobj_pp) == (po_ + off_, do_ + off_)
    nb, nc = oracle.model_norms(m, n, b, c, lbs, ubs)
    for ftol, otol in ((1e-6, 1e-8), (1e3, 1e3), (1e3, 1e-8), (1e-6, 1e3)):
        feas = r2["presidual"] <= ftol * (1.0 + nb) and r2["dresidual"] <= ftol * (1.0 + nc)
        opt = abs(pobj_pp - dobj_pp) <= otol * (1.0 + abs(0.5 * (pobj_pp + dobj_pp)))
        assert ri.termination(ftol, otol) == (feas, opt, feas and opt)
    st = P["step"]
    for sp_, sd_, skip in ((0.7, 0.4, ()), (1.0, 1.0, ("dxu", "dzl")), (3.0, 5.0, ())):   # last: clamps at kBarrierMin
        args = {k: (None if k in skip else st[k]) for k in ("dx", "dxl", "dxu", "dy", "dzl", "dzu")}
        ri.initialize(P["it"])
        ri.update(sp_, args["dx"], args["dxl"], args["dxu"], sd_, args["dy"], args["dzl"], args["dzu"])
        got = oracle.iterate_update(m, n, P["state"], P["it"], sp_, args["dx"], args["dxl"], args["dxu"], sd_,
                                    args["dy"], args["dzl"], args["dzu"])
        want = ri.get()
        for key in want:
            assert np.array_equal(want[key], got[key]), key
    # StepToBoundary has no reference object to call (static in ipm.cc): sequential semantics by hand
    x, dx = P["it"]["xl"][P["state"] == 2], st["dxl"][P["state"] == 2]
    alpha, blk = oracle.step_to_boundary(x, dx)
    a, ib = 1.0, -1
    damp = 1.0 - np.finfo(float).eps
    for i in range(x.size):
        if x[i] + a * dx[i] < 0.0:
            a, ib = -(x[i] * damp) / dx[i], i
    assert alpha == a and blk == ib and np.all(x + alpha * dx >= 0.0)


def badly_scaled_lp(m, n, seed):
    """entries spanning ~40 binary orders of magnitude by rows and columns, some empty rows"""
    from ipx_amd import synth
    rng = np.random.default_rng(seed)
    A = synth.synthetic_lp(m, n, 5, seed)
    rs = 2.0 ** rng.integers(-20, 21, m)
    cs = 2.0 ** rng.integers(-20, 21, n)
    x = A.x * rs[A.i] * np.repeat(cs, np.diff(A.p)) * rng.uniform(0.7, 1.9, A.nnz)
    return synth.CscMatrix(m, n, A.p, A.i, x)


@pytest.mark.parametrize("seed,m,n", [(201, 150, 320), (202, 400, 850)])
def test_equilibrate_bitwise(oracle, ref, po, seed, m, n):
    """Presolver::EquilibrateMatrix: the restatement against the reference's own presolver (which ScaleModel
    runs inside PresolveModel, src/presolver.cc:266-292): scaled matrix, and the scaling factors as they show
    in the scaled objective (obj = 1 -> c = colscale) and right-hand side (rhs = 1 -> b = rowscale)"""
    from ipx_amd import synth
    A = badly_scaled_lp(m, n, seed)
    v = synth.lp_vectors(m, n)
    Ao = po.Csc(m, n, A.p, A.i, A.x)
    rm = ref.model(Ao, v["rhs"], v["constr_type"], v["obj"], v["lb"], v["ub"])
    if rm.dualized:
        pytest.skip("the reference dualizes this shape")
    x, cs, rs, rounds = oracle.equilibrate(Ao)
    assert rounds >= 1
    AI = rm.AI()
    assert np.array_equal(AI.i[: A.nnz], A.i) and np.array_equal(AI.x[: A.nnz], x)
    b, c, lb, ub = rm.vectors()
    assert np.array_equal(c[:n], cs) and np.array_equal(b, rs)
    # a matrix in range is left alone
    B = synth.synthetic_lp(m, n, 5, seed)
    x2, cs2, rs2, r2 = oracle.equilibrate(po.Csc(m, n, B.p, B.i, B.x))
    assert r2 == -1 and np.array_equal(x2, B.x) and np.all(cs2 == 1.0) and np.all(rs2 == 1.0)
