"""GPU: the device LU factorization (ipxk_lu_factorize*) through the C ABI
  * against the CPU restatement: permutations, patterns and dependent columns bit-exact, values bit-exact
    (the singleton part divides original entries by pivots; the dense bump applies its updates one pivot at a
    time in the restatement's order) -- for bumps of more than 1024 rows with IPXK_LU_MFMA_MIN=0: by default their
    trailing updates run on the matrix cores (v_mfma_f64_16x16x4_f64, fused accumulation), and the factors are
    held to the contract, the same pivots and 1e-11 against the restatement's values instead;
  * against the contract B[rowperm,colperm] = (L+I)U (src/lu_factorization.h:21-58);
  * under the reference's own objects: LuFactorization::Factorize's stability estimate and ForrestTomlin
    (oracle/_ref, where built);
  * Prepare straight from the device-resident factors against Prepare from the downloaded ones."""
import numpy as np
import pytest
import scipy.sparse as sp

from ipx_amd import synth
from test_lu_oracle import CASES, check_contract

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kkt():
    from ipx_amd import kkt as k
    k.load_library()
    assert k.load_library().ipxk_device_count() > 0, "no GPU visible"
    return k


@pytest.fixture(autouse=True)
def policy_of_round_4(monkeypatch, request):
    """the tests written for round 4's policy (tearing first for a bump beyond IPXK_LU_BUMP_MAX, dense as it stands below) keep it:
    IPXK_LU_SPARSE=t.  Tests of the default policy (elimination rounds for every bump of more than 1024 rows) carry the marker
    `default_policy`."""
    if "default_policy" in request.keywords:
        monkeypatch.delenv("IPXK_LU_SPARSE", raising=False)
    else:
        monkeypatch.setenv("IPXK_LU_SPARSE", "t")


@pytest.fixture(scope="module")
def ctx(kkt):
    c = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))     # the stand-alone entry point only needs a device
    yield c
    c.close()


def same_factors(F, Fo):
    for key in ("rowperm", "colperm", "dependent"):
        assert np.array_equal(F[key], Fo[key]), key
    for key in ("L", "U"):
        assert np.array_equal(F[key].p, Fo[key].p) and np.array_equal(F[key].i, Fo[key].i), key
        assert np.array_equal(F[key].x, Fo[key].x), key


BIG = [dict(dim=20000, bump=600, offdiag=3), dict(dim=5000, bump=97, window=4, frac_rowsing=0.4),
       dict(dim=3000, bump=1100, bump_density=0.05), dict(dim=700, bump=64, num_dependent=3),
       dict(dim=4000, bump=2300, bump_density=0.01)]


@pytest.mark.parametrize("kw", CASES + BIG, ids=[str(i) for i in range(len(CASES) + len(BIG))])
def test_lu_vs_oracle_and_contract(ctx, oracle, kw, monkeypatch):
    G = synth.lp_like_basis_matrix(seed=3, **kw)
    dim = G["dim"]
    Fo = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
    mfma = kw["bump"] > 1024           # the trailing updates of such a bump run on the matrix cores by default
    for mfma_min in (("0", None) if mfma else (None,)):
        if mfma_min is None:
            monkeypatch.delenv("IPXK_LU_MFMA_MIN", raising=False)
        else:
            monkeypatch.setenv("IPXK_LU_MFMA_MIN", mfma_min)
        F = ctx.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
        assert (F["col_singletons"], F["row_singletons"], F["bump"], F["num_dependent"]) == \
            (Fo["info"]["col_singletons"], Fo["info"]["row_singletons"], Fo["info"]["bump"], Fo["info"]["dependent"])
        assert F["bump"] == kw["bump"]
        if mfma and mfma_min is None:
            # fused accumulation: same pivots, values to rounding (entries that cancel exactly in one arithmetic need not in the other)
            for key in ("rowperm", "colperm", "dependent"):
                assert np.array_equal(F[key], Fo[key]), key
            for key in ("L", "U"):
                a = sp.csc_matrix((F[key].x, F[key].i, F[key].p), shape=(dim, dim))
                b = sp.csc_matrix((Fo[key].x, Fo[key].i, Fo[key].p), shape=(dim, dim))
                assert abs(a - b).max() <= 1e-11 * max(1.0, abs(b).max()), key
        else:
            same_factors(F, Fo)
        assert check_contract(G, F) < 1e-12


def test_lu_torn_bump_vs_oracle(kkt, oracle, monkeypatch):
    """a bump beyond the dense limit: spikes torn off, carried through the row singleton pivots on the device
    (forward substitution, 64 spikes at a time) -- the same spikes, pivot order and VALUES as the CPU restatement,
    bit for bit; the contract; at 60000 rows the planted chain structure leaves a bump of most of the matrix"""
    from test_lu_oracle import TORN
    for kw in TORN:
        kw = dict(kw)
        limit = kw.pop("limit")
        G = synth.disturbed_basis_matrix(seed=5, **kw)
        dim = G["dim"]
        monkeypatch.setenv("IPXK_LU_BUMP_MAX", str(limit))
        c = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
        F = c.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
        Fo = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, bump_limit=limit)
        assert (F["col_singletons"], F["row_singletons"], F["bump"], F["num_dependent"], F["spikes"], F["rounds"]) == \
            (Fo["info"]["col_singletons"], Fo["info"]["row_singletons"], Fo["info"]["bump"], Fo["info"]["dependent"],
             Fo["info"]["spikes"], Fo["info"]["rounds"]), kw
        assert F["spikes"] > 0
        same_factors(F, Fo)
        if dim <= 3000:
            assert check_contract(G, F) < 1e-10
        # a limit below the number of spikes: tearing alone refuses, loudly; by default the factorization then starts again with
        # elimination rounds and equals the restatement's (same pivots, same values)
        low = max(1, F["spikes"] // 2)
        monkeypatch.setenv("IPXK_LU_BUMP_MAX", str(low))
        monkeypatch.setenv("IPXK_LU_SPARSE", "0")
        with pytest.raises(RuntimeError, match="spikes"):
            c.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
        monkeypatch.setenv("IPXK_LU_SPARSE", "t")
        if dim <= 3000:
            Fs = c.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
            Fo = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, bump_limit=low, sparse_min=min(512, low))
            assert Fs["sparse_rounds"] == Fo["info"]["sparse_rounds"] > 0 and Fs["spikes"] == 0
            same_factors(Fs, Fo)
            assert check_contract(G, Fs) < 1e-10
        c.close()


def test_lu_elimination_rounds_vs_oracle(kkt, oracle, ref, monkeypatch):
    """IPXK_LU_SPARSE=1: a bump beyond the dense limit goes through elimination rounds (ipx_amd/csrc/lu.hip 2c) -- the same
    pivots in the same rounds, the same fill-in and the same VALUES as the CPU restatement, bit for bit (nonsingular bases
    after exchanges, and singular ones with misplaced columns); the contract; the reference's LuFactorization::Factorize
    calls the factors stable"""
    from test_lu_oracle import TORN
    monkeypatch.setenv("IPXK_LU_SPARSE", "1")
    monkeypatch.setenv("IPXK_LU_MFMA_MIN", "0")        # (a dense block of more than 1024 rows: the elimination's own arithmetic, not the matrix cores')
    c = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
    cases = [(synth.disturbed_basis_matrix(seed=5, **{k: v for k, v in kw.items() if k != "limit"}), kw["limit"], max(8, kw["limit"] // 4)) for kw in TORN]
    cases += [(synth.misplaced_basis_matrix(5000, 30, seed=3, bump=50), 200, 64), (synth.misplaced_basis_matrix(40000, 80, seed=4, bump=300), 1000, 256)]
    for G, limit, smin in cases:
        dim = G["dim"]
        monkeypatch.setenv("IPXK_LU_BUMP_MAX", str(limit))
        monkeypatch.setenv("IPXK_LU_SPARSE_MIN", str(smin))
        F = c.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
        Fo = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, bump_limit=limit, sparse_min=smin)
        assert (F["col_singletons"], F["row_singletons"], F["bump"], F["num_dependent"], F["sparse_pivots"], F["sparse_rounds"], F["rounds"]) == \
            (Fo["info"]["col_singletons"], Fo["info"]["row_singletons"], Fo["info"]["bump"], Fo["info"]["dependent"],
             Fo["info"]["sparse_pivots"], Fo["info"]["sparse_rounds"], Fo["info"]["rounds"]), (dim, limit)
        assert F["sparse_rounds"] > 0 and F["spikes"] == 0
        same_factors(F, Fo)
        if dim <= 5000:
            assert check_contract(G, F) < 1e-10
        if F["num_dependent"] == 0:
            R = ref.lu(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], F)
            assert R.stability < 1e-12 and R.flag == 0
    c.close()


def test_lu_elimination_rounds_random_bases(kkt, oracle, monkeypatch):
    """random sizes, kinds of bases (misplaced columns: singular; exchanged columns; a planted sparse bump), dense limits, end-of-rounds
    rules and seeds: the device's factors equal the restatement's bit for bit, also where the limit is never exceeded (no
    rounds: the dense code as it stands) -- the short form of scripts/gpu_sparse_lu_stress.py"""
    monkeypatch.setenv("IPXK_LU_SPARSE", "1")
    monkeypatch.setenv("IPXK_LU_MFMA_MIN", "0")
    c = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
    rng = np.random.default_rng(77)
    with_rounds = 0
    for case in range(12):
        dim, seed = int(rng.integers(1500, 9000)), int(rng.integers(1, 10**6))
        if case % 3 == 0:
            G = synth.misplaced_basis_matrix(dim, int(rng.integers(5, 40)), seed=seed, bump=int(rng.integers(10, 120)))
        elif case % 3 == 1:
            G = synth.disturbed_basis_matrix(seed=seed, dim=dim, num_exchanged=int(rng.integers(4, 25)), bump=int(rng.integers(10, 80)), offdiag=int(rng.integers(2, 4)))
        else:
            G = synth.lp_like_basis_matrix(dim=dim, bump=int(rng.integers(200, 700)), bump_density=float(rng.uniform(0.01, 0.05)), seed=seed)
        limit = int(rng.integers(16, 300))
        smin, slow = int(rng.integers(2, max(3, limit // 2))), int(rng.choice([0, 16, 256]))
        monkeypatch.setenv("IPXK_LU_BUMP_MAX", str(limit))
        monkeypatch.setenv("IPXK_LU_SPARSE_MIN", str(smin))
        monkeypatch.setenv("IPXK_LU_SPARSE_SLOW_DEN", str(slow))
        F = c.lu_factorize(G["dim"], G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
        Fo = oracle.lu_factorize(G["dim"], G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, bump_limit=limit, sparse_min=smin, slow_den=slow)
        assert Fo is not None and F["sparse_rounds"] == Fo["info"]["sparse_rounds"], (case, dim, limit)
        same_factors(F, Fo)
        with_rounds += F["sparse_rounds"] > 0
    assert with_rounds >= 6
    c.close()


def test_lu_elimination_rounds_bound_the_fill(kkt, ref, monkeypatch):
    """a 60000-row basis after 40 exchanges, default limits (dense block up to 8192 rows): tearing (the default, faster) and
    the elimination rounds both factorize it; the rounds keep nnz(L) + nnz(U) under 3.5 x nnz(B), and the reference calls
    both factorizations stable"""
    G = synth.disturbed_basis_matrix(seed=5, dim=60000, num_exchanged=40, bump=100, offdiag=3)
    dim, nb = G["dim"], len(G["Bi"])
    monkeypatch.setenv("IPXK_LU_BUMP_MAX", "2048")
    c = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
    fills = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("IPXK_LU_SPARSE", mode)
        F = c.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
        assert F["num_dependent"] == 0 and (F["spikes"] > 0) == (mode == "0") and (F["sparse_rounds"] > 0) == (mode == "1")
        fills[mode] = (F["lnz"] + F["unz"]) / nb
        R = ref.lu(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], F)
        assert R.stability < 1e-12 and R.flag == 0
    c.close()
    print("fill: torn %.2f, elimination rounds %.2f" % (fills["0"], fills["1"]))
    assert fills["1"] <= 3.2


@pytest.mark.parametrize("kw", [CASES[0], CASES[3], BIG[1]], ids=["plain", "singular", "big"])
def test_lu_under_the_reference(ctx, ref, kw):
    G = synth.lp_like_basis_matrix(seed=5, **kw)
    dim = G["dim"]
    F = ctx.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
    R = ref.lu(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], F)
    assert R.stability < 1e-12 and R.flag == (2 if F["num_dependent"] else 0)
    if R.flag == 0:
        B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(dim, dim))
        x = np.random.default_rng(0).standard_normal(dim)
        for trans in (False, True):
            y = R.solve_dense(x, trans)
            assert np.abs((B.T if trans else B) @ y - x).max() <= 1e-9 * (1 + np.abs(y).max())


def test_lu_unsorted_strict_and_limits(ctx, oracle, kkt, monkeypatch):
    G = synth.lp_like_basis_matrix(dim=400, bump=25, seed=9)
    rng = np.random.default_rng(1)
    Bp, Bi, Bx = G["Bp"], G["Bi"].copy(), G["Bx"].copy()
    for j in range(400):
        q = rng.permutation(Bp[j + 1] - Bp[j]) + Bp[j]
        Bi[Bp[j]:Bp[j + 1]], Bx[Bp[j]:Bp[j + 1]] = Bi[q], Bx[q]
    F = ctx.lu_factorize(400, Bp[:-1], Bp[1:], Bi, Bx)
    same_factors(F, oracle.lu_factorize(400, Bp[:-1], Bp[1:], Bi, Bx))
    jb = F["colperm"][400 - 3]
    Bx2 = Bx.copy()
    Bx2[Bp[jb]:Bp[jb + 1]] *= 1e-5
    for strict in (False, True):
        Fs = ctx.lu_factorize(400, Bp[:-1], Bp[1:], Bi, Bx2, 0.1, strict)
        same_factors(Fs, oracle.lu_factorize(400, Bp[:-1], Bp[1:], Bi, Bx2, 0.1, strict))
        assert (Fs["num_dependent"] >= 1) == strict
    # columns given as ranges of a larger array (Basis::Factorize passes AI's arrays), in any order
    order = rng.permutation(400)
    begin = np.zeros(400, np.int64)
    big_i, big_x, at = [], [], 0
    for j in order:
        begin[j] = at + 3
        big_i += [0, 0, 0] + list(Bi[Bp[j]:Bp[j + 1]])
        big_x += [9.0, 9.0, 9.0] + list(Bx[Bp[j]:Bp[j + 1]])
        at = len(big_i)
    end = begin + np.diff(Bp)
    same_factors(ctx.lu_factorize(400, begin, end, np.array(big_i), np.array(big_x)), F)
    # a bump over the limit is torn (same factors as the restatement with that limit); refused when even the
    # spikes exceed it; bad index, bad tolerance
    monkeypatch.setenv("IPXK_LU_BUMP_MAX", "24")
    Ft = ctx.lu_factorize(400, Bp[:-1], Bp[1:], Bi, Bx)
    assert 0 < Ft["spikes"] <= 24
    same_factors(Ft, oracle.lu_factorize(400, Bp[:-1], Bp[1:], Bi, Bx, bump_limit=24))
    monkeypatch.setenv("IPXK_LU_BUMP_MAX", "2")
    monkeypatch.setenv("IPXK_LU_SPARSE", "0")
    with pytest.raises(kkt.KktError, match="spikes"):
        ctx.lu_factorize(400, Bp[:-1], Bp[1:], Bi, Bx)
    monkeypatch.setenv("IPXK_LU_SPARSE", "t")             # by default the elimination rounds take over where tearing refuses
    Fs = ctx.lu_factorize(400, Bp[:-1], Bp[1:], Bi, Bx)
    assert Fs["sparse_rounds"] > 0 and Fs["bump"] <= 2
    same_factors(Fs, oracle.lu_factorize(400, Bp[:-1], Bp[1:], Bi, Bx, bump_limit=2, sparse_min=2))
    monkeypatch.delenv("IPXK_LU_BUMP_MAX")
    bad = Bi.copy()
    bad[7] = 400
    with pytest.raises(kkt.KktError, match="out of range"):
        ctx.lu_factorize(400, Bp[:-1], Bp[1:], bad, Bx)
    with pytest.raises(kkt.KktError, match="pivottol"):
        ctx.lu_factorize(400, Bp[:-1], Bp[1:], Bi, Bx, 0.0)
    # dimension 0 (src/basiclu_kernel.cc:39-46)
    F0 = ctx.lu_factorize(0, np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0))
    assert F0["lnz"] == F0["unz"] == 0 and F0["L"].p.tolist() == [0]


@pytest.mark.parametrize("m,n,bump", [(300, 700, 20), (6000, 13000, 250)])
def test_basis_factorize_and_prepare_on_device(kkt, oracle, m, n, bump):
    """ipxk_lu_factorize_basis takes B = AI[:, basis] from the resident matrix; ipxk_split_prepare_lu builds the
    split operator from the factors without a host round trip: same operator as ipxk_split_prepare on the
    downloaded factors, bit for bit"""
    P = synth.lp_like_basis(m, n, seed=4, bump=bump)
    colscale = synth.synthetic_basis_state(P["status"], 1.0, 4)
    ctx = kkt.KktContext(P["A"])
    F = ctx.lu_factorize_basis(P["basis"], 0.1)
    G = P["G"]
    same_factors(F, oracle.lu_factorize(m, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1))
    assert F["num_dependent"] == 0 and F["bump"] == bump
    ctx.split_prepare_lu(P["status"], colscale)
    rhs = np.random.default_rng(2).standard_normal(m)
    assert ctx.split_levels()[0] < (bump if bump >= 32 else 10 ** 9)      # the dense block is not a chain of levels
    lhs1, dot1 = ctx.split_apply(rhs)
    f1, b1 = ctx.forward_solve(rhs), ctx.backward_solve(rhs)
    ctx.split_prepare(F["L"], F["U"], F["rowperm"], F["colperm"], P["basis"], P["status"], colscale)
    lhs2, dot2 = ctx.split_apply(rhs)
    if bump < 32:
        assert np.array_equal(lhs1, lhs2) and dot1 == dot2
        assert np.array_equal(f1, ctx.forward_solve(rhs)) and np.array_equal(b1, ctx.backward_solve(rhs))
    else:
        # from the resident factors a bump of >= 32 rows is solved as a dense block between the sweeps of a pair
        # (blocked, inverted diagonal blocks) instead of a chain of levels: same operator, other rounding
        rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
        assert rel(lhs1, lhs2) < 1e-11 and abs(dot1 - dot2) <= 1e-11 * abs(dot2)
        assert rel(f1, ctx.forward_solve(rhs)) < 1e-11 and rel(b1, ctx.backward_solve(rhs)) < 1e-11
    # and it is the inverse of B: SolveDense
    AI = sp.hstack([P["A"].to_scipy(), sp.identity(m)]).tocsc()
    B = AI[:, P["basis"]]
    x = ctx.solve_dense(rhs, "n")
    assert np.abs(B @ x - rhs).max() <= 1e-9 * (1 + np.abs(x).max())
    # a singular basis is refused by Prepare
    basis2 = P["basis"].copy()
    dup = np.nonzero(basis2 < n)[0][:2]
    basis2[dup[1]] = basis2[dup[0]]                       # the same column twice
    ctx2 = kkt.KktContext(P["A"])
    F2 = ctx2.lu_factorize_basis(basis2, 0.1, download=False)
    assert F2["num_dependent"] >= 1
    with pytest.raises(kkt.KktError, match="dependent columns"):
        ctx2.split_prepare_lu(P["status"], colscale)
    ctx3 = kkt.KktContext(P["A"])
    with pytest.raises(kkt.KktError, match="ipxk_lu_factorize_basis"):
        ctx3.split_prepare_lu(P["status"], colscale)
    ctx.close(); ctx2.close(); ctx3.close()


def test_lu_and_maxvolume_at_baseline_size(kkt, ref):
    """BASELINE config 3 size (m = 1M, n = 2M) through size-independent properties: the LU of a nearly triangular
    basis satisfies the reference's own stability estimate and solves B x = b, B'y = b; Maxvolume from the slack
    basis brings exactly the variables with large scaling factors in, never refuses an exchange, gains volume with
    every exchange, and leaves a factorization that solves with the NEW basis"""
    import scipy.sparse as sp
    m, n, bump = 1000000, 2000000, 1000
    P = synth.lp_like_basis(m, n, seed=12345, bump=bump, offdiag=3)
    G = P["G"]
    ctx = kkt.KktContext(P["A"])
    F = ctx.lu_factorize_basis(P["basis"], 0.1)
    assert F["bump"] == bump and F["num_dependent"] == 0 and F["col_singletons"] + F["row_singletons"] + bump == m
    R = ref.lu(m, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], F)
    assert R.flag == 0 and R.stability < 1e-12                       # kLuStabilityThreshold
    colscale = synth.synthetic_basis_state(P["status"], 1.0, 12345)
    ctx.split_prepare_lu(P["status"], colscale)
    B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(m, m))
    rhs = np.random.default_rng(1).standard_normal(m)
    for trans in ("n", "t"):
        x = ctx.solve_dense(rhs, trans)
        assert np.abs((B.T if trans == "t" else B) @ x - rhs).max() <= 1e-9 * (1 + np.abs(x).max())
    ctx.close()
    # Maxvolume from the slack basis
    A = synth.synthetic_lp(m, n, 8, 12345)
    basis, status, colscale = synth.slack_basis_crash_state(m, n, 300, 1.0, 12345)
    ctx = kkt.KktContext(A)
    ctx.lu_factorize_basis(basis, 0.1, download=False)
    ctx.split_prepare_lu(status, colscale)
    r = ctx.maxvolume(status, colscale)
    assert r["errflag"] == 0 and r["refused"] == 0 and r["updates"] == 300 and r["slices"] == 105
    entered = r["exchanges"][:, 1]
    assert np.all(colscale[entered] >= 100.0) and len(set(entered)) == 300      # exactly the 300 large ones
    assert np.all(r["exchanges"][:, 0] >= n)                                     # slack variables left
    assert r["volinc"] >= 300.0                                                  # > volume_tol = 2 per exchange
    assert sorted(r["basis"]) == sorted(np.nonzero(r["status"] == 0)[0])
    AI = sp.hstack([A.to_scipy(), sp.identity(m, format="csc")]).tocsc()
    Bn = AI[:, r["basis"]]
    x = ctx.solve_dense(rhs, "n")
    assert np.abs(Bn @ x - rhs).max() <= 1e-9 * (1 + np.abs(x).max())
    ctx.close()


def test_dense_block_inverse_on_the_matrix_cores_against_the_blocked_solves(kkt, monkeypatch, capfd):
    """the explicit inverse of a dense block of the factors (trisolve.hip: cut_dense_block): triangular inverses by recursive
    doubling + one product on v_mfma_f64_16x16x4_f64 (dense_inverse.hip) against the older kernel -- one blocked solve per
    column of the identity -- on the same factors: operator applications and dense solves agree to 1e-9 (both inverses carry
    cond * eps).  Block sizes that
    are and are not multiples of 64, and a pair whose second block is short (1300 = 1024 + 276)."""
    from ipx_amd import synth
    for (m, n, bump) in ((40000, 90000, 1300), (20000, 45000, 640), (20000, 45000, 777)):
        P = synth.lp_like_basis(m, n, seed=5, bump=bump, offdiag=3)
        colscale = synth.synthetic_basis_state(P["status"], 1.0, 5)
        rhs = np.random.default_rng(2).standard_normal(m)
        out = {}
        monkeypatch.setenv("IPXK_VERBOSE", "1")
        for di_min in ("1", "0"):
            monkeypatch.setenv("IPXK_DENSE_INVERSE_MIN", di_min)
            ctx = kkt.KktContext(P["A"])
            F = ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
            assert F["bump"] == bump
            ctx.split_prepare_lu(P["status"], colscale)
            err = capfd.readouterr().err
            assert ("recursive doubling on the matrix cores" in err) == (di_min == "1"), err
            probes, rejected, worst = ctx.split_inverse_stats()
            assert probes >= 1 and rejected == 0 and worst < 1e-8, (probes, rejected, worst)
            out[di_min] = (ctx.split_apply(rhs)[0], ctx.solve_dense(rhs, "N"), ctx.solve_dense(rhs, "T"))
            ctx.close()
        a, b = out["1"], out["0"]
        for k in range(3):
            err = np.abs(a[k] - b[k]).max() / np.abs(b[k]).max()
            assert err <= 1e-9, (bump, k, err)
        assert not np.array_equal(a[0], b[0])


def test_dense_block_inverse_is_guarded(kkt, monkeypatch):
    """the explicit inverse of a dense block of the factors is probed like the inverted levels of the sweeps
    (|D (inverse z) - z| <= 1e-8 at Prepare); a block that fails keeps the blocked in-place solve.  A well conditioned
    block passes (residual ~ 1e-11); with IPXK_INVERSE_TOL=0 it is rejected and the operator falls back: same results
    to 1e-9 (the product with the inverse carries cond * eps, the blocked solve does not)."""
    from ipx_amd import synth
    m, n, bump = 40000, 90000, 1300
    P = synth.lp_like_basis(m, n, seed=5, bump=bump, offdiag=3)
    colscale = synth.synthetic_basis_state(P["status"], 1.0, 5)
    rhs = np.random.default_rng(2).standard_normal(m)
    out = {}
    for tol in ("default", "0"):
        if tol == "0":
            monkeypatch.setenv("IPXK_INVERSE_TOL", tol)
        else:
            monkeypatch.delenv("IPXK_INVERSE_TOL", raising=False)
        ctx = kkt.KktContext(P["A"])
        ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
        ctx.split_prepare_lu(P["status"], colscale)
        probes, rejected, worst = ctx.split_inverse_stats()
        assert probes >= 1 and worst < 1e-8 and (rejected >= 1) == (tol == "0"), (probes, rejected, worst)
        out[tol] = (ctx.split_apply(rhs)[0], ctx.solve_dense(rhs, "N"), ctx.solve_dense(rhs, "T"))
        ctx.close()
    for k in range(3):
        err = np.abs(out["0"][k] - out["default"][k]).max() / np.abs(out["default"][k]).max()
        assert err <= 1e-9, (k, err)
    assert not np.array_equal(out["0"][0], out["default"][0])


def test_dense_block_inverse_is_refined_before_it_is_rejected(kkt, monkeypatch):
    """an explicit inverse that fails the probe gets up to two refinement steps X += X (I - D X) on the matrix cores, and one
    that still misses the tolerance is held against the blocked solve's own residual on the same vectors (within four times
    that it stays: neither can do better on that block).  Forced here with a tolerance below the plain inverse's probe
    (1.3e-10 on this well conditioned block): three probes, the refined residual a fraction of the plain one, nothing rejected,
    the same solves to 1e-9.  (On the ill conditioned bases of an IPM -- scripts/gpu_lp_dropin_large.py 12000 30000 -- the plain
    probe is 1e-7 ... 3e-5 and one step brings it under 1e-8: without it every block fell back to the one-workgroup solve.)"""
    from ipx_amd import synth
    m, n, bump = 40000, 90000, 1300
    P = synth.lp_like_basis(m, n, seed=5, bump=bump, offdiag=3)
    colscale = synth.synthetic_basis_state(P["status"], 1.0, 5)
    rhs = np.random.default_rng(2).standard_normal(m)
    out = {}
    for tol in ("default", "1e-12"):
        if tol == "default":
            monkeypatch.delenv("IPXK_INVERSE_TOL", raising=False)
        else:
            monkeypatch.setenv("IPXK_INVERSE_TOL", tol)
        ctx = kkt.KktContext(P["A"])
        ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
        ctx.split_prepare_lu(P["status"], colscale)
        out[tol] = (ctx.split_inverse_stats(), ctx.solve_dense(rhs, "N"), ctx.solve_dense(rhs, "T"))
        ctx.close()
    (p0, r0, w0), (p1, r1, w1) = out["default"][0], out["1e-12"][0]
    assert (p0, r0) == (1, 0) and 1e-12 < w0 < 1e-8
    assert p1 == 3 and r1 == 0 and w1 < 0.5 * w0, (p1, r1, w1, w0)
    for k in (1, 2):
        assert np.abs(out["1e-12"][k] - out["default"][k]).max() <= 1e-9 * np.abs(out["default"][k]).max()


def test_dense_lu_launch_variants_give_identical_factors(kkt, monkeypatch):
    """the dense LU's launch structure does not change a bit of the factors: the outer panel by ONE launch of cooperating workgroups
    (default since round 5; 1 / 2 rows per thread) against the one-workgroup sub-panel launches (IPXK_LU_COOP=0) with the sub-panel's rows
    of U and its update of the rest of the outer panel in one launch or two (IPXK_LU_FUSED_SUB), each with and without the look-ahead
    (the late trailing update on a second, CU-masked stream; default from 6144 rows on, forced here for the smaller bumps too) -- bumps
    of 1500 / 2600 / 6500 / 9000 rows, i.e. 2 / 4 / 8 / 16 rows per thread in the sub-panel kernels (and 16, forced, on the two smaller
    ones), trailing update on the matrix cores"""
    c = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
    monkeypatch.setenv("IPXK_LU_BUMP_MAX", "20000")          # (the 9000-row bump dense as it stands)
    for dim, bump, dens in ((4000, 1500, 0.05), (6000, 2600, 0.02), (12000, 6500, 0.01), (12000, 9000, 0.004)):
        G = synth.lp_like_basis_matrix(dim=dim, bump=bump, bump_density=dens, seed=7)
        ref = None
        variants = (("c", "1"), ("c", "0"), ("1", "1"), ("0", "0"), ("1", "0"), ("0", "1"), ("w2", "1")) if bump < 5000 else (("c", "1"), ("c", "0"), ("1", "1"), ("0", "0"))
        for fused, look in variants:
            monkeypatch.delenv("IPXK_LU_PANEL_W", raising=False)
            monkeypatch.delenv("IPXK_LU_COOP", raising=False)
            if fused == "c":                  # the cooperative outer panel
                fused = "1"
            elif fused == "w2":               # sub-panels of 2 columns, 16 rows per thread: the panels of bumps beyond 8192 rows
                monkeypatch.setenv("IPXK_LU_PANEL_W", "2")
                fused = "1"
            else:
                monkeypatch.setenv("IPXK_LU_COOP", "0")
            monkeypatch.setenv("IPXK_LU_FUSED_SUB", fused)
            monkeypatch.setenv("IPXK_LU_LOOKAHEAD", look)
            F = c.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
            assert F["bump"] == bump and F["num_dependent"] == 0
            if ref is None:
                ref = F
                if bump < 5000:
                    assert check_contract(G, F) < 1e-9
            else:
                same_factors(F, ref)
    c.close()


def test_lu_bump_beyond_the_dense_limit_that_tearing_cannot_cut_down(kkt, monkeypatch):
    """a 14 000-row basis with a planted sparse bump of 9500 rows: the singleton rounds stall on it and tearing sets more than 9000
    columns aside -- more than the 8192 rows at which a bump is torn at all, but the spikes may fill the largest dense block the
    panel kernels take (16 384 rows: 16 rows per thread), so the basis is factorized instead of being refused as in round 3;
    Prepare inverts the block (refined if the probe asks for it) and B x = r, B' x = r are solved to 1e-9"""
    import scipy.sparse as sp
    m, n, bump = 14000, 32000, 9500
    P = synth.lp_like_basis(m, n, seed=5, bump=bump, offdiag=3, bump_density=0.01)
    colscale = synth.synthetic_basis_state(P["status"], 1.0, 5)
    ctx = kkt.KktContext(P["A"])
    F = ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
    assert 8192 < F["spikes"] == F["bump"] <= bump and F["sparse_rounds"] == 0 and F["num_dependent"] == 0       # (round 4's policy, see the fixture)
    ctx.split_prepare_lu(P["status"], colscale)
    probes, rejected, worst = ctx.split_inverse_stats()
    assert probes >= 1 and rejected == 0 and worst < 1e-8
    G = P["G"]
    B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(m, m))
    rhs = np.random.default_rng(1).standard_normal(m)
    for tr in ("N", "T"):
        x = ctx.solve_dense(rhs, tr)
        r = (B if tr == "N" else B.T) @ x - rhs
        assert np.abs(r).max() <= 1e-9 * (1 + np.abs(x).max()), tr
    ctx.close()


# ---- the default policy since round 5: elimination rounds for every bump of more than 1024 rows ------------------------------------
def ipm_basis_16000():
    """tests/golden/ipm_basis_16000.npz: a basis B = AI[:, basis] the reference's IPM (IPM::Driver over KKTSolverBasisHip, Maxvolume on
    the device) held in its 20th iteration on the 16000 x 40000 LP of tests/test_gpu_lp_dropin.general_lp(16000, 40000, 31); dumped on the
    MI355X with IPXK_LU_DUMP (scripts/gpu_lu_study.py).  11 400 structural columns of 8 entries + 4 600 slack columns, 95 905 entries."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ipm_basis_16000.npz"))
    return dict(dim=int(g["dim"]), Bp=g["Bp"].astype(np.int64), Bi=g["Bi"].astype(np.int64), Bx=g["Bx"])


@pytest.mark.default_policy
def test_lu_default_policy_vs_oracle(kkt, oracle, ref, monkeypatch):
    """the default policy against its CPU restatement (orc_lu_factorize_policy), bit for bit: which bumps go through the rounds, where
    the rounds end (density, slow rounds, sparse_min), the dense rest; the contract; the reference's stability estimate"""
    monkeypatch.setenv("IPXK_LU_MFMA_MIN", "0")        # (a dense rest of more than 1024 rows: the elimination's own arithmetic, not the matrix cores')
    c = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
    cases = [("exchanged", synth.disturbed_basis_matrix(seed=5, dim=6000, num_exchanged=40, bump=100, offdiag=3), {}),
             ("sparse bump 2300", synth.lp_like_basis_matrix(seed=3, dim=4000, bump=2300, bump_density=0.01), {}),
             ("denser bump 1100", synth.lp_like_basis_matrix(seed=3, dim=3000, bump=1100, bump_density=0.05), {}),
             ("small bump: dense as it stands", synth.lp_like_basis_matrix(seed=3, dim=5000, bump=600, offdiag=3), {}),
             ("misplaced (singular)", synth.disturbed_basis_matrix(seed=7, dim=5000, num_exchanged=60, bump=64, offdiag=2), {})]
    for name, G, pol in cases:
        dim, nb = G["dim"], len(G["Bi"])
        F = c.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
        Fo = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, policy=pol)
        assert Fo is not None, name
        assert (F["col_singletons"], F["row_singletons"], F["bump"], F["num_dependent"], F["sparse_pivots"], F["sparse_rounds"], F["rounds"]) == \
            (Fo["info"]["col_singletons"], Fo["info"]["row_singletons"], Fo["info"]["bump"], Fo["info"]["dependent"],
             Fo["info"]["sparse_pivots"], Fo["info"]["sparse_rounds"], Fo["info"]["rounds"]), name
        assert F["spikes"] == 0, name
        same_factors(F, Fo)
        assert check_contract(G, F) < 1e-10, name
        if ref is not None and F["num_dependent"] == 0:
            R = ref.lu(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], F)
            assert R.stability < 1e-12 and R.flag == 0, name
        print("%-32s dim %5d nnz(B) %6d: %5d sparse pivots in %3d rounds, dense rest %4d, fill %.2f" %
              (name, dim, nb, F["sparse_pivots"], F["sparse_rounds"], F["bump"], (F["lnz"] + F["unz"]) / nb))
    c.close()


@pytest.mark.default_policy
def test_lu_fill_on_an_ipm_basis(kkt, ref, monkeypatch):
    """a basis of the IPM on a random 16000 x 40000 LP (fixture): the default policy eliminates two thirds of the bump sparsely and
    leaves a dense rest of about 5000 rows -- nnz(L) + nnz(U) within 1.25 x what the SEQUENTIAL minimum-Markowitz elimination of
    the same pattern ends with (22.19 M entries = 231 x nnz(B), profiles/r05_lu_fill_study.txt; SuperLU / COLAMD on bases of this
    kind: 600 x) and well below what tearing leaves (round 4's policy, checked here too: 49 M on this basis, 95 M on others of the same run); both
    factorizations pass the reference's stability test and solve with the matrix to 1e-9"""
    G = ipm_basis_16000()
    dim, nb = G["dim"], len(G["Bi"])
    B = sp.csc_matrix((G["Bx"], G["Bi"], G["Bp"]), shape=(dim, dim))
    c = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
    out = {}
    for mode in (None, "t"):
        if mode is None:
            monkeypatch.delenv("IPXK_LU_SPARSE", raising=False)
        else:
            monkeypatch.setenv("IPXK_LU_SPARSE", mode)
        F = c.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, download=(mode is None))
        out[mode] = F
        print("policy %s: bump %d, %d sparse pivots in %d rounds, %d spikes, nnz(L)+nnz(U) %d = %.0f x nnz(B)" %
              (mode or "default", F["bump"], F["sparse_pivots"], F["sparse_rounds"], F["spikes"], F["lnz"] + F["unz"], (F["lnz"] + F["unz"]) / nb))
    F = out[None]
    assert F["num_dependent"] == 0 and F["sparse_rounds"] > 0 and F["spikes"] == 0
    assert F["lnz"] + F["unz"] <= 1.25 * 22186887
    assert F["bump"] <= 5400
    assert out["t"]["lnz"] + out["t"]["unz"] >= 1.5 * (F["lnz"] + F["unz"])
    if ref is not None:
        R = ref.lu(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], F)
        assert R.stability < 1e-12 and R.flag == 0
        x = np.random.default_rng(0).standard_normal(dim)
        for trans in (False, True):
            y = R.solve_dense(x, trans)
            assert np.abs((B.T if trans else B) @ y - x).max() <= 1e-9 * (1 + np.abs(y).max())
    c.close()


@pytest.mark.default_policy
def test_lu_bump_of_20000_rows_returns_factors(kkt, monkeypatch):
    """a 26 000-row basis with a planted sparse bump of 20 000 rows -- beyond the 16 384 rows of the largest dense block: round 4
    refused it (and LuKernelHip threw); the default policy eliminates it in rounds and the dense rest fits.  B x = r and B'x = r
    through Prepare + solve_dense to 1e-9."""
    m, n, bump = 26000, 60000, 20000
    P = synth.lp_like_basis(m, n, seed=6, bump=bump, offdiag=3, bump_density=0.0004)
    colscale = synth.synthetic_basis_state(P["status"], 1.0, 6)
    ctx = kkt.KktContext(P["A"])
    F = ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
    nb = len(P["G"]["Bi"])
    print("bump %d -> %d sparse pivots in %d rounds, dense rest %d, fill %.1f" % (bump, F["sparse_pivots"], F["sparse_rounds"], F["bump"], (F["lnz"] + F["unz"]) / nb))
    assert F["sparse_rounds"] > 0 and F["spikes"] == 0 and F["bump"] <= 32768 and F["num_dependent"] == 0
    ctx.split_prepare_lu(P["status"], colscale)
    G = P["G"]
    B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(m, m))
    rhs = np.random.default_rng(1).standard_normal(m)
    for tr in ("N", "T"):
        x = ctx.solve_dense(rhs, tr)
        r = (B if tr == "N" else B.T) @ x - rhs
        assert np.abs(r).max() <= 1e-9 * (1 + np.abs(x).max()), tr
    ctx.close()


def test_lu_dense_block_beyond_16384_rows(kkt, monkeypatch):
    """a dense block of 16 385 ... 32 768 rows (cooperative panel with 2 rows per thread, or 32 rows per thread in the one-workgroup
    sub-panels): a sparse planted bump of 21 000 rows sent to the dense code as it stands (round 4's policy with the limit raised) --
    also more rows than the LDS of a compute unit holds as doubles (20 416), so the one-workgroup blocked solve that stands in for a
    rejected inverse keeps its unknowns in a global scratch vector.  Prepare inverts the block on the matrix cores and B x = r,
    B' x = r are solved to 1e-9; with the inverse rejected (IPXK_INVERSE_TOL=0) the blocked solve gives the same to 1e-9."""
    monkeypatch.setenv("IPXK_LU_BUMP_MAX", "30000")
    m, n, bump = 22500, 50000, 21000
    P = synth.lp_like_basis(m, n, seed=4, bump=bump, offdiag=3, bump_density=0.0015)
    colscale = synth.synthetic_basis_state(P["status"], 1.0, 4)
    ctx = kkt.KktContext(P["A"])
    F = ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
    assert 20416 < F["bump"] <= bump and F["sparse_rounds"] == 0 and F["spikes"] == 0 and F["num_dependent"] == 0
    G = P["G"]
    B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(m, m))
    rhs = np.random.default_rng(1).standard_normal(m)
    for tol in (None, "0"):
        if tol is None:
            monkeypatch.delenv("IPXK_INVERSE_TOL", raising=False)
        else:
            monkeypatch.setenv("IPXK_INVERSE_TOL", tol)
        ctx.split_prepare_lu(P["status"], colscale)
        for tr in ("N", "T"):
            x = ctx.solve_dense(rhs, tr)
            r = (B if tr == "N" else B.T) @ x - rhs
            assert np.abs(r).max() <= 1e-9 * (1 + np.abs(x).max()), (tol, tr)
    probes, rejected, worst = ctx.split_inverse_stats()
    assert rejected >= 1          # (the second Prepare's inverse, by the tolerance of zero)
    ctx.close()
