"""CPU: the LU restatement (oracle) under the reference's LuFactorization contract (src/lu_factorization.h:21-58).
The reference's kernel for it is BASICLU (absent: pivot order and values unpinned).  What the reference's own
code can say about a factorization it did not compute is checked here on every case: its
LuFactorization::Factorize stability estimate (src/lu_factorization.cc:87-127) below kLuStabilityThreshold, its
ForrestTomlin (src/forrest_tomlin.cc) solving with the factors, and the contract itself."""
import numpy as np
import pytest
import scipy.sparse as sp

from ipx_amd import synth

CASES = [dict(dim=300, bump=20), dict(dim=300, bump=0), dict(dim=500, bump=40, window=3),
         dict(dim=400, bump=30, num_dependent=2), dict(dim=200, bump=200, frac_rowsing=0.0),
         dict(dim=64, bump=5, frac_slack=1.0), dict(dim=1, bump=0), dict(dim=2500, bump=150, offdiag=3)]


def check_contract(G, F, pivottol=0.1):
    """B[rowperm,colperm] = (L+I)U with dependent columns replaced by unit columns; triangular shapes; sorted
    indices; U's diagonal last.  Returns the relative residual."""
    dim = G["dim"]
    L, U = F["L"], F["U"]
    Ls = sp.csc_matrix((L.x, L.i, L.p), shape=(dim, dim))
    Us = sp.csc_matrix((U.x, U.i, U.p), shape=(dim, dim))
    assert sorted(F["rowperm"]) == list(range(dim)) and sorted(F["colperm"]) == list(range(dim))
    for M, lower in ((L, True), (U, False)):
        cols = np.repeat(np.arange(dim), np.diff(M.p))
        assert np.all(M.i > cols) if lower else np.all(M.i <= cols)
        inner = np.ones(M.i.size, dtype=bool)
        inner[M.p[:-1][np.diff(M.p) > 0]] = False                # first entry of every column
        assert np.all(np.diff(M.i)[inner[1:]] > 0)                  # ascending inside a column
    assert np.all(U.i[U.p[1:] - 1] == np.arange(dim))               # diagonal last
    # copies: scipy sorts the indices of a matrix in place when it needs to, and shares `data` with the caller
    B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(dim, dim))
    Bp = B[F["rowperm"], :][:, F["colperm"]].tolil()
    for k in F["dependent"]:
        Bp[:, k] = 0
        Bp[k, k] = 1.0
    R = (Ls + sp.identity(dim)) @ Us - Bp.tocsc()
    return abs(R).max() / max(abs(B).max(), 1.0) if R.nnz else 0.0


@pytest.mark.parametrize("kw", CASES, ids=[str(i) for i in range(len(CASES))])
def test_oracle_lu_under_the_reference(oracle, ref, kw):
    G = synth.lp_like_basis_matrix(seed=3, **kw)
    dim = G["dim"]
    F = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
    assert check_contract(G, F) < 1e-13
    assert F["info"]["col_singletons"] + F["info"]["row_singletons"] + F["info"]["bump"] == dim
    assert F["info"]["bump"] == kw["bump"]                           # the planted triangular part is found
    assert len(F["dependent"]) == kw.get("num_dependent", 0)
    R = ref.lu(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], F)
    assert R.stability < 1e-12                                       # kLuStabilityThreshold, src/ipx_internal.h:33
    assert R.flag == (2 if len(F["dependent"]) else 0)               # LuUpdate::Factorize return code
    nb = G["Bp"][-1]
    assert R.fill_factor == pytest.approx((F["L"].nnz + F["U"].nnz) / nb)
    if R.flag == 0:
        B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(dim, dim))
        x = np.random.default_rng(0).standard_normal(dim)
        for trans in (False, True):
            y = R.solve_dense(x, trans)                              # the reference's ForrestTomlin::_SolveDense
            r = (B.T if trans else B) @ y - x
            assert np.abs(r).max() <= 1e-9 * (1 + np.abs(y).max())


def test_oracle_lu_unsorted_columns_and_strict_tolerance(oracle, ref):
    """indices need not be sorted (lu_factorization.h:36-37); strict_abs_pivottol uses kLuDependencyTol = 1e-3:
    a bump column whose entries are all below it is dependent"""
    G = synth.lp_like_basis_matrix(dim=200, bump=12, seed=9)
    rng = np.random.default_rng(1)
    Bp, Bi, Bx = G["Bp"], G["Bi"].copy(), G["Bx"].copy()
    for j in range(200):
        q = rng.permutation(Bp[j + 1] - Bp[j]) + Bp[j]
        Bi[Bp[j]:Bp[j + 1]], Bx[Bp[j]:Bp[j + 1]] = Bi[q], Bx[q]
    F0 = oracle.lu_factorize(200, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"])
    F1 = oracle.lu_factorize(200, Bp[:-1], Bp[1:], Bi, Bx)
    for key in ("rowperm", "colperm"):
        assert np.array_equal(F0[key], F1[key])
    assert np.array_equal(F0["U"].i, F1["U"].i) and np.array_equal(F0["U"].x, F1["U"].x)
    # scale one bump column down to 1e-5: dependent under the strict rule only
    F = oracle.lu_factorize(200, Bp[:-1], Bp[1:], Bi, Bx)
    jb = F["colperm"][200 - 3]                                       # a bump column (the bump is pivoted last)
    Bx2 = Bx.copy()
    Bx2[Bp[jb]:Bp[jb + 1]] *= 1e-5
    Fa = oracle.lu_factorize(200, Bp[:-1], Bp[1:], Bi, Bx2, 0.1, False)
    Fb = oracle.lu_factorize(200, Bp[:-1], Bp[1:], Bi, Bx2, 0.1, True)
    assert len(Fa["dependent"]) == 0 and len(Fb["dependent"]) >= 1
    G2 = dict(G, Bi=Bi, Bx=Bx2)
    assert check_contract(G2, Fb) < 1e-13
    Rb = ref.lu(200, Bp[:-1], Bp[1:], Bi, Bx2, Fb)
    assert (Rb.flag, Rb.stability < 1e-12) == (2, True), (Rb.flag, Rb.stability)


def test_bump_limit(oracle):
    """a bump beyond the limit is torn (spikes set aside until the rounds get through); refused only when the
    spikes themselves exceed the limit"""
    G = synth.lp_like_basis_matrix(dim=120, bump=30, seed=2)
    F = oracle.lu_factorize(120, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], bump_limit=30)
    assert F is not None and F["info"]["spikes"] == 0 and F["info"]["bump"] == 30
    assert oracle.lu_factorize(120, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], bump_limit=3) is None
    F = oracle.lu_factorize(120, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], bump_limit=29)   # a dense 30 x 30 bump: 29 spikes
    assert F is not None and 0 < F["info"]["spikes"] <= 29 and check_contract(G, F) < 1e-12


TORN = [dict(dim=300, num_exchanged=6, limit=30, bump=20), dict(dim=3000, num_exchanged=10, limit=20, bump=20),
        dict(dim=3000, num_exchanged=25, limit=64, bump=10, offdiag=3), dict(dim=2000, num_exchanged=12, limit=40, bump=30, window=5),
        dict(dim=60000, num_exchanged=40, limit=2048, bump=100, offdiag=3)]


@pytest.mark.parametrize("kw", TORN, ids=[str(i) for i in range(len(TORN))])
def test_oracle_lu_torn_bump_under_the_reference(oracle, ref, kw):
    """bases after a number of exchanges (a few columns replaced by random ones): the singleton rounds stall on a
    bump far beyond the dense limit; with the spikes torn off, the factorization satisfies the contract, the
    reference's LuFactorization::Factorize calls it stable and its ForrestTomlin solves with it"""
    kw = dict(kw)
    limit = kw.pop("limit")
    G = synth.disturbed_basis_matrix(seed=5, **kw)
    dim = G["dim"]
    plain = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], bump_limit=-1) if dim <= 3000 else None
    F = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], bump_limit=limit)
    assert F is not None
    inf = F["info"]
    assert 0 < inf["spikes"] <= limit and inf["bump"] == inf["spikes"] and inf["dependent"] == 0
    assert inf["col_singletons"] + inf["row_singletons"] + inf["bump"] == dim
    if plain is not None:
        assert plain["info"]["bump"] > limit and inf["spikes"] < plain["info"]["bump"] / 3
    R = ref.lu(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], F)
    assert R.stability < 1e-12 and R.flag == 0
    B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(dim, dim))
    x = np.random.default_rng(0).standard_normal(dim)
    for trans in (False, True):
        y = R.solve_dense(x, trans)
        r = (B.T if trans else B) @ y - x
        assert np.abs(r).max() <= 1e-8 * (1 + np.abs(y).max())
    if dim <= 3000:
        assert check_contract(G, F) < 1e-10


@pytest.mark.parametrize("kw", TORN, ids=[str(i) for i in range(len(TORN))])
def test_oracle_lu_elimination_rounds_under_the_reference(oracle, ref, kw):
    """the same bases with ELIMINATION ROUNDS in place of tearing (sets of low-Markowitz-cost pivots that form a diagonal
    block, eliminated together; the fill-in enters the current matrix): the contract, no dependent columns, every pivot
    accounted for; the reference's LuFactorization::Factorize calls the factors stable and its ForrestTomlin solves with
    them; and the factors hold no more entries than the torn ones do"""
    kw = dict(kw)
    limit = kw.pop("limit")
    G = synth.disturbed_basis_matrix(seed=5, **kw)
    dim = G["dim"]
    torn = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], bump_limit=limit)
    F = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], bump_limit=limit, sparse_min=max(8, limit // 4))
    assert F is not None
    inf = F["info"]
    assert inf["sparse_rounds"] > 0 and inf["sparse_pivots"] >= inf["sparse_rounds"] and inf["spikes"] == 0 and inf["dependent"] == 0
    assert inf["col_singletons"] + inf["row_singletons"] + inf["sparse_pivots"] + inf["bump"] == dim
    assert inf["bump"] <= limit
    assert F["L"].nnz + F["U"].nnz <= 1.05 * (torn["L"].nnz + torn["U"].nnz)
    R = ref.lu(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], F)
    assert R.stability < 1e-12 and R.flag == 0
    B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(dim, dim))
    x = np.random.default_rng(0).standard_normal(dim)
    for trans in (False, True):
        y = R.solve_dense(x, trans)
        r = (B.T if trans else B) @ y - x
        assert np.abs(r).max() <= 1e-8 * (1 + np.abs(y).max())
    if dim <= 3000:
        assert check_contract(G, F) < 1e-10


def test_oracle_lu_elimination_rounds_find_the_rank(oracle):
    """bases with misplaced columns at random positions are singular: the elimination rounds end with exactly as many
    dependent columns as the matrix lacks in rank (numpy), and the contract holds with the unit columns in place"""
    for seed in (1, 2, 3):
        G = synth.misplaced_basis_matrix(1500, 25, seed=seed, bump=40)
        dim = G["dim"]
        B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(dim, dim)).toarray()
        F = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], bump_limit=64, sparse_min=16)
        assert F["info"]["sparse_rounds"] > 0
        assert len(F["dependent"]) == dim - np.linalg.matrix_rank(B)
        assert check_contract(G, F) < 1e-10


def test_oracle_lu_default_policy_under_the_reference(oracle, ref):
    """the policy of round 5 (orc_lu_factorize_policy: elimination rounds for every bump of more than sparse_from rows, ended by the
    density of what is left / slow rounds / sparse_min, the rest dense): the contract, the reference's stability estimate and
    ForrestTomlin's solves -- with the defaults and with limits small enough that every end rule fires on these sizes"""
    G = synth.lp_like_basis_matrix(seed=3, dim=3000, bump=1400, bump_density=0.01)
    dim = G["dim"]
    B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(dim, dim))
    seen = set()
    for pol in ({}, dict(sparse_from=64, sparse_min=16, dense_at=0.05), dict(sparse_from=64, sparse_min=16, dense_at=0.0, slow_den=8),
                dict(sparse_from=64, sparse_min=1000, dense_at=0.0, slow_den=0), dict(sparse_from=64, sparse_min=16, fill_max=1, dense_at=0.6, slow_den=0, rest_limit=1200)):
        F = oracle.lu_factorize(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], policy=pol)
        assert F is not None, pol
        inf = F["info"]
        assert inf["sparse_rounds"] > 0 and inf["spikes"] == 0 and inf["dependent"] == 0
        assert inf["col_singletons"] + inf["row_singletons"] + inf["sparse_pivots"] + inf["bump"] == dim
        seen.add(inf["bump"])
        R = ref.lu(dim, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], F)
        assert R.stability < 1e-12 and R.flag == 0
        x = np.random.default_rng(0).standard_normal(dim)
        for trans in (False, True):
            y = R.solve_dense(x, trans)
            assert np.abs((B.T if trans else B) @ y - x).max() <= 1e-8 * (1 + np.abs(y).max())
        assert check_contract(G, F) < 1e-10
    assert len(seen) >= 4          # the end rules end the rounds at different points
