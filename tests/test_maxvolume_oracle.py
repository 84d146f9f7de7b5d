"""CPU: the Maxvolume restatement (oracle).  Maxvolume itself needs a live ipx::Basis (BASICLU) and cannot run in
the reference here: PARITY UNPINNED for the heuristic as a whole.  Pinned / checked instead:
  * the basis operations it drives -- tableau columns (FTRAN), rows of the inverse (BTRAN) and dense solves after
    a sequence of exchanges -- against the reference's own ForrestTomlin performing the same exchanges with its
    own update (oracle/_ref: src/forrest_tomlin.cc), although the restatement keeps product-form etas instead;
  * the defining property of every accepted exchange: the volume of the scaled basis grows by the logged factor
    (log2 |det(B D)| against a dense determinant), and every exchange passes the volume tolerance."""
import numpy as np
import pytest
import scipy.sparse as sp

from ipx_amd import synth


def setup(po, m, n, bump, seed, spread=1.0, num_free=0, num_fixed=0):
    P = synth.lp_like_basis(m, n, seed=seed, bump=bump)
    status = P["status"].copy()
    rng = np.random.default_rng(seed)
    if num_free:
        status[rng.choice(P["basis"], num_free, replace=False)] = 1
    if num_fixed:
        status[rng.choice(np.nonzero(status == -1)[0], num_fixed, replace=False)] = -2
    colscale = synth.synthetic_maxvolume_state(status, spread, seed)
    A = P["A"]
    return P, status, colscale, po.Csc(m, n, A.p, A.i, A.x)


def basis_matrix(A, basis):
    m = A.nrow
    AI = sp.hstack([sp.csc_matrix((A.x, A.i, A.p), shape=(m, A.ncol)), sp.identity(m)]).tocsc()
    return AI[:, basis]


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


@pytest.mark.parametrize("m,n,bump,seed", [(300, 700, 20, 4), (700, 1600, 40, 8)])
def test_basis_operations_against_reference_forrest_tomlin(oracle, ref, po, m, n, bump, seed):
    P, status, colscale, Ao = setup(po, m, n, bump, seed)
    B = oracle.basis(Ao, P["basis"], status, max_etas=7)         # several refactorizations on the way
    r = B.maxvolume(colscale, rows_per_slice=100)
    assert r["errflag"] == 0 and r["updates"] >= 10 and r["refused"] == 0
    # the reference's ForrestTomlin on the INITIAL basis (factors from the LU restatement), then the same exchanges
    G = P["G"]
    F = oracle.lu_factorize(m, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"])
    R = ref.lu(m, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], F)
    B2 = oracle.basis(Ao, P["basis"], status, max_etas=5)
    basis = P["basis"].copy()
    AI = Ao
    rng = np.random.default_rng(1)
    for k, (jb, jn) in enumerate(r["exchanges"]):
        pos = int(np.nonzero(basis == jb)[0][0])
        ci, cx = (AI.i[AI.p[jn]:AI.p[jn + 1]], AI.x[AI.p[jn]:AI.p[jn + 1]]) if jn < n else (np.array([jn - n]), np.array([1.0]))
        x_ref = R.ftran(ci, cx)                                   # FtranForUpdate
        x_orc = B2.solve_for_update(jn)
        assert np.abs(x_ref - x_orc).max() <= 1e-9 * (1 + np.abs(x_ref).max())
        y_ref = R.btran(pos)                                      # BtranForUpdate
        y_orc, row = B2.tableau_row(jb)
        assert np.abs(y_ref - y_orc).max() <= 1e-9 * (1 + np.abs(y_ref).max())
        assert abs(row[jn] - x_ref[pos]) <= 1e-9 * abs(x_ref[pos])   # the pivot, from the row and from the column
        assert R.update(x_ref[pos]) == 0
        err, exchanged = B2.exchange_if_stable(jb, jn, row[jn])
        assert err == 0 and exchanged
        basis[pos] = jn
        if k % 5 == 0:
            v = rng.standard_normal(m)
            for trans in ("N", "T"):
                a, b = R.solve_dense(v, trans == "T"), B2.solve_dense(v, trans)
                assert np.abs(a - b).max() <= 1e-9 * (1 + np.abs(a).max())
    assert np.array_equal(B2.get()[0], basis) and np.array_equal(B.get()[0], basis)
    assert B2.get()[2]["factorizations"] > 2


@pytest.mark.parametrize("m,n,bump,seed,free,fixed", [(120, 300, 8, 3, 0, 0), (200, 450, 15, 5, 3, 6)])
def test_maxvolume_grows_the_volume(oracle, po, m, n, bump, seed, free, fixed):
    P, status, colscale, Ao = setup(po, m, n, bump, seed, num_free=free, num_fixed=fixed)
    B = oracle.basis(Ao, P["basis"], status)
    r = B.maxvolume(colscale, volume_tol=2.0, rows_per_slice=50)
    assert r["errflag"] == 0 and r["updates"] > 0
    basis, status2, _ = B.get()
    # BASIC_FREE never leaves, NONBASIC_FIXED never enters (src/maxvolume.cc:24-33, :130-133)
    assert np.array_equal(np.nonzero(status2 == 1)[0], np.nonzero(status == 1)[0])
    assert np.array_equal(np.nonzero(status2 == -2)[0], np.nonzero(status == -2)[0])
    assert sorted(basis) == sorted(np.nonzero(status2 >= 0)[0]) and (status2 >= 0).sum() == m
    # volume of the scaled basis (free columns unscaled): log2|det| grows by volinc
    def logvol(bs):
        d = np.where(status[bs] == 1, 1.0, colscale[bs])
        sign, ld = np.linalg.slogdet(basis_matrix(Ao, bs).toarray() * d)
        assert sign != 0
        return ld / np.log(2.0)
    assert logvol(basis) - logvol(P["basis"]) == pytest.approx(r["volinc"], rel=1e-8, abs=1e-8)
    assert r["volinc"] >= r["updates"] * 1.0            # every exchange gained more than volume_tol = 2


@pytest.mark.parametrize("m,n,bump,seed,free,fixed", [(120, 300, 8, 3, 0, 0), (200, 450, 15, 5, 3, 6)])
def test_maxvolume_sequential_grows_the_volume(oracle, po, m, n, bump, seed, free, fixed):
    """Maxvolume::RunSequential (src/maxvolume.cc:14-106, update_heuristic == 0): same defining properties as the
    heuristic; after the last pass no NONBASIC column can enter with a volume gain above the tolerance"""
    P, status, colscale, Ao = setup(po, m, n, bump, seed, num_free=free, num_fixed=fixed)
    B = oracle.basis(Ao, P["basis"], status)
    r = B.maxvolume_sequential(colscale, volume_tol=2.0)
    assert r["errflag"] == 0 and r["updates"] > 0 and r["passes"] >= 2          # the last pass finds nothing
    basis, status2, _ = B.get()
    assert np.array_equal(np.nonzero(status2 == 1)[0], np.nonzero(status == 1)[0])
    assert np.array_equal(np.nonzero(status2 == -2)[0], np.nonzero(status == -2)[0])
    assert sorted(basis) == sorted(np.nonzero(status2 >= 0)[0]) and (status2 >= 0).sum() == m
    def logvol(bs):
        d = np.where(status[bs] == 1, 1.0, colscale[bs])
        sign, ld = np.linalg.slogdet(basis_matrix(Ao, bs).toarray() * d)
        assert sign != 0
        return ld / np.log(2.0)
    assert logvol(basis) - logvol(P["basis"]) == pytest.approx(r["volinc"], rel=1e-8, abs=1e-8)
    assert r["volinc"] >= r["updates"] * 1.0
    # the termination criterion, checked densely: every scaled tableau entry of a NONBASIC column is <= volume_tol
    Bm = basis_matrix(Ao, basis).toarray()
    AI = sp.hstack([sp.csc_matrix((Ao.x, Ao.i, Ao.p), shape=(m, n)), sp.identity(m)]).tocsc()
    nb = np.nonzero(status2 == -1)[0]
    T = np.linalg.solve(Bm, AI[:, nb].toarray())
    inv = np.where(status2[basis] == 0, 1.0 / colscale[basis], 0.0)
    V = np.abs(T) * inv[:, None] * colscale[nb][None, :]
    assert V.max() <= 2.0 * (1 + 1e-9)
