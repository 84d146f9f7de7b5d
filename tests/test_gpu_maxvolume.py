"""GPU: Maxvolume on the device (ipxk_maxvolume) against the CPU restatement of
Maxvolume::RunHeuristic (oracle; its basis operations are pinned against the reference's ForrestTomlin in
tests/test_maxvolume_oracle.py): the same exchanges in the same order, the same final basis, the same counters;
the volume gained to 1e-9; and on return the context holds the operator of the NEW basis."""
import numpy as np
import pytest
import scipy.sparse as sp

from ipx_amd import synth
from test_maxvolume_oracle import basis_matrix, setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kkt():
    from ipx_amd import kkt as k
    k.load_library()
    assert k.load_library().ipxk_device_count() > 0, "no GPU visible"
    return k


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


@pytest.mark.parametrize("m,n,bump,seed,free,fixed,max_etas,rps", [
    (300, 700, 20, 4, 0, 0, 100, 100), (1200, 2600, 60, 8, 0, 0, 7, 100), (200, 450, 15, 5, 3, 6, 4, 50),
    (1000, 2300, 50, 11, 2, 5, 25, 300)])
def test_maxvolume_vs_oracle(kkt, oracle, po, m, n, bump, seed, free, fixed, max_etas, rps):
    P, status, colscale, Ao = setup(po, m, n, bump, seed, num_free=free, num_fixed=fixed)
    B = oracle.basis(Ao, P["basis"], status, max_etas=max_etas)
    want = B.maxvolume(colscale, volume_tol=2.0, maxskip_updates=10, rows_per_slice=rps)
    assert want["errflag"] == 0 and want["updates"] > 5
    ctx = kkt.KktContext(P["A"])
    ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
    ctx.split_prepare_lu(status, colscale)
    got = ctx.maxvolume(status, colscale, volume_tol=2.0, maxskip_updates=10, rows_per_slice=rps, max_etas=max_etas)
    assert got["errflag"] == 0
    assert np.array_equal(got["exchanges"], want["exchanges"])
    assert (got["updates"], got["skipped"], got["slices"], got["refused"]) == \
        (want["updates"], want["skipped"], want["slices"], want["refused"])
    assert got["volinc"] == pytest.approx(want["volinc"], rel=1e-9)
    basis_o, status_o, counts = B.get()
    assert np.array_equal(got["basis"], basis_o) and np.array_equal(got["status"], status_o)
    # (the final refactorization is left out when the last exchanges are cheaper to carry as etas behind the factors: kept_etas)
    assert got["kept_etas"] in (0, counts["etas"])
    assert got["factorizations"] == counts["factorizations"] - 1 + (1 if counts["etas"] > 0 and got["kept_etas"] == 0 else 0)
    # the context now holds the operator of the new basis (fresh factors, or the earlier ones with the etas behind them)
    Bm = basis_matrix(Ao, got["basis"])
    rhs = np.random.default_rng(3).standard_normal(m)
    x = ctx.solve_dense(rhs, "n")
    assert np.abs(Bm @ x - rhs).max() <= 1e-8 * (1 + np.abs(x).max())
    xt = ctx.solve_dense(rhs, "t")
    assert np.abs(Bm.T @ xt - rhs).max() <= 1e-8 * (1 + np.abs(xt).max())
    st = synth.synthetic_ipm_state(m, n, 1.0, seed)
    lhs1, dot1 = ctx.split_apply(rhs)
    x1, y1, it1, e1, _ = ctx.kkt_basis_solve(st["a"], st["b"], 1e-9, 2000)
    F = ctx.lu_factorize_basis(got["basis"], 0.1)
    ctx.split_prepare(F["L"], F["U"], F["rowperm"], F["colperm"], got["basis"], got["status"], colscale)
    lhs2, dot2 = ctx.split_apply(rhs)
    x2, y2, it2, e2, _ = ctx.kkt_basis_solve(st["a"], st["b"], 1e-9, 2000)
    if got["kept_etas"] == 0:
        # (a bump of >= 32 rows is solved as a dense block when the factors are resident: same operator, other rounding)
        assert np.abs(lhs1 - lhs2).max() <= 1e-10 * np.abs(lhs2).max() and abs(dot1 - dot2) <= 1e-10 * abs(dot2)
    # (with etas behind them the factors' pivot order is that of an EARLIER basis: the two operators act on differently ordered vectors;
    # what does not depend on the order is the KKT solve -- the same system through either operator)
    assert e1 == e2 == 0 and abs(it1 - it2) <= max(3, it2 // 10), (it1, e1, it2, e2)
    assert np.abs(x1 - x2).max() <= 1e-6 * np.abs(x2).max() and np.abs(y1 - y2).max() <= 1e-6 * np.abs(y2).max()
    ctx.close()


def test_maxvolume_preconditions_and_no_op(kkt, po, oracle):
    P, status, colscale, Ao = setup(po, 300, 700, 20, 4)
    ctx = kkt.KktContext(P["A"])
    with pytest.raises(kkt.KktError, match="ipxk_lu_factorize_basis"):
        ctx.maxvolume(status, colscale)
    ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
    with pytest.raises(kkt.KktError, match="ipxk_split_prepare_lu"):
        ctx.maxvolume(status, colscale)
    # a basis that already holds the large scaling factors: nothing to exchange, nothing refactorized
    good = synth.synthetic_basis_state(status, 1.0, 4)
    ctx.split_prepare_lu(status, good)
    got = ctx.maxvolume(status, good, volume_tol=2.0)
    want = oracle.basis(Ao, P["basis"], status).maxvolume(good, volume_tol=2.0)
    assert got["updates"] == want["updates"] and got["skipped"] == want["skipped"]
    if got["updates"] == 0:
        assert got["factorizations"] == 0 and np.array_equal(got["basis"], P["basis"])
    ctx.close()


@pytest.mark.parametrize("m,n,bump,seed,free,fixed,max_etas", [(300, 700, 20, 4, 0, 0, 100), (200, 450, 15, 5, 3, 6, 4),
                                                             (900, 2000, 40, 12, 2, 4, 25)])
def test_maxvolume_sequential_vs_oracle(kkt, oracle, po, m, n, bump, seed, free, fixed, max_etas):
    """Maxvolume::RunSequential (update_heuristic == 0) on the device against its CPU restatement: the same exchanges in
    the same order, the same final basis and counters, and the operator of the new basis in the context"""
    P, status, colscale, Ao = setup(po, m, n, bump, seed, num_free=free, num_fixed=fixed)
    B = oracle.basis(Ao, P["basis"], status, max_etas=max_etas)
    want = B.maxvolume_sequential(colscale, volume_tol=2.0)
    assert want["errflag"] == 0 and want["updates"] > 5
    ctx = kkt.KktContext(P["A"])
    ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
    ctx.split_prepare_lu(status, colscale)
    got = ctx.maxvolume_sequential(status, colscale, volume_tol=2.0, max_etas=max_etas)
    assert got["errflag"] == 0
    assert np.array_equal(got["exchanges"], want["exchanges"])
    assert (got["updates"], got["skipped"], got["passes"], got["refused"]) == \
        (want["updates"], want["skipped"], want["passes"], want["refused"])
    assert got["volinc"] == pytest.approx(want["volinc"], rel=1e-9)
    basis_o, status_o, _ = B.get()
    assert np.array_equal(got["basis"], basis_o) and np.array_equal(got["status"], status_o)
    Bm = basis_matrix(Ao, got["basis"])
    rhs = np.random.default_rng(3).standard_normal(m)
    x = ctx.solve_dense(rhs, "n")
    assert np.abs(Bm @ x - rhs).max() <= 1e-8 * (1 + np.abs(x).max())
    ctx.close()


def test_maxvolume_polls_the_interrupt_callback(kkt, oracle, po):
    """Control::InterruptCheck inside Maxvolume (src/maxvolume.cc:52,250): ipxk_set_interrupt's callback is polled once per
    candidate column; a nonzero value ends the run with that errflag, the exchanges made before it are kept and are
    the first ones of the uninterrupted run"""
    m, n = 300, 700
    P, status, colscale, Ao = setup(po, m, n, 20, 4)
    runs = {}
    for variant in ("heuristic", "sequential"):
        for stop_after in (None, 7):
            ctx = kkt.KktContext(P["A"])
            ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
            ctx.split_prepare_lu(status, colscale)
            polls = [0]

            def cb():
                polls[0] += 1
                return 998 if stop_after is not None and polls[0] > stop_after else 0
            ctx.set_interrupt(cb)
            got = ctx.maxvolume(status, colscale, volume_tol=2.0) if variant == "heuristic" else ctx.maxvolume_sequential(status, colscale, volume_tol=2.0)
            runs[(variant, stop_after)] = (got, polls[0])
            ctx.set_interrupt(None)
            ctx.close()
        full, cut = runs[(variant, None)], runs[(variant, 7)]
        assert full[0]["errflag"] == 0 and full[1] >= full[0]["updates"] > 7
        assert cut[0]["errflag"] == 998 and cut[1] == 8 and cut[0]["updates"] <= 7
        k = cut[0]["updates"]
        assert np.array_equal(np.asarray(cut[0]["exchanges"])[:k], np.asarray(full[0]["exchanges"])[:k])


@pytest.mark.parametrize("args", ["2000 5000 300 12345", "3000 7000 100 7", "20000 45000 400 3", "600 1500 60 5 1", "900 2000 150 11 1"])
def test_maxvolume_against_the_reference_itself(args):
    """oracle/_ref/test_maxvol_dropin (tests/dropin/maxvol_main.cc): the reference's own ipx::Maxvolume (RunHeuristic; with
    a fifth argument RunSequential) on the reference's ipx::Basis against ipxk_maxvolume / ipxk_maxvolume_sequential from
    the same slack basis, scaling factors and parameters -- the program fails unless the final bases agree in >= 99 % of
    their columns and the volume gained to 1e-6; measured on the MI355X: IDENTICAL final bases, update and skip counts,
    volume to 12 digits in every case"""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "test_maxvol_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/test_maxvol_dropin not built (needs the reference sources at build time)")
    r = subprocess.run([exe] + args.split(), capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0 and "DONE" in r.stdout and "PASS" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert "IDENTICAL decisions" in r.stdout, r.stdout


def test_dense_etas_equal_the_lists_with_hundreds_of_etas(kkt, po, monkeypatch):
    """the dense form of the eta file (blocked triangular solves: one wavefront per 64 etas) against the lists walked one eta after the
    other, with up to 1024 etas between refactorizations: the same exchanges in the same order, the same counters"""
    m, n, bump, seed = 6000, 14000, 300, 9
    P, status, colscale, Ao = setup(po, m, n, bump, seed, spread=2.0)
    runs = {}
    for form in ("1", "0"):
        monkeypatch.setenv("IPXK_MAXVOL_DENSE_ETAS", form)
        monkeypatch.setenv("IPXK_MAXVOL_KEEP_ETAS", "0")
        ctx = kkt.KktContext(P["A"])
        ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
        ctx.split_prepare_lu(status, colscale)
        got = ctx.maxvolume(status, colscale, volume_tol=2.0, maxskip_updates=10, rows_per_slice=1000, max_etas=1024)
        assert got["errflag"] == 0
        runs[form] = got
        ctx.close()
    d, l = runs["1"], runs["0"]
    print("updates %d, factorizations dense %d / lists %d" % (d["updates"], d["factorizations"], l["factorizations"]))
    assert d["updates"] > 1500, d["updates"]                   # (more than one refactorization apart: files of several hundred etas)
    assert np.array_equal(d["exchanges"], l["exchanges"]) and np.array_equal(d["basis"], l["basis"])
    assert (d["updates"], d["skipped"], d["refused"]) == (l["updates"], l["skipped"], l["refused"]) and d["volinc"] == l["volinc"]


def test_maxvolume_goes_on_with_the_etas_it_kept(kkt, oracle, po, monkeypatch):
    """two Maxvolume calls in a row with different scaling factors, the first one leaving its exchanges behind the factors as etas
    (IPXK_MAXVOL_KEEP_ETAS=1): the second call goes on with the same eta file, as the restatement's basis object does with its updates --
    the same exchanges in the same order in both calls, the same final basis; and the operator of the final basis solves the KKT system
    like the one built from a fresh factorization"""
    m, n, bump, seed = 1000, 2300, 50, 11
    P, status, colscale, Ao = setup(po, m, n, bump, seed)
    monkeypatch.setenv("IPXK_MAXVOL_KEEP_ETAS", "1")
    B = oracle.basis(Ao, P["basis"], status, max_etas=400)
    ctx = kkt.KktContext(P["A"])
    ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
    ctx.split_prepare_lu(status, colscale)
    want1 = B.maxvolume(colscale, volume_tol=2.0, maxskip_updates=10, rows_per_slice=300)
    got1 = ctx.maxvolume(status, colscale, volume_tol=2.0, maxskip_updates=10, rows_per_slice=300, max_etas=400)
    assert got1["errflag"] == 0 and got1["kept_etas"] > 0 and got1["updates"] > 5
    assert np.array_equal(got1["exchanges"], want1["exchanges"])
    # new scaling factors for the basis that came out (as the next interior-point iterate would bring them)
    basis_o, status_o, _ = B.get()
    assert np.array_equal(got1["basis"], basis_o) and np.array_equal(got1["status"], status_o)
    colscale2 = synth.synthetic_maxvolume_state(got1["status"], 1.0, seed + 100)
    ctx.split_rescale(got1["status"], colscale2)
    want2 = B.maxvolume(colscale2, volume_tol=2.0, maxskip_updates=10, rows_per_slice=300)
    got2 = ctx.maxvolume(got1["status"], colscale2, volume_tol=2.0, maxskip_updates=10, rows_per_slice=300, max_etas=400)
    assert got2["errflag"] == 0 and want2["updates"] > 5
    assert np.array_equal(got2["exchanges"], want2["exchanges"])
    assert (got2["updates"], got2["skipped"], got2["refused"]) == (want2["updates"], want2["skipped"], want2["refused"])
    basis_o, status_o, counts = B.get()
    assert np.array_equal(got2["basis"], basis_o) and np.array_equal(got2["status"], status_o)
    assert got2["kept_etas"] == counts["etas"] > got1["kept_etas"]          # one file over both calls, never refactorized in between
    # the operator of the final basis: SolveDense and the KKT solve against a fresh factorization
    Bm = basis_matrix(Ao, got2["basis"])
    rhs = np.random.default_rng(3).standard_normal(m)
    for tr, Mx in (("n", Bm), ("t", Bm.T)):
        x = ctx.solve_dense(rhs, tr)
        assert np.abs(Mx @ x - rhs).max() <= 1e-8 * (1 + np.abs(x).max())
    st = synth.synthetic_ipm_state(m, n, 1.0, seed)
    x1, y1, it1, e1, _ = ctx.kkt_basis_solve(st["a"], st["b"], 1e-9, 2000)
    ctx.lu_factorize_basis(got2["basis"], 0.1, download=False)
    ctx.split_prepare_lu(got2["status"], colscale2)
    x2, y2, it2, e2, _ = ctx.kkt_basis_solve(st["a"], st["b"], 1e-9, 2000)
    assert e1 == e2 == 0 and abs(it1 - it2) <= max(3, it2 // 10)
    assert np.abs(x1 - x2).max() <= 1e-6 * np.abs(x2).max() and np.abs(y1 - y2).max() <= 1e-6 * np.abs(y2).max()
    ctx.close()
