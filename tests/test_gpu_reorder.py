"""GPU: the locality-recovering renumbering of the model (ipx_amd/csrc/layout_device.hip, reorder_model; SURVEY.md section 7
"row/column reordering ... must stay a pure permutation").  Index arithmetic bit-exact, solves equal to the unpermuted ones."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import relerr
from ipx_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kkt():
    from ipx_amd import kkt as k
    k.load_library()
    assert k.load_library().ipxk_device_count() > 0, "no GPU visible"
    return k


def solve_both(kkt, A, st, monkeypatch, force=True):
    out = {}
    for mode in ("0", "1" if force else None):
        if mode is None:
            monkeypatch.delenv("IPXK_REORDER", raising=False)
        else:
            monkeypatch.setenv("IPXK_REORDER", mode)
        ctx = kkt.KktContext(A)
        info = ctx.reorder_info()
        assert ctx.kkt_diag_factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"]) == 0
        x, y, it, err, _ = ctx.kkt_diag_solve(st["a"], st["b"], 0.3 * np.sqrt(st["mu"]) * 1e-4, 500)
        perms = ctx.reordering() if info["levels"] > 0 and (info["active"] or mode == "1") else None
        ctx.close()
        out[mode] = (x, y, it, err, info, perms)
    return out


def test_renumbering_is_a_pure_permutation_and_the_solve_is_the_same(kkt, monkeypatch):
    """a banded LP with rows and columns shuffled, forced through the renumbered copy (IPXK_REORDER=1): rowperm / colperm are
    permutations, the renumbered matrix has its entries back within a few thousand rows of each other (the band is found), and
    the KKT solve returns the same x, y (1e-8 / 1e-6) in the same number of CR iterations (+-2) as on the model as given"""
    m, n = 20000, 41000
    A0 = synth.banded_lp(m, n, 8, 512, 5)
    A, _, _ = synth.shuffled(A0, 6)
    st = synth.synthetic_ipm_state(m, n, 1.0, 5)
    out = solve_both(kkt, A, st, monkeypatch)
    x0, y0, it0, e0, i0, _ = out["0"]
    x1, y1, it1, e1, i1, perms = out["1"]
    assert i0["active"] == 0 and i1["active"] == 1 and i1["levels"] > 8
    rp, cp = perms
    assert np.array_equal(np.sort(rp), np.arange(m)) and np.array_equal(np.sort(cp), np.arange(n))
    M = sp.csc_matrix((A.x, A.i, A.p), shape=(m, n))[rp][:, cp].tocsc()
    M.sort_indices()
    span = np.array([M.indices[M.indptr[j]:M.indptr[j + 1]].max() - M.indices[M.indptr[j]:M.indptr[j + 1]].min() for j in range(0, n, 97)])
    span_given = np.array([A.i[A.p[j]:A.p[j + 1]].max() - A.i[A.p[j]:A.p[j + 1]].min() for j in range(0, n, 97)])
    print("row span of a column: as given median %d, renumbered median %d (band of the unshuffled matrix: 512)" % (np.median(span_given), np.median(span)))
    assert np.median(span) < 4096 < np.median(span_given)
    assert e0 == e1 == 0 and abs(it0 - it1) <= 2, (it0, it1)
    assert relerr(y1, y0) < 1e-8 and relerr(x1, x0) < 1e-6        # (two solves stopped at the same tolerance, row sums in different orders)


def test_expander_is_left_alone(kkt, monkeypatch):
    """the uniformly random LP has nothing to recover: recognised within 8 levels, no copy, no change"""
    monkeypatch.delenv("IPXK_REORDER", raising=False)
    m, n = 150000, 300000                        # nnz 2.4 M: above the size from which a renumbering is looked for
    A = synth.synthetic_lp(m, n, 8, 7)
    ctx = kkt.KktContext(A)
    info = ctx.reorder_info()
    ctx.close()
    print(info)
    assert info["active"] == 0 and 0 < info["levels"] <= 12 and info["ms"] < 100


def test_renumbering_chosen_by_timing_at_scale(kkt, monkeypatch):
    """1M x 2M banded LP, shuffled: the renumbering is found, the renumbered copy's two products are faster than those on the model
    as given, so it is in use without being forced -- and the solve agrees with the one on the model as given"""
    m, n = 1000000, 2000000
    A0 = synth.banded_lp(m, n, 8, 4096, 12345)
    A, _, _ = synth.shuffled(A0, 7)
    st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
    out = solve_both(kkt, A, st, monkeypatch, force=False)
    x0, y0, it0, e0, i0, _ = out["0"]
    x1, y1, it1, e1, i1, _ = out[None]
    print(i1)
    assert i1["active"] == 1 and i1["us_reordered"] < 0.9 * i1["us_original"] and i1["ms"] < 500
    assert e0 == e1 == 0 and abs(it0 - it1) <= 2
    assert relerr(y1, y0) < 1e-6 and relerr(x1, x0) < 1e-5
