"""GPU: the model upload and the gather-matrix layouts built on the device (ipx_amd/csrc/layout_device.hip)
against the host builders (ipx_amd/csrc/spmv.hip), array by array, and one device context per Model.

The reference constructs its KKT solvers for free (NormalMatrix stores a reference to the model,
src/normal_matrix.h:20-27; three solver objects per LpSolver::Solve, src/lp_solver.cc:375,386,457).  What
replaces that here is ONE upload + Transpose (src/sparse_matrix.cc:120-151) + layout build per model, done with
radix sorts on the device.  Index arithmetic is bit-exact: every array of the sliced and
the sorted layout must equal the host builder's, and the row-wise copy must equal scipy's / the oracle's
Transpose.  IPXK_LAYOUT_BUILD=host forces the host builders (the reference of this test)."""
import os

import numpy as np
import pytest

from ipx_amd import kkt, synth

pytestmark = pytest.mark.gpu

ARRAYS = {0: "sliced.tile_ptr", 1: "sliced.cnt", 2: "sliced.idx", 3: "sliced.val", 4: "sorted.sub_ptr", 5: "sorted.cnt", 6: "sorted.pack",
          7: "sorted.val", 11: "acc.tile_batch", 12: "acc.bptr", 13: "acc.pack", 14: "acc.val"}


def _ctx(A, build, env):
    env = dict(env, IPXK_LAYOUT_BUILD=build, IPXK_BUILD_ALL_LAYOUTS="1")       # also the sorted sub-tiles, which the accumulated tiles replace
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return kkt.KktContext(A, device=0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _compare(A, env, want_sorted=True):
    host = _ctx(A, "host", env)
    dev = _ctx(A, "device", env)
    try:
        for which in (0, 1):
            ih, _ = host.layout_info(which)
            idv, ms = dev.layout_info(which)
            assert idv["use_sliced"] == 1 and idv["sliced_built"] == 1 and idv["use_acc"] == 1 and idv["acc_built"] == 1, idv
            if want_sorted:
                assert idv["sorted_built"] == 1 or ih["sorted_built"] == 0
            # which of the two bit-identical layouts is in use is a timing decision; everything else must agree
            skip = {"use_sorted", "sorted_built", "so_nslices", "so_nsub", "so_nrb", "so_RB", "so_nrows_pad", "so_max_sub",
                    "so_slice_elems", "acc_deferred"} if ih["sorted_built"] != idv["sorted_built"] else {"use_sorted", "acc_deferred"}   # acc_deferred: a statistic (waiting events), counted differently by the two builders
            for k in ih:
                if k not in skip:
                    assert ih[k] == idv[k], (which, k, ih[k], idv[k])
            for a, name in ARRAYS.items():
                if 4 <= a <= 7 and not (ih["sorted_built"] and idv["sorted_built"]):
                    continue
                x, y = host.layout_array(which, a), dev.layout_array(which, a)
                assert x.shape == y.shape and x.size > 0, (which, name, x.shape, y.shape)
                assert np.array_equal(x, y), (which, name, int(np.flatnonzero(x != y)[0]))
        # the row-wise copy the device holds = Transpose of the CSC (ascending source column inside a row)
        S = A.to_scipy().tocsr()
        S.sort_indices()
        tp, ti, tx = dev.layout_array(1, 8), dev.layout_array(1, 9), dev.layout_array(1, 10)
        assert np.array_equal(tp, S.indptr) and np.array_equal(ti, S.indices) and np.array_equal(tx, S.data)
        # and the products agree bit for bit (same arrays, same kernels)
        rng = np.random.default_rng(5)
        W = rng.uniform(0.1, 10.0, A.nrow + A.ncol)
        y = rng.standard_normal(A.nrow)
        host.normal_prepare(W)
        dev.normal_prepare(W)
        l1, d1 = host.normal_apply(y)
        l2, d2 = dev.normal_apply(y)
        assert host.spmv_layout()[0] == dev.spmv_layout()[0] == ("acc", "acc")
        assert np.array_equal(l1, l2) and d1 == d2
        # the accumulated tiles against scipy (the oracle's order differs in association across slices only)
        S0 = A.to_scipy()
        ref = W[A.ncol:] * y + S0 @ (W[:A.ncol] * (S0.T @ y))
        assert np.abs(l2 - ref).max() <= 1e-12 * np.abs(ref).max()
        return dev.layout_info(0)[1]
    finally:
        host.close()
        dev.close()


def test_device_layouts_equal_host_layouts_small_slices():
    # small matrices made eligible for slicing (16 KiB slices): several row-block sizes, ragged last blocks
    for (m, n, seed) in ((9000, 20011, 1), (20000, 45000, 2), (33333, 70001, 3)):
        A = synth.synthetic_lp(m, n, 8, seed)
        _compare(A, {"IPXK_SLICE_TEST_KB": "16"})


def test_device_layouts_equal_host_layouts_unsorted_columns_and_duplicates_of_rows():
    # columns with descending row indices and columns of different lengths (1..20 entries): storage order is kept
    rng = np.random.default_rng(11)
    m, n = 12000, 26000
    lens = rng.integers(1, 21, n)
    Ap = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    Ai = np.empty(Ap[-1], np.int64)
    for j in range(n):
        rows = rng.choice(m, lens[j], replace=False)
        Ai[Ap[j]:Ap[j + 1]] = np.sort(rows)[::-1] if j % 2 else rows
    Ax = rng.uniform(0.5, 4.0, Ap[-1]) * rng.choice([-1.0, 1.0], Ap[-1])
    A = synth.CscMatrix(m, n, Ap, Ai, Ax)
    _compare(A, {"IPXK_SLICE_TEST_KB": "16"})


def test_device_layouts_equal_host_layouts_benchmark_size():
    # C3: 1M x 2M, 16M entries -- the size the create time is quoted on
    A = synth.synthetic_lp(1 << 20, 2 << 20, 8, 12345)
    _compare(A, {})
    c = kkt.KktContext(A, device=0)                  # default environment: what a solver object pays
    ms = c.layout_info(0)[1]
    assert c.spmv_layout()[0] == ("acc", "acc")
    c.close()
    total = sum(ms)
    print("ipxk_create at 1M x 2M: upload + transpose %.1f ms, A' layouts %.1f ms, A layouts %.1f ms, rest %.1f ms" % tuple(ms))
    assert total < 400.0, ms          # 3.2 s with the host builders (round 3); target 100 ms


def test_matrices_outside_the_device_path_take_the_host_builders():
    # x fits an XCD's L2 (no slicing), and a model with dense columns (long rows): both still build and solve
    A = synth.synthetic_lp(20000, 45000, 8, 4)
    c = kkt.KktContext(A, device=0)
    assert c.spmv_layout()[0][0] in ("phased", "fused", "sortedfused", "plain", "accfused")
    c.close()
    A = synth.synthetic_lp(30000, 70000, 8, 5, num_dense=4)
    c = kkt.KktContext(A, device=0)
    assert c.num_dense_cols == 4
    c.close()


def test_device_layouts_of_matrices_with_locality_equal_host_layouts():
    """matrices whose gathers have locality (banded) or whose gathered vector fits an XCD's L2: the fused tiles, the fused sorted
    tiles and the fused accumulated tiles built on the device against the host builders (forced one by one with
    IPXK_SPMV_LAYOUT), array by array; and the products of either context against the phased layout's, bit for bit"""
    cases = [synth.banded_lp(30000, 70000, 8, 2048, 7), synth.banded_lp(200000, 450000, 8, 4096, 8), synth.synthetic_lp(20000, 45000, 8, 4)]
    rng = np.random.default_rng(3)
    for A in cases:
        dev = _ctx(A, "device", {})
        W = rng.uniform(0.1, 10.0, A.nrow + A.ncol)
        y = rng.standard_normal(A.nrow)
        dev.normal_prepare(W)
        ld, dd = dev.normal_apply(y)
        try:
            for which in (0, 1):
                idv, _ = dev.layout_info(which)
                assert idv["use_sliced"] == 1 and idv["sliced_built"] == 1 and idv["nslices"] == 1, idv
            for layout, ids in (("fused", (0, 1, 2, 3)), ("sortedfused", (4, 5, 6, 7, 20)), ("accfused", (15, 16, 17, 18, 19)), ("phased", ())):
                host = _ctx(A, "host", {"IPXK_SPMV_LAYOUT": layout})
                try:
                    for which in (0, 1):
                        ih, _ = host.layout_info(which)
                        idv, _ = dev.layout_info(which)
                        built = {"fused": ih["sliced_built"], "sortedfused": ih["sorted_built"], "accfused": ih["accf_built"], "phased": 1}[layout]
                        dbuilt = {"fused": idv["sliced_built"], "sortedfused": idv["sorted_built"], "accfused": idv["accf_built"], "phased": 1}[layout]
                        assert built == dbuilt, (layout, which, ih, idv)
                        if not built:
                            continue
                        for a in ids:
                            x, z = host.layout_array(which, a), dev.layout_array(which, a)
                            assert x.shape == z.shape and x.size > 0 and np.array_equal(x, z), (layout, which, a, x.shape, z.shape)
                    host.normal_prepare(W)
                    lh, dh = host.normal_apply(y)
                    assert np.array_equal(lh, ld), layout            # every one of these layouts sums a row in storage order
                finally:
                    host.close()
        finally:
            dev.close()


def _lp_with_long_rows_and_columns(m, n, seed):
    """synthetic LP plus a ladder of longer and longer columns and rows (12 ... 6000 entries, each at most 1.5 x the one before, so
    that the reference's FindDenseColumns, src/model.cc, flags none of them): rows of more than 255 entries in both gather matrices,
    some of them longer than one segment of the long-row kernels (2048)"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    lens = [12]
    while lens[-1] < 6000:
        lens.append(int(lens[-1] * 1.5))
    rows, cols, vals = [], [], []
    for k, L in enumerate(lens):
        for rep in range(3):
            j = int(rng.integers(0, n))
            r = rng.choice(m, L, replace=False)
            rows += list(r); cols += [j] * L; vals += list(rng.uniform(0.5, 2.0, L) * rng.choice([-1.0, 1.0], L))
            i = int(rng.integers(0, m))
            c = rng.choice(n, L, replace=False)
            rows += [i] * L; cols += list(c); vals += list(rng.uniform(0.5, 2.0, L) * rng.choice([-1.0, 1.0], L))
    E = sp.coo_matrix((vals, (rows, cols)), shape=(m, n)).tocsc()
    T = (synth.synthetic_lp(m, n, 8, seed).to_scipy() + E).tocsc()
    T.sum_duplicates(); T.sort_indices()
    return synth.CscMatrix(m, n, T.indptr.astype(np.int64), T.indices.astype(np.int64), T.data.copy())


def test_long_rows_are_taken_out_on_the_device():
    """a model with rows of more than 255 entries in A and in A' (none of them a dense column by the reference's rule): the device builders
    take them out (layout_device.hip, device_strip_long_rows) instead of handing the model to the host builders; products and a KKT solve
    against the host-built context and scipy"""
    A = _lp_with_long_rows_and_columns(300000, 700000, 11)
    rng = np.random.default_rng(2)
    W = rng.uniform(0.1, 10.0, A.nrow + A.ncol)
    y = rng.standard_normal(A.nrow)
    S0 = A.to_scipy()
    ref = W[A.ncol:] * y + S0 @ (W[:A.ncol] * (S0.T @ y))
    st = synth.synthetic_ipm_state(A.nrow, A.ncol, 1.0, 3)
    out = {}
    for mode in ("device", "host"):
        env = {"IPXK_LONG_ROWS_HOST": "1"} if mode == "host" else {}
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            c = kkt.KktContext(A, device=0)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        try:
            assert c.num_dense_cols == 0
            info = [c.layout_info(w) for w in (0, 1)]
            assert info[0][0]["nlong"] > 0 and info[1][0]["nlong"] > 0, info
            c.normal_prepare(W)
            l, d = c.normal_apply(y)
            assert np.abs(l - ref).max() <= 1e-12 * np.abs(ref).max(), mode
            c.kkt_diag_factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"])
            x, yy, it, e, _ = c.kkt_diag_solve(st["a"], st["b"], 0.3 * np.sqrt(st["mu"]), 500)
            out[mode] = (l, it, e, yy, [i[0]["nlong"] for i in info], sum(info[0][1]), c.spmv_layout()[0])
        finally:
            c.close()
    d, h = out["device"], out["host"]
    assert d[4] == h[4], (d[4], h[4])
    assert np.abs(d[0] - h[0]).max() <= 1e-13 * np.abs(h[0]).max()
    assert d[2] == h[2] == 0 and abs(d[1] - h[1]) <= 1
    assert np.abs(d[3] - h[3]).max() <= 1e-6 * np.abs(h[3]).max()
    print("create with long rows: device builders %.1f ms (%s), host builders %.1f ms (%s)" % (d[5], d[6], h[5], h[6]))
    assert d[5] < 0.5 * h[5], (d[5], h[5])


def test_basis_operator_on_a_model_with_long_rows_built_on_the_device():
    """the basis-split operator C = I + inv(B) N N' inv(B') (N N' on the model matrix with masked values: the model has long rows, so N is
    not built as a matrix of its own) on the device-built layouts against the host-built ones and against scipy"""
    import scipy.sparse as sp
    m, n = 200000, 450000
    A0 = _lp_with_long_rows_and_columns(m, n, 12)
    B = synth.planted_lu_basis(A0, offdiag=3, seed=21)
    colscale = synth.synthetic_basis_state(B["status"], 1.0, 21)
    u = np.random.default_rng(4).standard_normal(m)
    out = {}
    for mode in ("device", "host"):
        env = {"IPXK_LONG_ROWS_HOST": "1"} if mode == "host" else {}
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            c = kkt.KktContext(B["A"], device=0)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        try:
            assert c.layout_info(0)[0]["nlong"] > 0 or c.layout_info(1)[0]["nlong"] > 0
            c.split_prepare(B["L"], B["U"], B["rowperm"], B["colperm"], B["basis"], B["status"], colscale)
            out[mode] = c.split_apply(u)[0]
        finally:
            c.close()
    assert np.abs(out["device"] - out["host"]).max() <= 1e-12 * np.abs(out["host"]).max()
    # scipy: C u = u + inv(Bs) N N' inv(Bs') u in pivot order; Bs = (L+I) U diag(s) with s the scaling of the basic columns in pivot order,
    # N = the nonbasic columns, scaled, rows in pivot order
    from scipy.sparse.linalg import spsolve_triangular
    AI = sp.hstack([B["A"].to_scipy(), sp.identity(m, format="csc")]).tocsc()
    s_piv = colscale[B["basis"][B["colperm"]]]
    nb = np.flatnonzero(B["status"] == -1)
    N = (AI[:, nb] @ sp.diags(colscale[nb])).tocsr()[B["rowperm"], :].tocsc()
    L1 = (B["L"].to_scipy() + sp.identity(m)).tocsr()
    U = B["U"].to_scipy().tocsr()
    w = spsolve_triangular(L1.T.tocsr(), spsolve_triangular(U.T.tocsr(), u / s_piv, lower=True), lower=False)
    t = N @ (N.T @ w)
    ref = u + spsolve_triangular(U, spsolve_triangular(L1, t, lower=True), lower=False) / s_piv
    assert np.abs(out["device"] - ref).max() <= 1e-9 * np.abs(ref).max()
