"""CPU, world_size 2 (gloo): the row partition used for multi-GPU runs (ipx_amd/partition.py,
SURVEY.md section 8e).  Each rank holds a slab of rows; one all-reduce of the n-vector per
NormalMatrix apply and all-reduced scalars in the CR loop reproduce the unpartitioned solve.
The local arithmetic here is plain numpy/scipy (the GPU kernels need a GPU); the oracle is only
the checker."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _partitioned_pcr(dist, Ag, W_s, W_Ig, diag_g, rhs_g, resscale_g, tol, maxiter):
    """Preconditioned CR of reference src/conjugate_residuals.cc:90-213 on row slabs."""
    import torch

    def allsum(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t[0])

    def allmax(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def applyC(y_g):
        t = torch.from_numpy(W_s * (Ag.T @ y_g))
        dist.all_reduce(t)                     # the one exchange step per apply
        lhs = W_Ig * y_g + Ag @ t.numpy()
        return lhs, allsum(float(y_g @ lhs))

    def applyP(r_g):
        l = r_g / diag_g
        return l, allsum(float(l @ r_g))

    lhs = np.zeros_like(rhs_g)
    residual = rhs_g.copy()
    sres, rps = applyP(residual)
    Csres, cdot = applyC(sres)
    step, Cstep = sres.copy(), Csres.copy()
    it, err = 0, 0
    while True:
        resnorm = allmax(float(np.abs(resscale_g * residual).max()))
        if resnorm <= tol:
            break
        if it == maxiter:
            err = 201
            break
        if cdot <= 0:
            err = 202
            break
        pC, pdot = applyP(Cstep)
        if pdot <= 0:
            err = 203
            break
        alpha = cdot / pdot
        lhs += alpha * step
        residual -= alpha * Cstep
        sres -= alpha * pC
        Csres, cdotnew = applyC(sres)
        beta = cdotnew / cdot
        step = sres + beta * step
        Cstep = Csres + beta * Cstep
        cdot = cdotnew
        it += 1
        if it % 5 == 0:
            sres, rsdot = applyP(residual)
            if rsdot >= rps:
                err = 204
                break
            rps = rsdot
    return lhs, it, err


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from ipx_amd import partition, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n = 501, 1100            # odd row count: ragged slabs
        A = synth.synthetic_lp(m, n, 8, 5)
        st = synth.synthetic_ipm_state(m, n, 1.0, 5)
        slab = partition.row_slab(A, st, rank, world)
        Ag = slab.A.to_scipy()
        mg = slab.A.nrow
        Wg = slab.xl / slab.zl      # local view [W_s ; W_I slice]
        W_s, W_Ig = Wg[:n], Wg[n:]
        # partitioned apply
        y = np.random.default_rng(0).standard_normal(m)
        y_g = y[slab.r0:slab.r1]
        t = torch.from_numpy(W_s * (Ag.T @ y_g))
        dist.all_reduce(t)
        lhs_g = W_Ig * y_g + Ag @ t.numpy()
        # partitioned KKTSolverDiag::Solve (kkt_solver_diag.cc:82-118)
        diag_g = W_Ig + (Ag.multiply(Ag)) @ W_s
        a_g, b_g = slab.a, slab.b
        rhs_g = -b_g + W_Ig * a_g[n:] + Ag @ (W_s * a_g[:n])
        tol = 0.3 * np.sqrt(st["mu"])
        y_sol, it, err = _partitioned_pcr(dist, Ag, W_s, W_Ig, diag_g, rhs_g, 1.0 / np.sqrt(W_Ig), tol, 500)
        aty = torch.from_numpy(Ag.T @ y_sol)
        dist.all_reduce(aty)
        x_s = W_s * (a_g[:n] - aty.numpy())
        x_Ig = b_g - Ag @ x_s
        gathered = [None] * world
        dist.all_gather_object(gathered, dict(r0=slab.r0, r1=slab.r1, lhs=lhs_g, y=y_sol,
                                              x=np.concatenate([x_s, x_Ig]), it=it, err=err))
        if rank == 0:
            np.save(out, np.array([gathered], dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


def test_row_partition_world2(oracle, tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ipx_amd import partition, synth
    from oracle import pyoracle as po
    out = str(tmp_path / "parts.npy")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    parts = np.load(out, allow_pickle=True)[0]
    m, n = 501, 1100
    A = synth.synthetic_lp(m, n, 8, 5)
    st = synth.synthetic_ipm_state(m, n, 1.0, 5)
    # ranges tile [0, m) exactly
    assert [p["r0"] for p in parts] == [0, 251] and [p["r1"] for p in parts] == [251, 501]
    assert partition.row_range(m, 0, 2) == (0, 251) and partition.row_range(7, 2, 3) == (5, 7)
    Ao = po.Csc(m, n, A.p, A.i, A.x)
    W = st["xl"] / st["zl"]
    y = np.random.default_rng(0).standard_normal(m)
    lhs_ref, _ = oracle.normal_apply(Ao, W, y)
    lhs = np.concatenate([p["lhs"] for p in parts])
    assert np.abs(lhs - lhs_ref).max() <= 1e-12 * np.abs(lhs_ref).max()
    k = oracle.kkt_diag(Ao, maxiter=500)
    k.factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"])
    x_ref, y_ref, it_ref, err_ref, _ = k.solve(st["a"], st["b"], 0.3 * np.sqrt(st["mu"]))
    assert all(p["err"] == err_ref == 0 for p in parts)
    assert all(abs(p["it"] - it_ref) <= 2 for p in parts) and parts[0]["it"] == parts[1]["it"]
    x, ysol = partition.assemble(n, [p["x"] for p in parts], [p["y"] for p in parts])
    assert np.abs(ysol - y_ref).max() <= 1e-6 * np.abs(y_ref).max()
    assert np.abs(x - x_ref).max() <= 1e-5 * np.abs(x_ref).max()
    # structural part is replicated identically
    assert np.array_equal(parts[0]["x"][:n], parts[1]["x"][:n])


def _col_worker(rank, world, port, out):
    """Column partition (ipx_amd/partition.py, col_slab): every m-vector replicated, one all-reduce of
    the m partial sums per apply, NO scalar exchange -- each rank runs the plain serial PCR."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from ipx_amd import partition, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n = 501, 1101            # odd column count: ragged slabs
        A = synth.synthetic_lp(m, n, 8, 5)
        st = synth.synthetic_ipm_state(m, n, 1.0, 5)
        slab = partition.col_slab(A, st, rank, world)
        Ag = slab.A.to_scipy()
        ng = slab.A.ncol
        Wg = slab.xl / slab.zl      # [W of my columns ; W_I (all rows)]
        W_s, W_I = Wg[:ng], Wg[ng:]

        def allsum_vec(v):
            t = torch.from_numpy(np.ascontiguousarray(v))
            dist.all_reduce(t)
            return t.numpy()

        def applyC(y):
            lhs = W_I * y + allsum_vec(Ag @ (W_s * (Ag.T @ y)))
            return lhs, float(y @ lhs)

        diag = W_I + allsum_vec((Ag.multiply(Ag)) @ W_s)
        applyP = lambda r: ((r / diag), float((r / diag) @ r))
        y0 = np.random.default_rng(0).standard_normal(m)
        lhs0, _ = applyC(y0)
        a_g, b = slab.a, slab.b
        rhs = -b + W_I * a_g[ng:] + allsum_vec(Ag @ (W_s * a_g[:ng]))
        tol = 0.3 * np.sqrt(st["mu"])
        resscale = 1.0 / np.sqrt(W_I)
        # serial PCR (conjugate_residuals.cc:90-213), identical on every rank
        lhs = np.zeros(m); residual = rhs.copy()
        sres, rps = applyP(residual); Csres, cdot = applyC(sres)
        step, Cstep = sres.copy(), Csres.copy()
        it, err = 0, 0
        while True:
            if np.abs(resscale * residual).max() <= tol: break
            if it == 500: err = 201; break
            if cdot <= 0: err = 202; break
            pC, pdot = applyP(Cstep)
            if pdot <= 0: err = 203; break
            alpha = cdot / pdot
            lhs += alpha * step; residual -= alpha * Cstep; sres -= alpha * pC
            Csres, cdotnew = applyC(sres)
            beta = cdotnew / cdot
            step = sres + beta * step; Cstep = Csres + beta * Cstep; cdot = cdotnew
            it += 1
            if it % 5 == 0:
                sres, rsdot = applyP(residual)
                if rsdot >= rps: err = 204; break
                rps = rsdot
        x_s = W_s * (a_g[:ng] - Ag.T @ lhs)
        x_I = b - allsum_vec(Ag @ x_s)
        gathered = [None] * world
        dist.all_gather_object(gathered, dict(c0=slab.c0, c1=slab.c1, lhs=lhs0, y=lhs,
                                              x=np.concatenate([x_s, x_I]), it=it, err=err))
        if rank == 0:
            np.save(out, np.array([gathered], dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


def test_column_partition_world2(oracle, tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ipx_amd import partition, synth
    from oracle import pyoracle as po
    out = str(tmp_path / "cparts.npy")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_col_worker, args=(2, port, out), nprocs=2, join=True)
    parts = np.load(out, allow_pickle=True)[0]
    m, n = 501, 1101
    A = synth.synthetic_lp(m, n, 8, 5)
    st = synth.synthetic_ipm_state(m, n, 1.0, 5)
    assert [p["c0"] for p in parts] == [0, 551] and [p["c1"] for p in parts] == [551, 1101]
    Ao = po.Csc(m, n, A.p, A.i, A.x)
    W = st["xl"] / st["zl"]
    lhs_ref, _ = oracle.normal_apply(Ao, W, np.random.default_rng(0).standard_normal(m))
    for p in parts:   # replicated result
        assert np.abs(p["lhs"] - lhs_ref).max() <= 1e-12 * np.abs(lhs_ref).max()
    assert np.array_equal(parts[0]["lhs"], parts[1]["lhs"]) and np.array_equal(parts[0]["y"], parts[1]["y"])
    k = oracle.kkt_diag(Ao, maxiter=500)
    k.factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"])
    x_ref, y_ref, it_ref, err_ref, _ = k.solve(st["a"], st["b"], 0.3 * np.sqrt(st["mu"]))
    assert all(p["err"] == err_ref == 0 for p in parts)
    assert all(abs(p["it"] - it_ref) <= 2 for p in parts) and parts[0]["it"] == parts[1]["it"]
    x = partition.assemble_cols(m, [p["x"] for p in parts])
    assert np.abs(parts[0]["y"] - y_ref).max() <= 1e-6 * np.abs(y_ref).max()
    assert np.abs(x - x_ref).max() <= 1e-5 * np.abs(x_ref).max()


def test_col_slab_roundtrip():
    sys.path.insert(0, ROOT)
    from ipx_amd import partition, synth
    A = synth.synthetic_lp(97, 41, 5, 2)
    S = A.to_scipy().toarray()
    v = np.arange(41 + 97, dtype=float)
    for world in (1, 2, 3, 8):
        cols, parts = [], []
        for r in range(world):
            c0, c1 = partition.row_range(41, r, world)
            B = partition.col_slab_matrix(A, c0, c1)
            assert B.nrow == 97 and B.ncol == c1 - c0
            assert np.array_equal(B.to_scipy().toarray(), S[:, c0:c1])
            cols.append(c1 - c0)
            parts.append(partition.col_local_vector(v, 41, c0, c1))
        assert sum(cols) == 41 and max(cols) - min(cols) <= 1
        assert np.array_equal(partition.assemble_cols(97, parts), v)


def test_slab_matrix_roundtrip():
    sys.path.insert(0, ROOT)
    from ipx_amd import partition, synth
    A = synth.synthetic_lp(97, 40, 5, 2)
    S = A.to_scipy().toarray()
    for world in (1, 2, 3, 8):
        rows = []
        for r in range(world):
            r0, r1 = partition.row_range(97, r, world)
            B = partition.slab_matrix(A, r0, r1)
            assert B.nrow == r1 - r0 and B.ncol == 40
            assert np.array_equal(B.to_scipy().toarray(), S[r0:r1])
            rows.append(r1 - r0)
        assert sum(rows) == 97 and max(rows) - min(rows) <= 1
