"""GPU: partitioned solves with REAL separate rank processes.  The test boxes have one GPU and RCCL
refuses two ranks on one device, so the ranks share GPU 0 and the library's collectives run over its
direct exchange (IPXK_COMM=direct, ipx_amd/csrc/comm.hip: buffers mapped between the processes with hipIpc,
reduce-scatter + all-gather kernels, flag table in shared host memory) -- the same device code path that
carries the collectives between the GPUs of a node.  Everything else -- slab contexts, the per-rank CR loops
and their lock-step control flow, scalar exchange, partial products -- is what RCCL runs use too.  Results
are compared with the oracle's unpartitioned solve."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import diag_problem, relerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("part,world", [("rows", 2), ("rows", 3), ("columns", 2), ("columns", 3)])
def test_partitioned_solve_multiprocess(oracle, tmp_path, part, world):
    from ipx_amd import kkt, partition
    from oracle import pyoracle as po
    kkt.load_library()
    m, n, seed = 2501, 6007, 61           # ragged slabs for 2 and 3 ranks
    env = dict(os.environ, IPXK_COMM="direct")
    idfile, out = str(tmp_path / "uid"), str(tmp_path / "res")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), str(r), str(world),
                               idfile, out, part, str(m), str(n), str(seed)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank did not finish:\n" + "\n".join(logs))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [np.load(out + ".rank%d.npz" % r) for r in range(world)]
    A, st = diag_problem(m, n, seed=seed)
    Ao = po.Csc(m, n, A.p, A.i, A.x)
    k = oracle.kkt_diag(Ao, maxiter=500)
    k.factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"])
    tol = 0.3 * np.sqrt(st["mu"])
    x_ref, y_ref, it_ref, err_ref, _ = k.solve(st["a"], st["b"], tol)
    lhs_ref, dot_ref = oracle.normal_apply(Ao, k.get()[0], np.random.default_rng(0).standard_normal(m))
    its = [int(r["it"]) for r in res]
    assert all(int(r["err"]) == err_ref == 0 for r in res) and len(set(its)) == 1 and abs(its[0] - it_ref) <= 2
    assert all(abs(float(r["dot"]) - dot_ref) <= 1e-12 * abs(dot_ref) for r in res)
    if part == "rows":
        lhs = np.concatenate([r["lhs"] for r in res])
        x, y = partition.assemble(n, [r["x"] for r in res], [r["y"] for r in res])
        assert all(np.array_equal(res[0]["x"][:n], r["x"][:n]) for r in res)      # replicated structural part
    else:
        lhs, y = res[0]["lhs"], res[0]["y"]
        assert all(np.array_equal(res[0]["y"], r["y"]) and np.array_equal(res[0]["lhs"], r["lhs"]) for r in res)
        x = partition.assemble_cols(m, [r["x"] for r in res])
    assert relerr(lhs, lhs_ref) <= 1e-12
    assert relerr(y, y_ref) < 1e-6 and relerr(x, x_ref) < 1e-5
