"""The inverse of the dense block of the factors (Prepare): recursive doubling on the matrix cores against the older
one-blocked-solve-per-column kernel.  usage: python scripts/gpu_dense_inverse_bench.py [bump ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
bumps = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8000]
for bump in bumps:
    m, n = 100000, 220000
    P = synth.lp_like_basis(m, n, seed=12345, bump=bump, offdiag=3)
    colscale = synth.synthetic_basis_state(P["status"], 1.0, 12345)
    rhs = np.random.default_rng(1).standard_normal(m)
    res = {}
    for di in ("1", "0"):
        if di == "0" and bump > 4096:
            continue
        os.environ["IPXK_DENSE_INVERSE_MIN"] = di
        ctx = kkt.KktContext(P["A"])
        ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
        ctx.split_prepare_lu(P["status"], colscale)
        t = []
        for rep in range(3):
            t0 = time.perf_counter()
            ctx.split_prepare_lu(P["status"], colscale)
            t.append(time.perf_counter() - t0)
        os.environ["IPXK_BUMP_INVERSE_MIN"] = "0"
        res[di] = (min(t), ctx.solve_dense(rhs, "N"), ctx.split_inverse_stats())
        ctx.close()
    line = "bump %5d: Prepare with the inverse on the matrix cores %.1f ms (probe %.1e)" % (bump, res["1"][0] * 1e3, res["1"][2][2])
    if "0" in res:
        line += "; one blocked solve per column %.1f ms (probe %.1e); solves agree to %.1e" % (
            res["0"][0] * 1e3, res["0"][2][2], np.abs(res["1"][1] - res["0"][1]).max() / np.abs(res["0"][1]).max())
    print(line, flush=True)
