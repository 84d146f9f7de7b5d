"""Experiment: probe residuals of the inverted tail blocks on planted factors with an ill-conditioned tail.
usage: python scripts/gpu_guard_probe.py m n big_rows big"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IPXK_TAIL_MIN_DIM", "1000")
os.environ["IPXK_SWEEP_STATS"] = "1"
from ipx_amd import kkt, synth
m, n, br, big = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
B = synth.planted_lu_basis(synth.synthetic_lp(m, n, 8, 3), seed=3, big_rows=br, big=big)
cs = synth.synthetic_basis_state(B["status"], 1.0, 3)
import scipy.sparse as sp, scipy.sparse.linalg as spl
for tol in ("1e-10", "1e300"):
    os.environ["IPXK_INVERSE_TOL"] = tol
    import importlib
    ctx = kkt.KktContext(B["A"], device=0)
    ctx.split_prepare(B["L"], B["U"], B["rowperm"], B["colperm"], B["basis"], B["status"], cs)
    print("tol", tol, "stats", ctx.split_inverse_stats())
    rng = np.random.default_rng(1)
    r = rng.standard_normal(m)
    Bm = B["A"].to_scipy()[:, :m].tocsr()
    xt = rng.standard_normal(m)
    for trans in ("N", "T"):
        M = Bm.T if trans == "T" else Bm
        rr = M @ xt                       # a right-hand side whose solution is moderate
        x = ctx.solve_dense(rr, trans)
        print("  trans %s: residual / |r| %.2e, |x - x_true| / |x_true| %.2e" % (trans, np.abs(M @ x - rr).max() / np.abs(rr).max(),
                                                                              np.abs(x - xt).max() / np.abs(xt).max()))
    ctx.close()
