"""DiagonalPrecond with many dense columns: factorize (Schur assembly + Cholesky) and apply timings.
usage: python scripts/gpu_chol_bench.py [m n num_dense]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n, nd = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (20000, 42000, 1000)
A = synth.synthetic_lp(m, n, 8, 7, num_dense=nd)
st = synth.synthetic_ipm_state(m, n, 1.0, 7)
ctx = kkt.KktContext(A)
print("dense columns classified:", ctx.num_dense_cols, flush=True)
for rep in range(2):
    t0 = time.perf_counter()
    err = ctx.kkt_diag_factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"])
    t1 = time.perf_counter()
    print("factorize: errflag %d, %.1f ms" % (err, (t1 - t0) * 1e3), flush=True)
rhs = np.random.default_rng(0).standard_normal(m)
ctx.diag_apply(rhs)
t0 = time.perf_counter()
for _ in range(20): ctx.diag_apply(rhs)
print("preconditioner apply (host vectors, incl. transfers): %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
x, y, it, e, tm = ctx.kkt_diag_solve(st["a"], st["b"], 0.3 * st["mu"] ** 0.5, 500)
print("solve: %d CR iterations, errflag %d, CR loop %.1f ms = %.3f ms per iteration" % (it, e, tm.cr * 1e3, tm.cr * 1e3 / max(it, 1)), flush=True)
