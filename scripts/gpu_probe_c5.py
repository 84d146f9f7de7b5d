"""C5: dense-column stress LP (m=200k, n=400k, 32 dense columns) through KKTSolverDiag with SMW."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import kkt_residual_diag
m, n = 200000, 400000
A = synth.synthetic_lp(m, n, 8, 12345, num_dense=32)
st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
ctx = kkt.KktContext(A)
print("nnz", A.nnz, "dense cols", ctx.num_dense_cols)
t0 = time.time(); err = ctx.kkt_diag_factorize(st['xl'], st['xu'], st['zl'], st['zu'], st['mu']); t1 = time.time()
print("factorize err", err, "%.3f s" % (t1 - t0))
tol = 0.3 * np.sqrt(st['mu'])
for _ in range(2):
    t0 = time.time(); x, y, it, e, tm = ctx.kkt_diag_solve(st['a'], st['b'], tol, 500); t1 = time.time()
    print("solve: iters %d err %d total %.1f ms cr %.1f ms -> %.1f us/iter" % (it, e, (t1-t0)*1e3, tm.cr*1e3, tm.cr/max(it,1)*1e6))
W, _ = ctx.kkt_diag_get()
r1, r2 = kkt_residual_diag(A, W, st['a'], st['b'], x, y)
print("kkt residuals: |AIx-b| %.2e  scaled slack res %.3e (tol %.3e)  struct res %.2e" % (np.abs(r2).max(), np.abs(np.sqrt(W[n:]) * r1[n:]).max(), tol, np.abs(r1[:n]).max()))
# without SMW for comparison
err = ctx.kkt_diag_factorize(st['xl'], st['xu'], st['zl'], st['zu'], st['mu'], precond_dense_cols=False)
x, y, it, e, tm = ctx.kkt_diag_solve(st['a'], st['b'], tol, 500)
print("without dense-column treatment: iters %d err %d" % (it, e))
