"""C2 (m=50k, n=100k): where does a resident KKTSolverDiag::Solve spend its time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n = int(os.environ.get("M", 50000)), int(os.environ.get("N", 100000))
A = synth.synthetic_lp(m, n, 8, 12345)
st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
ctx = kkt.KktContext(A)
assert ctx.kkt_diag_factorize(st['xl'], st['xu'], st['zl'], st['zu'], st['mu']) == 0
ctx.set_pointer_mode(True)
a = ctx.vector(n + m, st['a']); b = ctx.vector(m, st['b']); x = ctx.vector(n + m); y = ctx.vector(m)
tol = 0.3 * np.sqrt(st['mu'])
for k in range(6):
    t0 = time.perf_counter()
    it, err, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, 500)
    t1 = time.perf_counter()
    print("solve %d: iters %d err %d total %.3f ms cr %.3f ms" % (k, it, err, (t1 - t0) * 1e3, tm.cr * 1e3), flush=True)
