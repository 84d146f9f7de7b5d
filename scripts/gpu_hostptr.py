"""Host-pointer (drop-in boundary) vs resident solve at C3: PCIe-inclusive cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n = int(os.environ.get("M", 1000000)), int(os.environ.get("N", 2000000))
A = synth.synthetic_lp(m, n, 8, 12345); st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
ctx = kkt.KktContext(A)
assert ctx.kkt_diag_factorize(st['xl'], st['xu'], st['zl'], st['zu'], st['mu']) == 0
tol = 0.3 * np.sqrt(st['mu'])
for k in range(4):
    t0 = time.perf_counter(); x, y, it, e, tm = ctx.kkt_diag_solve(st['a'], st['b'], tol, 500); t1 = time.perf_counter()
    print("host pointers: total %.2f ms, cr %.2f ms (%d its)" % ((t1 - t0) * 1e3, tm.cr * 1e3, it), flush=True)
ctx.set_pointer_mode(True)
a = ctx.vector(n + m, st['a']); b = ctx.vector(m, st['b']); xd = ctx.vector(n + m); yd = ctx.vector(m)
for k in range(3):
    t0 = time.perf_counter(); it, e, tm = ctx.kkt_diag_solve_resident(a, b, xd, yd, tol, 500); t1 = time.perf_counter()
    print("resident:      total %.2f ms, cr %.2f ms" % ((t1 - t0) * 1e3, tm.cr * 1e3), flush=True)
