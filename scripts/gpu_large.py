"""Scale check beyond the benchmark size: m=4M, n=8M, nnz=64M (4x C3) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IPXK_VERBOSE", "1")
import numpy as np
from ipx_amd import synth, kkt
m, n = int(os.environ.get("M", 4000000)), int(os.environ.get("N", 8000000))
t0 = time.time(); A = synth.synthetic_lp(m, n, 8, 12345); st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
print("generate %.1f s, nnz %d" % (time.time() - t0, A.nnz), flush=True)
t0 = time.time(); ctx = kkt.KktContext(A); print("ipxk_create %.1f s" % (time.time() - t0), ctx.spmv_layout(), flush=True)
assert ctx.kkt_diag_factorize(st['xl'], st['xu'], st['zl'], st['zu'], st['mu']) == 0
ctx.set_pointer_mode(True)
a = ctx.vector(n + m, st['a']); b = ctx.vector(m, st['b']); x = ctx.vector(n + m); y = ctx.vector(m)
tol = 0.3 * np.sqrt(st['mu'])
for k in range(3):
    t0 = time.perf_counter(); it, e, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, 500); t1 = time.perf_counter()
    print("solve: %d its err %d total %.1f ms -> %.1f us/iteration" % (it, e, (t1 - t0) * 1e3, tm.cr / max(it, 1) * 1e6), flush=True)
rhs = ctx.vector(m, np.random.default_rng(0).standard_normal(m)); lhs = ctx.vector(m)
ctx.time_normal_apply(rhs, lhs, 3)
ms = ctx.time_normal_apply(rhs, lhs, 20) / 20
B = ctx.normal_apply_bytes
print("apply %.1f us, %.1f MB algorithmic, %.2f TB/s" % (ms * 1e3, B / 1e6, B / (ms * 1e-3) / 1e12))
xs, ys = x.download(), y.download()
S = A.to_scipy(); W = st['xl'] / st['zl']
r2 = S @ xs[:n] + xs[n:] - st['b']
r1 = xs / W + np.concatenate([S.T @ ys, ys]) - st['a']
print("KKT residuals: |AIx-b| %.1e, scaled slack %.3e (tol %.3e), struct %.1e" % (np.abs(r2).max(), np.abs(np.sqrt(W[n:]) * r1[n:]).max(), tol, np.abs(r1[:n]).max()))
