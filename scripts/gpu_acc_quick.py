"""Quick look at the accumulated-tiles layout at C3: per-pass timing from ipxk_create's own measurement, the apply
time, and the product against scipy.  usage: python scripts/gpu_acc_quick.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipx_amd import kkt, synth
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1 << 20, 2 << 20)
A = synth.synthetic_lp(m, n, 8, 12345)
t0 = time.perf_counter()
c = kkt.KktContext(A, device=0)
print("create %.1f ms (wall)" % ((time.perf_counter() - t0) * 1e3), c.layout_info(0)[1])
print(c.spmv_layout())
rng = np.random.default_rng(1)
W = rng.uniform(0.1, 10, n + m)
y = rng.standard_normal(m)
c.normal_prepare(W)
l, d = c.normal_apply(y)
S = A.to_scipy()
ref = W[n:] * y + S @ (W[:n] * (S.T @ y))
print("apply relerr %.2e" % (np.abs(l - ref).max() / np.abs(ref).max()))
c.set_pointer_mode(True)
rhs = c.vector(m, y); lhs = c.vector(m)
c.time_normal_apply(rhs, lhs, 5)
print("apply %.1f us" % (c.time_normal_apply(rhs, lhs, 50) / 50 * 1e3))
c.time_normal_apply(rhs, lhs, 3000)          # half a second of sustained load
print("apply after 3000 more: %.1f us" % (c.time_normal_apply(rhs, lhs, 50) / 50 * 1e3))
c.close()
