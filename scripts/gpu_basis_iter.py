"""Basis path at BASELINE config 3 (m = 1M, n = 2M, planted LU factors): per-CR-iteration time and its split
into the two sweep pairs and N N' (HIP events around each part, ipxk_set_profiling).  Launch-plan knobs come
from the environment (IPXK_SWEEP_NARROW, IPXK_SWEEP_GRID, IPXK_SWEEP_XCD_WGS, IPXK_TRISOLVE=levels).
usage: python scripts/gpu_basis_iter.py [rows cols]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000000, 2000000)
A0 = synth.synthetic_lp(m, n, 8, 12345)
B = synth.planted_lu_basis(A0, offdiag=3, seed=12345)
st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
colscale = synth.synthetic_basis_state(B["status"], 1.0, 12345)
ctx = kkt.KktContext(B["A"])
for rep in range(2):
    t0 = time.perf_counter()
    ctx.split_prepare(B["L"], B["U"], B["rowperm"], B["colperm"], B["basis"], B["status"], colscale)
    t_prep = time.perf_counter() - t0
t0 = time.perf_counter()
ctx.split_rescale(B["status"], colscale)
t_resc = time.perf_counter() - t0
print("prepare %.1f ms (second call), rescale %.1f ms, levels %s, layouts %s" % (t_prep * 1e3, t_resc * 1e3, ctx.split_levels(), ctx.spmv_layout()), flush=True)
ctx.set_pointer_mode(True)
a, b = ctx.vector(n + m, st["a"]), ctx.vector(m, st["b"])
x, y = ctx.vector(n + m), ctx.vector(m)
tol = 0.3 * np.sqrt(st["mu"])
it, err, tm = ctx.kkt_basis_solve_resident(a, b, x, y, tol, 500)
t0 = time.perf_counter()
K = 3
for _ in range(K):
    it, err, tm = ctx.kkt_basis_solve_resident(a, b, x, y, tol, 500)
ctx.synchronize()
dt = (time.perf_counter() - t0) / K
print("solve %.2f ms, %d CR iterations (errflag %d), CR loop %.2f ms = %.1f us per iteration" % (dt * 1e3, it, err, tm.cr * 1e3, tm.cr / max(it, 1) * 1e6), flush=True)
ctx.set_profiling(True)
it, err, tm = ctx.kkt_basis_solve_resident(a, b, x, y, tol, 500)
napply = it + 1
print("profiled: per operator application: backward pair (U', L') %.1f us, N N' %.1f us, forward pair (L, U) %.1f us; CR loop %.1f us per iteration"
      % (tm.solve_Bt / napply * 1e6, tm.op / napply * 1e6, tm.solve_B / napply * 1e6, tm.cr / max(it, 1) * 1e6), flush=True)
ctx.close()
