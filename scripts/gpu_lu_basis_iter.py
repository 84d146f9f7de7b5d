"""Basis-preconditioned solve on factors that came out of the device LU (LP-like basis with a dense bump): CR
iteration time and its split, as scripts/gpu_basis_iter.py does for the planted factors.
usage: python scripts/gpu_lu_basis_iter.py [m n bump]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n, bump = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1000000, 2000000, 1000)
P = synth.lp_like_basis(m, n, seed=12345, bump=bump, offdiag=3)
st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
colscale = synth.synthetic_basis_state(P["status"], 1.0, 12345)
ctx = kkt.KktContext(P["A"])
F = ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
ctx.split_prepare_lu(P["status"], colscale)
print("levels", ctx.split_levels(), "bump", F["bump"], flush=True)
ctx.set_pointer_mode(True)
a, b = ctx.vector(n + m, st["a"]), ctx.vector(m, st["b"])
x, y = ctx.vector(n + m), ctx.vector(m)
tol = 0.3 * np.sqrt(st["mu"])
it, err, tm = ctx.kkt_basis_solve_resident(a, b, x, y, tol, 500)
ctx.set_profiling(True)
it, err, tm = ctx.kkt_basis_solve_resident(a, b, x, y, tol, 500)
napply = it + 1
print("%d CR iterations (errflag %d); per operator application: backward pair (U', L') %.1f us, N N' %.1f us, forward pair (L, U) %.1f us; CR loop %.1f us per iteration"
      % (it, err, tm.solve_Bt / napply * 1e6, tm.op / napply * 1e6, tm.solve_B / napply * 1e6, tm.cr / max(it, 1) * 1e6), flush=True)
ctx.close()
