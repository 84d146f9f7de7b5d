"""Level structure and launch plan of the four sweeps of the C3 planted basis (IPXK_SWEEP_STATS), with the parts of one
operator application timed by HIP events."""
import os, sys
os.environ["IPXK_SWEEP_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
A0 = synth.synthetic_lp(m, n, 8, 12345)
B = synth.planted_lu_basis(A0, offdiag=3, seed=12345)
st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
colscale = synth.synthetic_basis_state(B["status"], 1.0, 12345)
ctx = kkt.KktContext(B["A"])
ctx.split_prepare(B["L"], B["U"], B["rowperm"], B["colperm"], B["basis"], B["status"], colscale)
ctx.set_pointer_mode(True)
a, b = ctx.vector(n + m, st["a"]), ctx.vector(m, st["b"])
x, y = ctx.vector(n + m), ctx.vector(m)
tol = 0.3 * np.sqrt(st["mu"])
it, err, tm = ctx.kkt_basis_solve_resident(a, b, x, y, tol, 500)
cr = 0.0
for _ in range(5):
    it, err, tm = ctx.kkt_basis_solve_resident(a, b, x, y, tol, 500)
    cr += tm.cr
print("iterations %d err %d, %.1f us per CR iteration" % (it, err, cr / 5 / max(it, 1) * 1e6), flush=True)
ctx.set_profiling(True)
itp, errp, tmp = ctx.kkt_basis_solve_resident(a, b, x, y, tol, 500)
ctx.set_profiling(False)
na = itp + 1
print("per application: backward pair %.1f us, N N' %.1f us, forward pair %.1f us" % (tmp.solve_Bt / na * 1e6, tmp.op / na * 1e6, tmp.solve_B / na * 1e6))
print("levels", ctx.split_levels(), "nnz L", B["L"].nnz, "nnz U", B["U"].nnz)
ctx.close()
