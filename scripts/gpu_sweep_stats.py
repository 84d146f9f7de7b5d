"""Print the launch plan statistics of the four triangular sweeps for the C3 planted factors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["IPXK_SWEEP_STATS"] = os.environ.get("IPXK_SWEEP_STATS", "1")
import numpy as np
from ipx_amd import synth, kkt
m, n = int(os.environ.get("M", 1000000)), int(os.environ.get("N", 2000000))
A0 = synth.synthetic_lp(m, n, 8, 3)
B = synth.planted_lu_basis(A0, offdiag=3, seed=3)
st = synth.synthetic_ipm_state(m, n, 1.0, 3)
colscale = np.sqrt(st['xl'] / st['zl'])
colscale[B['status'] == 1] = np.inf
ctx = kkt.KktContext(B['A'])
for k in range(3):
    t0 = time.time()
    ctx.split_prepare(B['L'], B['U'], B['rowperm'], B['colperm'], B['basis'], B['status'], colscale)
    print("split_prepare call %d: %.1f ms" % (k, (time.time() - t0) * 1e3), flush=True)
print("levels", ctx.split_levels())
