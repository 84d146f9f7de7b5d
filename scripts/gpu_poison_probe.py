"""IPXK_POISON=1: where does the diag path read a double it never wrote?"""
import os, sys
os.environ["IPXK_POISON"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n = 3000, 7000
A = synth.synthetic_lp(m, n, 8, 1)
st = synth.synthetic_ipm_state(m, n, 1.0, 1)
c = kkt.KktContext(A)
rng = np.random.default_rng(0)
W = rng.uniform(0.1, 10, n + m); y = rng.standard_normal(m)
c.normal_prepare(W)
l, d = c.normal_apply(y)
print("normal_apply nan:", np.isnan(l).sum(), "dot", d)
err = c.kkt_diag_factorize(st["xl"], st["xu"], st["zl"], st["zu"], st["mu"])
Wd, rs = c.kkt_diag_get()
print("factorize err", err, "W nan", np.isnan(Wd).sum(), "resscale nan", np.isnan(rs).sum())
x, yy, it, e, _ = c.kkt_diag_solve(st["a"], st["b"], 0.3 * np.sqrt(st["mu"]), 500)
print("solve it", it, "err", e, "x nan", np.isnan(x).sum(), "y nan", np.isnan(yy).sum())
