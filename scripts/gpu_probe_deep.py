"""Stress point of SURVEY 8d: planted factors confined to a band -> thousands of narrow levels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ipx_amd import synth, kkt
from helpers import basis_problem
m, n, band = int(os.environ.get("M", 1000000)), int(os.environ.get("N", 2000000)), int(os.environ.get("BAND", 1000))
B, st, colscale = basis_problem(m, n, seed=12345, band=band)
ctx = kkt.KktContext(B["A"])
t0 = time.time()
ctx.split_prepare(B["L"], B["U"], B["rowperm"], B["colperm"], B["basis"], B["status"], colscale)
print("band %d: prepare %.2f s, levels %s" % (band, time.time() - t0, ctx.split_levels()), flush=True)
rhs = np.random.default_rng(0).standard_normal(m)
ctx.forward_solve(rhs)
t0 = time.perf_counter()
for _ in range(3): ctx.forward_solve(rhs)
print("forward solve (L then U, host vectors): %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
tol = 0.3 * np.sqrt(st["mu"])
t0 = time.perf_counter(); x, y, it, e, tm = ctx.kkt_basis_solve(st["a"], st["b"], tol, 500); t1 = time.perf_counter()
print("kkt_basis_solve: %d its err %d, %.1f ms, %.0f us per CR iteration" % (it, e, (t1 - t0) * 1e3, tm.cr / max(it, 1) * 1e6))
