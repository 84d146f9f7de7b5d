"""Per-kernel SpMV timing experiment (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt

def run(m, n, k=8, reps=20):
    A = synth.synthetic_lp(m, n, k, 12345)
    ctx = kkt.KktContext(A)
    rng = np.random.default_rng(0)
    W = 10.0 ** rng.uniform(-2, 2, n + m)
    ctx.normal_prepare(W)
    rhs = ctx.vector(m, rng.standard_normal(m)); lhs = ctx.vector(m)
    ctx.time_normal_apply(rhs, lhs, 3)
    ms = ctx.time_normal_apply(rhs, lhs, reps)
    B = ctx.normal_apply_bytes
    print("m=%d n=%d nnz=%d: %.1f us/apply %.2f TB/s" % (m, n, A.nnz, ms/reps*1e3, B/(ms/reps*1e-3)/1e12), flush=True)
    ctx.close()

for (m, n) in [(1000000, 2000000), (250000, 2000000), (125000, 250000), (2000000, 250000)]:
    run(m, n)
