"""Prepare at C3 (planted factors from the host; then the device LU's factors): wall time per call, for profiling
the set-up path alone.  usage: python scripts/gpu_prepare_only.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n = 1000000, 2000000
A0 = synth.synthetic_lp(m, n, 8, 12345)
B = synth.planted_lu_basis(A0, offdiag=3, seed=12345)
colscale = synth.synthetic_basis_state(B["status"], 1.0, 12345)
ctx = kkt.KktContext(B["A"])
for rep in range(3):
    t0 = time.perf_counter()
    ctx.split_prepare(B["L"], B["U"], B["rowperm"], B["colperm"], B["basis"], B["status"], colscale)
    print("split_prepare (host factors) %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
ctx.close()
P = synth.lp_like_basis(m, n, seed=12345, bump=1000, offdiag=3)
cs = synth.synthetic_basis_state(P["status"], 1.0, 12345)
ctx = kkt.KktContext(P["A"])
for rep in range(3):
    t0 = time.perf_counter()
    F = ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
    t1 = time.perf_counter()
    ctx.split_prepare_lu(P["status"], cs)
    print("lu_factorize_basis %.1f ms, split_prepare_lu %.1f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3), flush=True)
ctx.close()
