"""Maxvolume on the device at the C3 model size: an LP-like basis with `K` misplaced variables.
usage: python scripts/gpu_maxvol_bench.py [m n bump K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt

m, n, bump, K = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (1000000, 2000000, 1000, 200)
vt = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
shallow = len(sys.argv) > 6 and sys.argv[6] == "shallow"
if os.environ.get("STATE", "crash") == "crash":          # slack basis, K structural columns want to enter
    A = synth.synthetic_lp(m, n, 8, 12345)
    basis, status, colscale = synth.slack_basis_crash_state(m, n, K, 1.0, 12345)
    P = dict(A=A, basis=basis, status=status)
else:
    P = synth.lp_like_basis(m, n, seed=12345, bump=bump, offdiag=3, shallow=shallow, frac_slack=0.7 if shallow else 0.5)
    status = P["status"]
    colscale = synth.synthetic_slack_entering_state(P, K, 1.0, 12345) if os.environ.get("STATE") == "slack" else synth.synthetic_misplaced_state(status, K, 1.0, 12345)
ctx = kkt.KktContext(P["A"])
t0 = time.perf_counter()
ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
ctx.split_prepare_lu(status, colscale)
print("factorize + prepare %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
t0 = time.perf_counter()
r = ctx.maxvolume(status, colscale, volume_tol=vt, max_etas=int(os.environ.get('MAX_ETAS', '100')))
dt = time.perf_counter() - t0
print("maxvolume: %.1f ms, %d updates, %d skipped, %d slices, %d refused, %d factorizations, volinc %.2f" %
      (dt * 1e3, r["updates"], r["skipped"], r["slices"], r["refused"], r["factorizations"], r["volinc"]), flush=True)
steps = r["updates"] + r["skipped"]
print("per step (update or skipped column): %.2f ms" % (dt * 1e3 / max(steps, 1)), flush=True)
if "--cpu" in sys.argv:
    from oracle import pyoracle as po
    A = P["A"]
    t0 = time.perf_counter()
    B = po.Oracle().basis(po.Csc(m, n, A.p, A.i, A.x), P["basis"], status)
    w = B.maxvolume(colscale, volume_tol=vt)
    print("CPU restatement: %.2f s, %d updates; same exchanges: %s" % (time.perf_counter() - t0, w["updates"], np.array_equal(w["exchanges"], r["exchanges"])), flush=True)
ctx.close()
