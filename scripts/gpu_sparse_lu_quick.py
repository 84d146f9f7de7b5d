"""The elimination rounds of the device LU against the CPU restatement on bases with misplaced columns.
usage: python scripts/gpu_sparse_lu_quick.py [m K bump_max sparse_min]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
from oracle import pyoracle as po

m, K, bmax, smin = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (20000, 50, 200, 64)
bump = int(sys.argv[5]) if len(sys.argv) > 5 else 100
G = synth.misplaced_basis_matrix(m, K, seed=12345, bump=bump)
os.environ["IPXK_LU_BUMP_MAX"] = str(bmax)
os.environ.setdefault("IPXK_LU_SPARSE", "1")
os.environ["IPXK_LU_SPARSE_MIN"] = str(smin)
ctx = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
for rep in range(2):
    t0 = time.perf_counter()
    F = ctx.lu_factorize(m, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, download=(rep == 1 and "--cpu" in sys.argv))
    dt = time.perf_counter() - t0
    info = F
    print("device %.1f ms (singletons %.1f, bump %.1f, assembly %.1f): %s" % (dt * 1e3, info["seconds_singletons"] * 1e3, info["seconds_bump"] * 1e3,
          info["seconds_assemble"] * 1e3, {k: info[k] for k in ("col_singletons", "row_singletons", "bump", "rounds", "sparse_pivots", "sparse_rounds", "num_dependent", "lnz", "unz")}), flush=True)
    print("fill %.2f" % ((info["lnz"] + info["unz"]) / len(G["Bi"])), flush=True)
if "--cpu" in sys.argv:
    t0 = time.perf_counter()
    Fo = po.Oracle().lu_factorize(m, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, bump_limit=bmax, sparse_min=smin, slow_den=int(os.environ.get("IPXK_LU_SPARSE_SLOW_DEN", "256")))
    print("CPU restatement %.2f s" % (time.perf_counter() - t0), Fo["info"], flush=True)
    for key in ("rowperm", "colperm", "dependent"):
        print(key, "equal:", np.array_equal(F[key], Fo[key]))
    for key in ("L", "U"):
        same = np.array_equal(F[key].p, Fo[key].p) and np.array_equal(F[key].i, Fo[key].i)
        print(key, "pattern equal:", same, "values equal:", same and np.array_equal(F[key].x, Fo[key].x),
              "max diff", float(np.abs(F[key].x - Fo[key].x).max()) if same and len(F[key].x) else None)
ctx.close()
