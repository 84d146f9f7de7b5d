"""Device LU of an LP-like basis of the C3 model size (m = 1M): B = AI[:, basis] from the resident matrix,
phase timings, the hand-off to Prepare without a host round trip, the CPU restatement beside it.
usage: python scripts/gpu_lu_bench.py [m n bump]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt

m, n, bump = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1000000, 2000000, 1000)
t0 = time.perf_counter()
P = synth.lp_like_basis(m, n, seed=12345, bump=bump, offdiag=3)
colscale = synth.synthetic_basis_state(P["status"], 1.0, 12345)
G = P["G"]
print("generated %d x %d, nnz(B) %d in %.1f s" % (m, n, G["Bp"][-1], time.perf_counter() - t0), flush=True)
ctx = kkt.KktContext(P["A"])
for rep in range(3):
    t0 = time.perf_counter()
    F = ctx.lu_factorize_basis(P["basis"], 0.1, download=False)
    t1 = time.perf_counter()
    ctx.split_prepare_lu(P["status"], colscale)
    t2 = time.perf_counter()
    print("rep %d: factorize %.1f ms (singletons %.1f ms in %d rounds: %d col + %d row; bump %d: %.1f ms; assembly %.1f ms), nnz(L) %d nnz(U) %d; "
          "prepare from resident factors %.1f ms, levels %s" % (rep, (t1 - t0) * 1e3, F["seconds_singletons"] * 1e3, F["rounds"], F["col_singletons"],
          F["row_singletons"], F["bump"], F["seconds_bump"] * 1e3, F["seconds_assemble"] * 1e3, F["lnz"], F["unz"], (t2 - t1) * 1e3, ctx.split_levels()), flush=True)
t0 = time.perf_counter()
Fd = ctx.lu_factorize_basis(P["basis"], 0.1, download=True)
t1 = time.perf_counter()
ctx.split_prepare(Fd["L"], Fd["U"], Fd["rowperm"], Fd["colperm"], P["basis"], P["status"], colscale)
t2 = time.perf_counter()
print("through the host instead: factorize + download %.1f ms, prepare with upload %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
rhs = np.random.default_rng(1).standard_normal(m)
x = ctx.solve_dense(rhs, "n")
import scipy.sparse as sp
B = sp.csc_matrix((G["Bx"].copy(), G["Bi"].copy(), G["Bp"].copy()), shape=(m, m))
print("|B x - rhs| / (1 + |x|) = %.2e" % (np.abs(B @ x - rhs).max() / (1 + np.abs(x).max())), flush=True)
if "--cpu" in sys.argv:
    from oracle import pyoracle
    o = pyoracle.Oracle()
    t0 = time.perf_counter()
    Fo = o.lu_factorize(m, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1)
    print("CPU restatement: %.2f s; same factors: %s" % (time.perf_counter() - t0, all(np.array_equal(Fd[k], Fo[k]) for k in ("rowperm", "colperm")) and np.array_equal(Fd["U"].x, Fo["U"].x) and np.array_equal(Fd["L"].x, Fo["L"].x)), flush=True)
ctx.close()
