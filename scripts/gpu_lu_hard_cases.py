"""Bases whose bump exceeds the dense limit: tearing (IPXK_LU_SPARSE=0) against elimination rounds (=1) on the device.
Prints time (second call: workspaces allocated), pivots, fill = (nnz(L) + nnz(U)) / nnz(B).
usage: python scripts/gpu_lu_hard_cases.py [small]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt

small = len(sys.argv) > 1 and sys.argv[1] == "small"
cases = [("60k rows, 40 exchanges (the test's basis), dense limit 2048", lambda: synth.disturbed_basis_matrix(seed=5, dim=60000, num_exchanged=40, bump=100, offdiag=3), 2048),
         ("200k rows, planted bump 100, 200 misplaced columns", lambda: synth.misplaced_basis_matrix(200000, 200, seed=12345, bump=100), 8192),
         ("200k rows, planted bump 1000, 200 misplaced columns", lambda: synth.misplaced_basis_matrix(200000, 200, seed=12345, bump=1000), 8192)]
if not small:
    cases += [("1M rows, planted bump 100, 200 misplaced columns", lambda: synth.misplaced_basis_matrix(1000000, 200, seed=12345, bump=100), 8192),
              ("1M rows, planted bump 1000, 200 misplaced columns", lambda: synth.misplaced_basis_matrix(1000000, 200, seed=12345, bump=1000), 8192)]
ctx = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
for name, make, limit in cases:
    G = make()
    m, nb = G["dim"], len(G["Bi"])
    print("%s: nnz(B) %d" % (name, nb), flush=True)
    os.environ["IPXK_LU_BUMP_MAX"] = str(limit)
    for mode, label in (("0", "tearing"), ("1", "elimination rounds")):
        os.environ["IPXK_LU_SPARSE"] = mode
        try:
            for rep in range(2):
                t0 = time.perf_counter()
                F = ctx.lu_factorize(m, G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, download=False)
                dt = time.perf_counter() - t0
            print("  %-18s %8.1f ms  fill %6.2f  singletons %d + %d, spikes %d, sparse pivots %d in %d rounds, dense block %d (%d dependent), rounds %d" %
                  (label, dt * 1e3, (F["lnz"] + F["unz"]) / nb, F["col_singletons"], F["row_singletons"], F["spikes"], F["sparse_pivots"], F["sparse_rounds"],
                   F["bump"], F["num_dependent"], F["rounds"]), flush=True)
        except kkt.KktError as e:
            print("  %-18s refused: %s" % (label, str(e)[:160]), flush=True)
ctx.close()
