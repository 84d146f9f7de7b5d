"""CPU study behind profiles/r05_lu_fill_study.txt: how much fill the LU of an IPM basis of a random sparse LP MUST have.
  * tests/golden/ipm_basis_16000.npz (a basis of the reference's IPM on general_lp(16000, 40000, 31)) and a model of such bases
    (random structural columns of 8 entries + slack columns on the rows a maximum matching leaves free);
  * SuperLU (scipy.sparse.linalg.splu) with COLAMD / MMD orderings, threshold 0.1;
  * the sequential minimum-Markowitz elimination of the pattern (scripts/markowitz_symbolic.cc).
usage: python scripts/lu_fill_study.py [--superlu]      (SuperLU at 16000 rows takes minutes)"""
import os, subprocess, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as sla
from scipy.sparse.csgraph import maximum_bipartite_matching
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = "/tmp/markowitz_symbolic"
subprocess.check_call(["g++", "-O2", "-w", "-o", exe, os.path.join(ROOT, "scripts", "markowitz_symbolic.cc")])
rng = np.random.default_rng(1)


def model_basis(m, ns, k=8):
    rows = np.concatenate([rng.choice(m, k, replace=False) for _ in range(ns)])
    A = sp.csc_matrix((rng.uniform(0.5, 4, ns * k) * rng.choice([-1, 1], ns * k), (rows, np.repeat(np.arange(ns), k))), shape=(m, ns))
    match = maximum_bipartite_matching(A.tocsr(), perm_type="row")
    free = np.setdiff1d(np.arange(m), match)
    return sp.hstack([A, sp.csc_matrix((np.ones(m - ns), (free, np.arange(m - ns))), shape=(m, m - ns))]).tocsc()


def markowitz(B):
    C = B.tocoo()
    with open("/tmp/pattern.bin", "wb") as f:
        np.array([B.shape[0]], np.int32).tofile(f); np.array([C.nnz], np.int64).tofile(f)
        C.row.astype(np.int32).tofile(f); C.col.astype(np.int32).tofile(f)
    r = subprocess.run([exe, "/tmp/pattern.bin"], capture_output=True, text=True)
    dense = [ln for ln in r.stderr.splitlines() if ln.startswith("dense")]
    return r.stdout.strip() + (" | " + dense[0] if dense else "")


g = np.load(os.path.join(ROOT, "tests", "golden", "ipm_basis_16000.npz"))
cases = [("IPM basis 16000 (fixture)", sp.csc_matrix((g["Bx"], g["Bi"], g["Bp"]), shape=(int(g["dim"]),) * 2)),
         ("model basis 4000 (2850 structural)", model_basis(4000, 2850)), ("model basis 16000 (11400 structural)", model_basis(16000, 11400))]
for name, B in cases:
    print(name, "nnz(B)", B.nnz)
    print("   sequential minimum-Markowitz (pattern only):", markowitz(B), flush=True)
    if "--superlu" in sys.argv:
        for spec in ("COLAMD", "MMD_AT_PLUS_A"):
            t = time.time()
            lu = sla.splu(B, permc_spec=spec, diag_pivot_thresh=0.1)
            print("   SuperLU %-14s nnz(L+U) %d = %.0f x nnz(B)  (%.0f s)" % (spec, lu.L.nnz + lu.U.nnz, (lu.L.nnz + lu.U.nnz) / B.nnz, time.time() - t), flush=True)
