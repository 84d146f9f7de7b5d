"""Dense LU of a bump: the cooperative outer panel (default) against the one-workgroup sub-panel launches (IPXK_LU_COOP=0).
Bumps sent to the dense code as they stand (IPXK_LU_SPARSE=t), and the IPM basis of the fixture under the default policy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ipx_amd import kkt, synth
c = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
sizes = ((4000, 2600, 0.02), (7000, 5000, 0.01), (10000, 8000, 0.005), (15000, 12000, 0.003))
if len(sys.argv) > 1: sizes = tuple(t for t in sizes if t[1] == int(sys.argv[1]))          # one size only (profiling)
cases = [("bump %d" % b, synth.lp_like_basis_matrix(dim=d, bump=b, bump_density=dens, seed=7), "t") for d, b, dens in sizes]
g = np.load(os.path.join(ROOT, "tests", "golden", "ipm_basis_16000.npz"))
if len(sys.argv) <= 1: cases.append(("IPM basis 16000, default policy", dict(dim=int(g["dim"]), Bp=g["Bp"].astype(np.int64), Bi=g["Bi"].astype(np.int64), Bx=g["Bx"]), None))
modes = ("1", "r2", "0") if len(sys.argv) <= 2 else (sys.argv[2],)
for name, G, sparse in cases:
    if sparse: os.environ["IPXK_LU_SPARSE"] = sparse
    else: os.environ.pop("IPXK_LU_SPARSE", None)
    os.environ["IPXK_LU_BUMP_MAX"] = "20000"
    for coop in modes:
        os.environ.pop("IPXK_LU_COOP_R", None)
        if coop[0] == "r": os.environ["IPXK_LU_COOP_R"] = coop[1]
        os.environ["IPXK_LU_COOP"] = "0" if coop == "0" else "1"
        best = None
        for rep in range(3):
            F = c.lu_factorize(G["dim"], G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, download=False)
            t = F["seconds_bump"] * 1e3
            best = t if best is None else min(best, t)
        print("%-34s coop %-2s: dense block %5d rows, bump phase %7.1f ms (singletons %.1f, assembly %.1f)" % (name, coop, F["bump"], best, F["seconds_singletons"] * 1e3, F["seconds_assemble"] * 1e3), flush=True)
