"""Is the first factorization of a process (cold: first launches of the dense LU's kernels, fresh buffers) bit-identical to the later ones?
A basis whose dense block has 12 000 rows (two rows per thread in the cooperative panel, look-ahead on)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import kkt, synth
dim, bump, dens = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (15000, 12000, 0.003)
os.environ["IPXK_LU_SPARSE"] = "t"
os.environ["IPXK_LU_BUMP_MAX"] = "20000"
G = synth.lp_like_basis_matrix(dim=dim, bump=bump, bump_density=dens, seed=7)
c = kkt.KktContext(synth.synthetic_lp(8, 12, 2, 1))
runs = []
for rep in range(4):
    F = c.lu_factorize(G["dim"], G["Bp"][:-1], G["Bp"][1:], G["Bi"], G["Bx"], 0.1, download=True)
    runs.append(F)
    print("call %d: bump %d, nnz(L) %d nnz(U) %d, bump phase %.1f ms" % (rep, F["bump"], F["L"].nnz, F["U"].nnz, F["seconds_bump"] * 1e3), flush=True)
a = runs[0]
for rep in range(1, 4):
    b = runs[rep]
    same_pattern = np.array_equal(a["L"].i, b["L"].i) and np.array_equal(a["U"].i, b["U"].i) and np.array_equal(a["rowperm"], b["rowperm"]) and np.array_equal(a["colperm"], b["colperm"])
    dl = int(np.count_nonzero(a["L"].x != b["L"].x)) if a["L"].x.shape == b["L"].x.shape else -1
    du = int(np.count_nonzero(a["U"].x != b["U"].x)) if a["U"].x.shape == b["U"].x.shape else -1
    print("call 0 vs call %d: pattern and permutations %s, differing values L %d U %d" % (rep, "equal" if same_pattern else "DIFFERENT", dl, du))
    if dl > 0:
        k = np.flatnonzero(a["L"].x != b["L"].x)[:5]
        print("   first differing L values", a["L"].x[k], b["L"].x[k], "rows", a["L"].i[k])
