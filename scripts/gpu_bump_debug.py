import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from ipx_amd import synth, kkt
m, n, bump = (int(a) for a in sys.argv[1:4])
P = synth.lp_like_basis(m, n, seed=4, bump=bump)
colscale = np.ones(n + m)
ctx = kkt.KktContext(P["A"])
F = ctx.lu_factorize_basis(P["basis"], 0.1)
ctx.split_prepare_lu(P["status"], colscale)
print("levels", ctx.split_levels(), "bump", F["bump"])
AI = sp.hstack([P["A"].to_scipy(), sp.identity(m)]).tocsc()
B = AI[:, P["basis"]]
rhs = np.random.default_rng(2).standard_normal(m)
for trans in ("n", "t"):
    x = ctx.solve_dense(rhs, trans)
    r = (B.T if trans == "t" else B) @ x - rhs
    print(trans, "residual", np.abs(r).max(), "worst rows", np.argsort(-np.abs(r))[:5], np.sort(-np.abs(r))[:5])
# stage of each row / column
rp, cp = F["rowperm"], F["colperm"]
s0 = m - F["bump"]
print("bump rows", sorted(rp[s0:])[:10], "bump cols(positions)", sorted(cp[s0:])[:10])
colscale = synth.synthetic_basis_state(P["status"], 1.0, 4)
ctx.split_prepare_lu(P["status"], colscale)
f1, b1 = ctx.forward_solve(rhs), ctx.backward_solve(rhs)
l1, d1 = ctx.split_apply(rhs)
for trans in ("n", "t"):
    x = ctx.solve_dense(rhs, trans)
    r = (B.T if trans == "t" else B) @ x - rhs
    print("after the scaled prepare: unscaled", trans, "residual", np.abs(r).max())
# expected scaled forward solve in pivot order: U D x = (L+I)^-1 r  (identity row permutation is folded elsewhere: compare norms only)
Ls = sp.csc_matrix((F["L"].x, F["L"].i, F["L"].p), shape=(m, m)); Us = sp.csc_matrix((F["U"].x, F["U"].i, F["U"].p), shape=(m, m))
import scipy.sparse.linalg as spl
d = np.where(P["status"][P["basis"][F["colperm"]]] == 0, colscale[P["basis"][F["colperm"]]], 1.0)
w = spl.spsolve_triangular((Ls + sp.identity(m)).tocsr(), rhs, lower=True)
xe = spl.spsolve_triangular(Us.tocsr(), w, lower=False) / d
print("expected (pivot order) vs lu path", np.abs(f1 - xe).max() / np.abs(xe).max())
ctx.split_prepare(F["L"], F["U"], F["rowperm"], F["colperm"], P["basis"], P["status"], colscale)
f2, b2 = ctx.forward_solve(rhs), ctx.backward_solve(rhs)
l2, d2 = ctx.split_apply(rhs)
rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
print("scaled: forward", rel(f1, f2), "backward", rel(b1, b2), "apply", rel(l1, l2))
print("expected vs host path", np.abs(f2 - xe).max() / np.abs(xe).max())
s0 = m - F["bump"]
print("per unknown ratio lu/host at bump:", (f1[s0:] / f2[s0:])[:8], " d at bump:", d[s0:][:8])
print("ratio before bump (first 8 with largest diff):", [(int(i), f1[i] / f2[i]) for i in np.argsort(-np.abs(f1 - f2))[:8]])
print("forward diffs at", np.argsort(-np.abs(f1 - f2))[:8], "stage of those", [int(np.nonzero(F["colperm"] == i)[0][0]) if False else 0 for i in range(1)])
