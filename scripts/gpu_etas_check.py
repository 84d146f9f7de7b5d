"""Maxvolume that keeps its last exchanges as etas behind the factors (ipxk_maxvolume_info.kept_etas) against the same run with the
final refactorization: the operator by basis position, SolveDense, the KKT solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ipx_amd import synth, kkt
from oracle import pyoracle as po
from test_maxvolume_oracle import setup, basis_matrix
m, n, bump, seed = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1200, 2600, 60, 8)
P, status, colscale, Ao = setup(po, m, n, bump, seed)
st = synth.synthetic_ipm_state(m, n, 1.0, seed)
res = {}
for keep in ("1", "0"):
    os.environ["IPXK_MAXVOL_KEEP_ETAS"] = keep
    ctx = kkt.KktContext(P["A"])
    F0 = ctx.lu_factorize_basis(P["basis"], 0.1)
    ctx.split_prepare_lu(status, colscale)
    got = ctx.maxvolume(status, colscale, volume_tol=2.0, maxskip_updates=10, rows_per_slice=100, max_etas=100)
    print("keep", keep, "updates", got["updates"], "factorizations", got["factorizations"], "kept_etas", got["kept_etas"])
    colperm = F0["colperm"] if got["kept_etas"] > 0 else None
    if colperm is None:
        F1 = ctx.lu_factorize_basis(got["basis"], 0.1)
        ctx.split_prepare_lu(got["status"], colscale)
        colperm = F1["colperm"]
    rng = np.random.default_rng(3)
    v_pos = rng.standard_normal(m)
    lhs, dot = ctx.split_apply(v_pos[colperm])
    out_pos = np.zeros(m); out_pos[colperm] = lhs
    Bm = basis_matrix(Ao, got["basis"])
    rhs = rng.standard_normal(m)
    xn, xt = ctx.solve_dense(rhs, "n"), ctx.solve_dense(rhs, "t")
    print("   SolveDense residuals N %.1e T %.1e" % (np.abs(Bm @ xn - rhs).max(), np.abs(Bm.T @ xt - rhs).max()))
    x, y, it, e, _ = ctx.kkt_basis_solve(st["a"], st["b"], 1e-9, 500)
    print("   kkt solve iters %d err %d" % (it, e))
    res[keep] = (out_pos, dot, x, y, got["basis"])
    ctx.close()
a, b = res["1"], res["0"]
assert np.array_equal(a[4], b[4])
print("operator by position: relerr %.2e, dot %.2e" % (np.abs(a[0] - b[0]).max() / np.abs(b[0]).max(), abs(a[1] - b[1]) / abs(b[1])))
print("kkt solve: x relerr %.2e y relerr %.2e" % (np.abs(a[2] - b[2]).max() / np.abs(b[2]).max(), np.abs(a[3] - b[3]).max() / np.abs(b[3]).max()))
