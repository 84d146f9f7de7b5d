"""Generates the committed golden vectors tests/golden/*.npz by driving the REFERENCE'S OWN
objects (oracle/_ref/libipx_ref.so, built by oracle/Makefile from /root/reference/src).

Run in the build container only:   python tests/golden/make_golden.py
The fixtures are data (inputs + the reference's outputs); nothing of the reference's source
is stored.  afiro's model data are the arrays of the reference's example (example/afiro.cc:12-46).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ipx_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
import helpers  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
INF = np.inf


def afiro():
    obj = [-0.2194, 0, 0, 0, 0, 0, 0, 0, -0.32, -0.5564, 0.6, -0.48]
    ub = [80.0, 283.303, 283.303, 312.813, 349.187, INF, INF, INF, 57.201, 500.0, 500.501, 357.501]
    Ap = [0, 2, 6, 10, 14, 18, 20, 22, 24, 26, 28, 30, 32]
    Ai = [0, 5, 1, 6, 7, 8, 2, 6, 7, 8, 3, 6, 7, 8, 4, 6, 7, 8, 1, 2, 2, 3, 2, 4, 0, 6, 0, 5, 2, 5, 5, 7]
    Ax = [-1.0, 0.301, 1.0, -1.0, 0.301, 1.06, 1.0, -1.0, 0.313, 1.06, 1.0, -1.0, 0.313, 0.96, 1.0,
          -1.0, 0.326, 0.86, -1.0, 0.99078, 1.00922, -1.0, 1.01802, -1.0, 1.4, 1.0, 0.109, -1.0,
          -0.419111, 1.0, 1.4, -1.0]
    rhs = [0.0, 80.0, 0.0, 0.0, 0.0, 0.0, 0.0, 44.0, 300.0]
    ct = "<<=<<=<<<"
    A = po.Csc(9, 12, Ap, Ai, Ax)
    return A, np.array(rhs), ct, np.array(obj), np.zeros(12), np.array(ub)


def model_arrays(rm):
    AI = rm.AI()
    return dict(m=rm.m, n=rm.n, AIp=AI.p, AIi=AI.i, AIx=AI.x, num_dense=rm.num_dense)


def gen_afiro(ref):
    A, rhs, ct, obj, lb, ub = afiro()
    rm = ref.model(A, rhs, ct, obj, lb, ub)
    d = model_arrays(rm)
    d["dualized"] = rm.dualized
    AIt = rm.AIt()
    d.update(AItp=AIt.p, AIti=AIt.i, AItx=AIt.x)
    m, n = rm.m, rm.n
    rng = np.random.default_rng(7)
    a, b = rng.uniform(-1, 1, n + m), rng.uniform(-1, 1, m)
    k = rm.kkt_diag(maxiter=-1)
    err, _ = k.factorize()            # Factorize(nullptr): G = identity
    x, y, it, e = k.solve(a, b, 1e-8)
    d.update(a=a, b=b, tol=1e-8, x=x, y=y, iter=it, errflag=e, fact_err=err)
    # with an interior iterate
    lbv, ubv = rm.vectors()[2], rm.vectors()[3]
    xl = np.where(np.isfinite(lbv), 10.0 ** rng.uniform(-1, 1, n + m), INF)
    xu = np.where(np.isfinite(ubv), 10.0 ** rng.uniform(-1, 1, n + m), INF)
    zl = np.where(np.isfinite(lbv), 10.0 ** rng.uniform(-1, 1, n + m), 0.0)
    zu = np.where(np.isfinite(ubv), 10.0 ** rng.uniform(-1, 1, n + m), 0.0)
    err, mu = k.factorize(np.ones(n + m), xl, xu, np.zeros(m), zl, zu)
    x2, y2, it2, e2 = k.solve(a, b, 1e-6)
    d.update(xl=xl, xu=xu, zl=zl, zu=zu, mu=mu, tol2=1e-6, x2=x2, y2=y2, iter2=it2, errflag2=e2)
    np.savez_compressed(os.path.join(OUT, "afiro.npz"), **d)


def gen_diag(ref, name, m, n, seed, num_dense, spread=1.0):
    A, st = helpers.diag_problem(m, n, seed=seed, spread=spread, num_dense=num_dense)
    v = synth.lp_vectors(m, n)
    rm = ref.model(po.Csc(m, n, A.p, A.i, A.x), v["rhs"], v["constr_type"], v["obj"], v["lb"], v["ub"])
    assert rm.m == m and rm.n == n and not rm.dualized
    d = dict(m=m, n=n, Ap=A.p, Ai=A.i, Ax=A.x, num_dense=rm.num_dense,
             is_dense=np.array([rm.is_dense(j) for j in range(n)], dtype=np.int8))
    AIt = rm.AIt()
    d.update(AItp=AIt.p, AIti=AIt.i, AItx=AIt.x)
    W = st["xl"] / st["zl"]
    rng = np.random.default_rng(seed + 100)
    rhs = rng.standard_normal(m)
    lhs, dot = rm.normal_apply(W, rhs)
    d.update(W=W, rhs=rhs, normal_lhs=lhs, normal_dot=dot)
    lhs0, dot0 = rm.normal_apply(None, rhs)      # W == NULL variant
    d.update(normal0_lhs=lhs0, normal0_dot=dot0)
    pl, pd, pe = rm.diagprec_apply(W, True, rhs)
    d.update(prec_lhs=pl, prec_dot=pd, prec_err=pe)
    resscale = 1.0 / np.sqrt(W[n:])
    y, it, e, ch, ph = rm.pcr_solve(W, True, rhs, 1e-7, resscale, 500)
    d.update(resscale=resscale, pcr_tol=1e-7, pcr_y=y, pcr_iter=it, pcr_err=e, pcr_cdot=ch, pcr_pdot=ph)
    # iteration limit -> 201
    y, it, e, ch, ph = rm.pcr_solve(W, True, rhs, 1e-30, resscale, 7)
    d.update(lim_iter=it, lim_err=e, lim_y=y)
    # indefinite weights -> the matrix is not positive definite -> 202
    Wneg = W.copy()
    Wneg[: n // 2] *= -1.0
    y, it, e, ch, ph = rm.pcr_solve(Wneg, False, rhs, 1e-12, None, 200)
    d.update(Wneg=Wneg, neg_iter=it, neg_err=e)
    # KKTSolverDiag with the interior iterate
    k = rm.kkt_diag(maxiter=500)
    err, mu = k.factorize(np.ones(n + m), st["xl"], st["xu"], np.zeros(m), st["zl"], st["zu"])
    tol = 0.3 * np.sqrt(mu)
    x, yk, itk, ek = k.solve(st["a"], st["b"], tol)
    d.update(xl=st["xl"], xu=st["xu"], zl=st["zl"], zu=st["zu"], mu=mu, a=st["a"], b=st["b"],
             kkt_tol=tol, kkt_x=x, kkt_y=yk, kkt_iter=itk, kkt_err=ek, kkt_fact_err=err)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)


def gen_basis(ref, name, m, n, seed, num_free, num_fixed):
    B, st, colscale = helpers.basis_problem(m, n, seed=seed, num_free=num_free, num_fixed=num_fixed)
    A, L, U = B["A"], B["L"], B["U"]
    cs = lambda M: po.Csc(M.nrow, M.ncol, M.p, M.i, M.x)
    AI = A.with_identity()
    status, basis, rowperm, colperm = B["status"], B["basis"], B["rowperm"], B["colperm"]
    rng = np.random.default_rng(seed + 200)
    d = dict(m=m, n=n, Ap=A.p, Ai=A.i, Ax=A.x, Lp=L.p, Li=L.i, Lx=L.x, Up=U.p, Ui=U.i, Ux=U.x,
             rowperm=rowperm, colperm=colperm, basis=basis, status=status, colscale=colscale)
    # reference kernels on explicit inputs
    rpi = ref.inverse_perm(rowperm)
    AT = ref.transpose(cs(L))
    d.update(rowperm_inv=rpi, LTp=AT.p, LTi=AT.i, LTx=AT.x)
    x0 = rng.standard_normal(m)
    d["x0"] = x0
    for trans, uplo, unit, key, T in (("t", "u", 0, "Ut", U), ("t", "l", 1, "Lt", L),
                                      ("n", "l", 1, "Lf", L), ("n", "u", 0, "Uf", U)):
        xs, nz = ref.trisolve(cs(T), x0, trans, uplo, unit)
        d["tri_" + key] = xs
    d["fwd"] = ref.forward_solve(cs(L), cs(U), x0)
    d["bwd"] = ref.backward_solve(cs(L), cs(U), x0)
    # the operator of splitted_normal_matrix.cc:90-117 from explicit factors: build the scaled U
    # and N exactly as Prepare does (:30-55) with the reference's CopyColumns/PermuteRows/ScaleColumn
    Ux = U.x.copy()
    free_positions = []
    for k in range(m):
        j = basis[colperm[k]]
        if status[j] == 0:
            Ux[U.p[k]:U.p[k + 1]] *= colscale[j]
        elif status[j] == 1:
            free_positions.append(k)
    Us = po.Csc(m, m, U.p, U.i, Ux)
    nonbasic = np.nonzero(status == -1)[0]
    N = ref.copy_permute_scale(cs(AI), nonbasic, rpi, colscale[nonbasic])
    d.update(Ux_scaled=Ux, free_positions=np.array(free_positions, dtype=np.int64), Np=N.p, Ni=N.i, Nx=N.x)
    op = ref.split(cs(L), Us, N, free_positions)
    lhs, dot = op.apply(x0)
    d.update(split_lhs=lhs, split_dot=dot)
    # CR right-hand sides vanish at free positions (kkt_solver_basis.cc:128-138)
    rhs_cr = x0.copy()
    rhs_cr[free_positions] = 0.0
    y, it, e, ch = op.cr_solve(rhs_cr, 1e-9, -1)
    d.update(cr_rhs=rhs_cr, cr_tol=1e-9, cr_y=y, cr_iter=it, cr_err=e, cr_cdot=ch)
    y, it, e, ch = op.cr_solve(rhs_cr, 1e-30, 5)
    d.update(cr_lim_iter=it, cr_lim_err=e)
    d.update(a=st["a"], b=st["b"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)


def gen_iterate(ref, name, m, n, seed):
    """The reference's ipx::Iterate: states, residuals, complementarity, and Update with two step sizes."""
    P = synth.synthetic_iterate(m, n, seed)
    A = P["A"]
    rm = ref.model(po.Csc(m, n, A.p, A.i, A.x), P["rhs"], P["constr_type"], P["obj"], P["lb"], P["ub"])
    b, c, lbs, ubs = rm.vectors()
    ri = rm.iterate()
    ri.initialize(P["it"])
    d = dict(m=m, n=n, Ap=A.p, Ai=A.i, Ax=A.x, b=b, c=c, lbs=lbs, ubs=ubs, state=ri.states())
    d.update({"it_" + k: v for k, v in P["it"].items()})
    d.update({"step_" + k: v for k, v in P["step"].items()})
    r = ri.residuals()
    d.update(rb=r["rb"], rc=r["rc"], rl=r["rl"], ru=r["ru"], presidual=r["presidual"], dresidual=r["dresidual"])
    comp = ri.complementarity()
    d.update(complementarity=comp["complementarity"], mu=comp["mu"], mu_min=comp["mu_min"], mu_max=comp["mu_max"])
    st = P["step"]
    for tag, sp_, sd_ in (("a", 0.7, 0.4), ("b", 3.0, 5.0)):       # b: truncation at kBarrierMin
        ri.initialize(P["it"])
        ri.update(sp_, st["dx"], st["dxl"], st["dxu"], sd_, st["dy"], st["dzl"], st["dzu"])
        d.update({"upd_%s_%s" % (tag, k): v for k, v in ri.get().items()})
        d["upd_%s_sp" % tag], d["upd_%s_sd" % tag] = sp_, sd_
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)


def main():
    po.build(ref=True)
    ref = po.Ref()
    gen_iterate(ref, "iterate_150", 150, 360, seed=14)
    gen_afiro(ref)
    gen_diag(ref, "diag_200", 200, 400, seed=11, num_dense=0)
    gen_diag(ref, "dense_300", 300, 640, seed=12, num_dense=4)
    gen_basis(ref, "basis_200", 200, 420, seed=13, num_free=3, num_fixed=4)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
