import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
from oracle import pyoracle as po
orc = po.Oracle()

def check(m, n, seed=3, num_free=0, num_fixed=0, big=False):
    A0 = synth.synthetic_lp(m, n, 8, seed)
    B = synth.planted_lu_basis(A0, offdiag=3, seed=seed, num_free=num_free, num_fixed=num_fixed)
    A = B['A']
    st = synth.synthetic_ipm_state(m, n, 1.0, seed)
    colscale = np.sqrt(st['xl'] / st['zl'])
    status = B['status']
    colscale[status == 1] = np.inf
    colscale[status == -2] = 0.0
    ctx = kkt.KktContext(A)
    t0 = time.time()
    ctx.split_prepare(B['L'], B['U'], B['rowperm'], B['colperm'], B['basis'], status, colscale)
    print("m=%d n=%d prepare %.2fs levels %s" % (m, n, time.time() - t0, ctx.split_levels()))
    rng = np.random.default_rng(0)
    rhs = rng.standard_normal(m)
    if not big:
        AI = A.with_identity()
        cs = lambda M: po.Csc(M.nrow, M.ncol, M.p, M.i, M.x)
        os_ = orc.split_prepare(cs(AI), n, cs(B['L']), cs(B['U']), B['rowperm'], B['colperm'], B['basis'], status, colscale)
        pre = os_.get()
        Us = po.Csc(m, m, B['U'].p, B['U'].i, pre['Ux'])
        x1 = ctx.forward_solve(rhs); x2 = orc.forward_solve(cs(B['L']), Us, rhs)
        print(" forward bitwise", np.array_equal(x1, x2), "rel %.1e" % (np.abs(x1-x2).max()/np.abs(x2).max()))
        x1 = ctx.backward_solve(rhs); x2 = orc.backward_solve(cs(B['L']), Us, rhs)
        print(" backward bitwise", np.array_equal(x1, x2), "rel %.1e" % (np.abs(x1-x2).max()/np.abs(x2).max()))
        for tr in "NT":
            x1 = ctx.solve_dense(rhs, tr); x2 = os_.solve_dense(rhs, tr)
            print(" solve_dense", tr, "bitwise", np.array_equal(x1, x2), "rel %.1e" % (np.abs(x1-x2).max()/np.abs(x2).max()))
        l1, d1 = ctx.split_apply(rhs); l2, d2 = os_.apply(rhs)
        print(" split_apply rel %.1e dot rel %.1e" % (np.abs(l1-l2).max()/np.abs(l2).max(), abs(d1-d2)/abs(d2)))
        y1, it1, e1, h1, tm = ctx.cr_solve(rhs, 1e-8, None, -1, hist_cap=2000)
        y2, it2, e2, h2 = orc.cr_solve(lambda v: os_.apply(v), rhs, 1e-8, None, -1, hist_cap=2000)
        print(" cr_solve iters", it1, it2, "err", e1, e2, "rel %.1e" % (np.abs(y1-y2).max()/np.abs(y2).max()), "hist rel first5 %.1e" % (np.abs(h1[:5]-h2[:5])/h2[:5]).max())
        tol = 1e-6
        x1, yy1, it1, e1, tm = ctx.kkt_basis_solve(st['a'], st['b'], tol)
        x2, yy2, it2, e2, h = os_.kkt_solve(st['a'], st['b'], tol)
        print(" kkt_basis_solve iters", it1, it2, "err", e1, e2, "x rel %.1e y rel %.1e" % (np.abs(x1-x2).max()/np.abs(x2).max(), np.abs(yy1-yy2).max()/np.abs(yy2).max()))
    else:
        t0 = time.time(); l1, d1 = ctx.split_apply(rhs); print(" split_apply %.4fs" % (time.time()-t0))
        mu = st['mu']; tol = 0.3*np.sqrt(mu)
        for _ in range(2):
            t0 = time.time(); x1, yy1, it1, e1, tm = ctx.kkt_basis_solve(st['a'], st['b'], tol, 500); t1 = time.time()
            print(" kkt_basis_solve iters %d err %d total %.1f ms cr %.1f ms -> %.1f us/iter" % (it1, e1, (t1-t0)*1e3, tm.cr*1e3, tm.cr/max(it1,1)*1e6))
    ctx.close()

check(300, 600)
check(3000, 6000, num_free=5, num_fixed=7)
check(1000000, 2000000, seed=12345, big=True)
