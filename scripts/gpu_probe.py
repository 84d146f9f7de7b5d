"""Exploratory GPU check: HIP path vs oracle on small + medium synthetic LPs; SpMV timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
from oracle import pyoracle as po

orc = po.Oracle()

def check(m, n, seed=1, spread=1.0, ndense=0):
    A = synth.synthetic_lp(m, n, 8, seed, num_dense=ndense)
    st = synth.synthetic_ipm_state(m, n, spread, seed)
    Ac = po.Csc(m, n, A.p, A.i, A.x)
    ctx = kkt.KktContext(A)
    print("m=%d n=%d nnz=%d dense=%d" % (m, n, A.nnz, ctx.num_dense_cols))
    # index parity of row-wise copy
    AT = orc.transpose(Ac)
    p, i, x = ctx.get_rowwise()
    print(" rowwise bit-exact:", np.array_equal(p, AT.p), np.array_equal(i, AT.i), np.array_equal(x, AT.x))
    W = st['xl'] / st['zl']
    rng = np.random.default_rng(0)
    rhs = rng.standard_normal(m)
    ctx.normal_prepare(W)
    l1, d1 = ctx.normal_apply(rhs)
    l2, d2 = orc.normal_apply(Ac, W, rhs)
    print(" normal_apply: bitwise", np.array_equal(l1, l2), "relerr %.2e" % (np.abs(l1-l2).max()/np.abs(l2).max()), "dot rel %.2e" % (abs(d1-d2)/abs(d2)))
    nzd = orc.find_dense_columns(Ac)[1]
    e = ctx.diag_factorize(W, True)
    P, e2 = orc.diag_factorize(Ac, W, nzd, True)
    l1, d1 = ctx.diag_apply(rhs); l2, d2 = P.apply(rhs)
    print(" diag_apply: err", e, e2, "bitwise", np.array_equal(l1, l2), "relerr %.2e" % (np.abs(l1-l2).max()/np.abs(l2).max()), "dot rel %.2e" % (abs(d1-d2)/abs(d2)))
    # kkt diag
    mu = st['mu']
    e = ctx.kkt_diag_factorize(st['xl'], st['xu'], st['zl'], st['zu'], mu)
    ko = orc.kkt_diag(Ac, maxiter=500); ko.factorize(st['xl'], st['xu'], st['zl'], st['zu'], mu)
    W1, r1 = ctx.kkt_diag_get(); W2, r2 = ko.get()
    print(" W bitwise", np.array_equal(W1, W2), "resscale bitwise", np.array_equal(r1, r2))
    tol = 0.3*np.sqrt(mu)
    t0=time.time(); x1, y1, it1, e1, tm = ctx.kkt_diag_solve(st['a'], st['b'], tol, 500); t1=time.time()
    x2, y2, it2, e2, h = ko.solve(st['a'], st['b'], tol, hist_cap=600); t2=time.time()
    print(" kkt_diag_solve: iters", it1, it2, "err", e1, e2, "y rel %.2e x rel %.2e" % (np.abs(y1-y2).max()/np.abs(y2).max(), np.abs(x1-x2).max()/np.abs(x2).max()), "gpu %.4fs (cr %.4fs) cpu %.4fs" % (t1-t0, tm.cr, t2-t1))
    # pcr with history
    rhs2 = rng.standard_normal(m)
    ctx.normal_prepare(W1); 
    yl, it, err, hist, tm = ctx.pcr_solve(rhs2, 1e-6, r1, 500, hist_cap=600)
    yo, ito, erro, histo = orc.pcr_solve(lambda v: orc.normal_apply(Ac, W2, v), lambda v: ko_apply(v), rhs2, 1e-6, r2, 500, hist_cap=600) if False else (None,0,0,None)
    print(" pcr: it", it, "err", err, "hist[:3]", hist[:3], "len", len(hist))
    ctx.close()
    return

def spmv_time(m, n):
    A = synth.synthetic_lp(m, n, 8, 12345)
    st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
    ctx = kkt.KktContext(A)
    W = st['xl']/st['zl']
    ctx.normal_prepare(W)
    rng = np.random.default_rng(0)
    rhs = ctx.vector(m, rng.standard_normal(m)); lhs = ctx.vector(m)
    ctx.time_normal_apply(rhs, lhs, 5)
    reps = 50
    ms = ctx.time_normal_apply(rhs, lhs, reps)
    B = ctx.normal_apply_bytes
    print("SpMV m=%d n=%d: %.1f us/apply, %.1f MB algorithmic, %.2f TB/s (%.1f%% of 8 TB/s)" % (m, n, ms/reps*1e3, B/1e6, B/(ms/reps*1e-3)/1e12, B/(ms/reps*1e-3)/8e12*100))
    # kkt solve resident
    e = ctx.kkt_diag_factorize(st['xl'], st['xu'], st['zl'], st['zu'], st['mu'])
    ctx.set_pointer_mode(True)
    a = ctx.vector(n+m, st['a']); b = ctx.vector(m, st['b']); x = ctx.vector(n+m); y = ctx.vector(m)
    tol = 0.3*np.sqrt(st['mu'])
    it, err, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, 500)
    t0 = time.time(); K = 5
    for _ in range(K): it, err, tm = ctx.kkt_diag_solve_resident(a, b, x, y, tol, 500)
    t1 = time.time()
    print(" kkt_diag_solve resident: iters %d err %d %.3f ms/solve (cr %.3f ms) -> %.1f us/iter" % (it, err, (t1-t0)/K*1e3, tm.cr*1e3, tm.cr/max(it,1)*1e6))
    ctx.close()

check(200, 400)
check(2000, 4000)
check(3000, 6000, ndense=4)
spmv_time(50000, 100000)
spmv_time(1000000, 2000000)
