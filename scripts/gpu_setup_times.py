"""Set-up costs at C3: context creation (layouts + tuning), KKTSolverDiag::Factorize."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n = 1000000, 2000000
t0 = time.time(); A = synth.synthetic_lp(m, n, 8, 12345); st = synth.synthetic_ipm_state(m, n, 1.0, 12345)
print("generate %.2f s" % (time.time() - t0))
for layout in ("auto", "phased"):
    os.environ["IPXK_SPMV_LAYOUT"] = layout
    t0 = time.time(); ctx = kkt.KktContext(A); t1 = time.time()
    print("ipxk_create [%s]: %.2f s" % (layout, t1 - t0), ctx.spmv_layout())
    for k in range(3):
        t0 = time.time(); e = ctx.kkt_diag_factorize(st['xl'], st['xu'], st['zl'], st['zu'], st['mu']); t1 = time.time()
        print("  kkt_diag_factorize (host pointers): %.1f ms err %d" % ((t1 - t0) * 1e3, e))
    ctx.close()
