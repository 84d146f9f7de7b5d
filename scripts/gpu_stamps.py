import os, sys, ctypes as C
os.environ["IPXK_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n = 1000000, 2000000
A = synth.synthetic_lp(m, n, 8, 12345)
ctx = kkt.KktContext(A)
rng = np.random.default_rng(0)
ctx.normal_prepare(10.0 ** rng.uniform(-2, 2, n + m))
y = ctx.vector(m, rng.standard_normal(m)); t = ctx.vector(n); lhs = ctx.vector(m)
ms = C.c_double(0)
for which, x, out in [(1, y, t), (2, t, lhs)]:
    ctx.lib.ipxk_debug_time_pass(ctx.h, which, x.as_arg(), out.as_arg(), 3, C.byref(ms))
    ctx.lib.ipxk_debug_time_pass(ctx.h, which, x.as_arg(), out.as_arg(), 1, C.byref(ms))
    geom = (C.c_int * 4)()
    buf = np.zeros(1 << 20, np.uint64)
    ctx.lib.ipxk_debug_get_stamps(ctx.h, which, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.c_int64(buf.size), geom)
    P, G, RT, Q = list(geom)
    st = buf[:P * G].reshape(P, G).astype(np.float64)
    busy = st[0] > 0
    t0 = st[0][busy].min()
    print("pass%d: %.1f us  P=%d G=%d RT=%d busy=%d" % (which, ms.value * 1e3, P, G, RT, busy.sum()))
    for p in range(P):
        v = (st[p][busy] - t0) / 100.0   # 100 MHz -> us
        print("  phase %2d start: min %7.1f med %7.1f max %7.1f us" % (p, v.min(), np.median(v), v.max()))
    info = buf[P * G: P * G + G]
    xcc = (info >> 32) & 0xf
    hw = info & 0xffffffff
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    dur = (st[P - 1] - st[0]) / 100.0
    print("  duration (phase0->last phase start) per XCC:")
    for xc in range(8):
        sel = busy & (xcc == xc)
        if sel.sum(): print("    xcc %d: n=%d  min %.1f med %.1f max %.1f" % (xc, sel.sum(), dur[sel].min(), np.median(dur[sel]), dur[sel].max()))
    key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
    uk, cnt = np.unique(key[busy], return_counts=True)
    print("  distinct (xcc,se,sh,cu):", uk.size, "busy WGs per CU: min %d max %d" % (cnt.min(), cnt.max()), "hist", np.bincount(cnt))
    per_cu_n = dict(zip(uk, cnt))
    nn = np.array([per_cu_n[k_] for k_ in key[busy]])
    for c_ in np.unique(nn):
        print("    WGs on CUs with %d busy WGs: median dur %.1f" % (c_, np.median(dur[busy][nn == c_])))
