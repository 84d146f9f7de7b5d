import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n = int(sys.argv[1]), int(sys.argv[2])
A = synth.banded_lp(m, n, 8, int(os.environ['BAND']), 12345) if os.environ.get('BAND') else synth.synthetic_lp(m, n, 8, 12345)
ctx = kkt.KktContext(A)
rng = np.random.default_rng(0)
ctx.normal_prepare(10.0 ** rng.uniform(-2, 2, n + m))
y = ctx.vector(m, rng.standard_normal(m)); t = ctx.vector(n); lhs = ctx.vector(m)
ms = C.c_double(0)
for which, x, out in [(1, y, t), (2, t, lhs)]:
    ctx.lib.ipxk_debug_time_pass(ctx.h, which, x.as_arg(), out.as_arg(), 3, C.byref(ms))
    ctx.lib.ipxk_debug_time_pass(ctx.h, which, x.as_arg(), out.as_arg(), 20, C.byref(ms))
    print("  pass%d %.1f us" % (which, ms.value / 20 * 1e3), end="")
print(" | slice %s maxwg %s band %s" % (os.environ.get("IPXK_SLICE_KB"), os.environ.get("IPXK_MAX_WG"), os.environ.get("BAND")), flush=True)
