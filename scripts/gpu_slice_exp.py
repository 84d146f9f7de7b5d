"""NormalMatrix apply at C3 with the sliced layout forced and different slice sizes (IPXK_SLICE_TEST_KB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n = 1000000, 2000000
A = synth.synthetic_lp(m, n, 8, 12345)
ctx = kkt.KktContext(A)
rng = np.random.default_rng(0)
ctx.normal_prepare(10.0 ** rng.uniform(-2, 2, n + m))
ctx.set_pointer_mode(True)
rhs, lhs = ctx.vector(m, rng.standard_normal(m)), ctx.vector(m)
ctx.time_normal_apply(rhs, lhs, 5)
ms = ctx.time_normal_apply(rhs, lhs, 50) / 50
print("IPXK_SLICE_TEST_KB=%s FORCE2=%s layouts %s: %.1f us per apply" % (os.environ.get("IPXK_SLICE_TEST_KB"), os.environ.get("IPXK_SLICE_FORCE2"), ctx.spmv_layout()[0], ms * 1e3), flush=True)
