"""The banded probe of the bench line (8 rows per column within a 4096-row band, C3 size) with the layout timings of ipxk_create."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IPXK_VERBOSE", "1")
import numpy as np
from ipx_amd import kkt, synth
m, n = 1000000, 2000000
A = synth.banded_lp(m, n, 8, 4096, 12345)
ctx = kkt.KktContext(A)
rng = np.random.default_rng(0)
ctx.normal_prepare(10.0 ** rng.uniform(-2, 2, n + m))
ctx.set_pointer_mode(True)
rhs, lhs = ctx.vector(m, rng.standard_normal(m)), ctx.vector(m)
ctx.time_normal_apply(rhs, lhs, 5)
ms = ctx.time_normal_apply(rhs, lhs, 50) / 50
print("layouts", ctx.spmv_layout()[0], "apply %.1f us = %.3f of the 8 TB/s roof" % (ms * 1e3, ctx.normal_apply_bytes / (ms * 1e-3) / 8e12))
