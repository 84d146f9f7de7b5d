"""The banded-matrix probe of the bench line on its own (for rocprofv3: kernel trace / --pmc FETCH_SIZE / WRITE_SIZE):
NormalMatrix apply on banded_lp(1M, 2M, 8 rows per column within a 4096-row band)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ipx_amd import synth, kkt
m, n = 1000000, 2000000
A = synth.banded_lp(m, n, 8, 4096, 12345)
ctx = kkt.KktContext(A)
rng = np.random.default_rng(0)
ctx.normal_prepare(10.0 ** rng.uniform(-2, 2, n + m))
ctx.set_pointer_mode(True)
rhs, lhs = ctx.vector(m, rng.standard_normal(m)), ctx.vector(m)
ctx.time_normal_apply(rhs, lhs, 5)
ms = ctx.time_normal_apply(rhs, lhs, 20) / 20
nb = 2 * A.nnz * 12 + (n + m + 2) * 4 + 8 * (3 * n + 4 * m)
print("banded probe: layouts %s, %.1f us per apply, %.0f GB/s algorithmic (%.3f of 8 TB/s)" % (ctx.spmv_layout()[0], ms * 1e3, nb / (ms * 1e-3) / 1e9, nb / (ms * 1e-3) / 8e12), flush=True)
