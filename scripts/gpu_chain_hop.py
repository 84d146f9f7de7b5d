"""Cost of one dependency hop in the sweeps: forward / backward solves on pure chains (bidiagonal factors:
as many levels as unknowns, one unknown per level) and on chains of narrow levels (k unknowns per level)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from ipx_amd import synth, kkt
rng = np.random.default_rng(3)
for width in (1, 8, 64, 512):
    levels = 4000
    m = levels * width
    n = 2 * m + 5
    # unknown i depends on 2 unknowns of the previous block of `width`
    rows, cols = [], []
    for d in range(2):
        i = np.arange(width, m)
        j = (i // width - 1) * width + rng.integers(0, width, i.size)
        rows.append(i); cols.append(j)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    Lm = sp.coo_matrix((rng.uniform(-0.4, 0.4, rows.size), (rows, cols)), shape=(m, m)).tocsc(); Lm.sum_duplicates(); Lm.sort_indices()
    Um = (sp.coo_matrix((rng.uniform(-0.4, 0.4, rows.size), (cols, rows)), shape=(m, m)) + sp.diags(rng.uniform(1, 2, m))).tocsc(); Um.sum_duplicates(); Um.sort_indices()
    mk = lambda M: synth.CscMatrix(m, m, M.indptr, M.indices, M.data)
    ctx = kkt.KktContext(synth.synthetic_lp(m, n, 2, 1))
    ident = np.arange(m, dtype=np.int64)
    status = np.full(n + m, -1, dtype=np.int64); status[:m] = 0
    ctx.split_prepare(mk(Lm), mk(Um), ident, ident, ident, status, np.ones(n + m))
    ctx.set_pointer_mode(True)
    lib = ctx.lib
    x = ctx.vector(m, rng.standard_normal(m))
    for name, fn in (("forward (L, U)", lib.ipxk_forward_solve), ("backward (U', L')", lib.ipxk_backward_solve)):
        fn(ctx.h, x.as_arg()); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn(ctx.h, x.as_arg())
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("width %4d, %d levels per sweep: %s pair %.1f us = %.3f us per level (two sweeps)" % (width, ctx.split_levels()[0], name, dt * 1e6, dt * 1e6 / (2 * levels)), flush=True)
    ctx.close()
