"""Row partition of the KKT system over the GPUs of one node.

Rank g owns the contiguous rows R_g of AI = [A I] for all n structural columns, the slices
y_g, b_g of every m-vector, and its own slack columns.  n-vectors (t = Ws.*(A'y), the
structural parts of a, x, W) are replicated.  One all-reduce of the n-vector t per NormalMatrix
apply, scalar all-gathers for the CR dot products / norms; nothing else crosses ranks.
Pure numpy index arithmetic (no GPU, no oracle): the same code feeds the HIP contexts in
bench.py and the gloo CPU tests.
"""
from collections import namedtuple

import numpy as np

from .synth import CscMatrix

i64 = np.int64
Slab = namedtuple("Slab", "A r0 r1 xl xu zl zu a b")


def row_range(m, rank, world):
    """Contiguous, balanced row ranges: the first m % world ranks get one extra row."""
    base, extra = divmod(m, world)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def slab_matrix(A, r0, r1):
    """A[r0:r1, :] as CSC with local row indices (entry order within a column is kept)."""
    if r0 == 0 and r1 == A.nrow:
        return A
    sel = (A.i >= r0) & (A.i < r1)
    col = np.repeat(np.arange(A.ncol, dtype=i64), np.diff(A.p))
    counts = np.bincount(col[sel], minlength=A.ncol).astype(i64)
    p = np.concatenate([[0], np.cumsum(counts)]).astype(i64)
    return CscMatrix(r1 - r0, A.ncol, p, A.i[sel] - r0, A.x[sel])


def local_vector(v, n, r0, r1):
    """[structural part (replicated) ; this rank's slack slice] of an (n+m)-vector."""
    if v is None:
        return None
    return np.concatenate([v[:n], v[n + r0:n + r1]])


def row_slab(A, st, rank, world):
    m, n = A.nrow, A.ncol
    r0, r1 = row_range(m, rank, world)
    loc = lambda key: local_vector(st[key], n, r0, r1)
    return Slab(slab_matrix(A, r0, r1), r0, r1, loc("xl"), loc("xu"), loc("zl"), loc("zu"), loc("a"),
                st["b"][r0:r1])


def assemble(n, parts_x, parts_y):
    """Inverse of the partition for results: x = [x_s ; x_I slices], y = concatenated slices.
    parts_x[g] has length n + m_g (structural part identical on every rank)."""
    x = np.concatenate([parts_x[0][:n]] + [p[n:] for p in parts_x])
    return x, np.concatenate(parts_y)
