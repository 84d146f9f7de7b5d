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


# ---- alternative: column partition --------------------------------------------------------
# Rank g owns the structural columns C_g (contiguous) of A for all m rows; every m-vector
# (y, b, the slack parts of a, x, W and all CR work vectors) is replicated.  t_g = W_g.*(A_g'y)
# is local; the one exchange per NormalMatrix apply is the all-reduce of the m partial sums
# A_g t_g, after which all ranks hold identical vectors and compute identical scalars.
ColSlab = namedtuple("ColSlab", "A c0 c1 xl xu zl zu a b")


def col_slab_matrix(A, c0, c1):
    """A[:, c0:c1] (columns are contiguous in CSC)."""
    p0, p1 = int(A.p[c0]), int(A.p[c1])
    return CscMatrix(A.nrow, c1 - c0, (A.p[c0:c1 + 1] - A.p[c0]).astype(i64), A.i[p0:p1], A.x[p0:p1])


def col_local_vector(v, n, c0, c1):
    """[this rank's structural slice ; all m slack entries] of an (n+m)-vector."""
    if v is None:
        return None
    return np.concatenate([v[c0:c1], v[n:]])


def col_slab(A, st, rank, world):
    n = A.ncol
    c0, c1 = row_range(n, rank, world)
    loc = lambda key: col_local_vector(st[key], n, c0, c1)
    return ColSlab(col_slab_matrix(A, c0, c1), c0, c1, loc("xl"), loc("xu"), loc("zl"), loc("zu"), loc("a"),
                   st["b"])


def assemble_cols(m, parts_x):
    """x = [structural slices in rank order ; slack part (identical on every rank)]."""
    return np.concatenate([p[:len(p) - m] for p in parts_x] + [parts_x[0][len(parts_x[0]) - m:]])


def assemble(n, parts_x, parts_y):
    """Inverse of the partition for results: x = [x_s ; x_I slices], y = concatenated slices.
    parts_x[g] has length n + m_g (structural part identical on every rank)."""
    x = np.concatenate([parts_x[0][:n]] + [p[n:] for p in parts_x])
    return x, np.concatenate(parts_y)
