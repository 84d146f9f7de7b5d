"""Seeded synthetic LPs, IPM states and planted-LU bases for tests and bench.py.

Shapes follow SURVEY.md section 8(d): every structural column has k distinct
uniformly drawn rows, values +-U[0.5,4) (the reference's power-of-2 equilibration
is then a no-op, reference src/presolver.cc:913-924, so solver matrix == input
matrix), all constraints '<' (slack in [0,inf)), lb=0, ub=inf.  Pure numpy; no
reference or oracle code is involved.
"""
import numpy as np

i64 = np.int64
f64 = np.float64


class CscMatrix:
    """m x n CSC matrix, int64 indices, sorted row indices within columns."""

    def __init__(self, nrow, ncol, colptr, rowidx, values):
        self.nrow, self.ncol = int(nrow), int(ncol)
        self.p = np.ascontiguousarray(colptr, dtype=i64)
        self.i = np.ascontiguousarray(rowidx, dtype=i64)
        self.x = np.ascontiguousarray(values, dtype=f64)

    @property
    def nnz(self):
        return int(self.p[-1])

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.x, self.i, self.p), shape=(self.nrow, self.ncol))

    def with_identity(self):
        """[A I] as the reference's Model::AI() (src/model.h:61)."""
        m, n = self.nrow, self.ncol
        p = np.concatenate([self.p, self.nnz + 1 + np.arange(m, dtype=i64)])
        i = np.concatenate([self.i, np.arange(m, dtype=i64)])
        x = np.concatenate([self.x, np.ones(m, dtype=f64)])
        return CscMatrix(m, n + m, p, i, x)


def _distinct_rows(rng, m, n, k):
    rows = np.sort(rng.integers(0, m, size=(n, k), dtype=i64), axis=1)
    while True:
        bad = np.nonzero((rows[:, 1:] == rows[:, :-1]).any(axis=1))[0]
        if bad.size == 0:
            return rows
        rows[bad] = np.sort(rng.integers(0, m, size=(bad.size, k), dtype=i64), axis=1)


def synthetic_lp(m, n, k=8, seed=12345, num_dense=0):
    """Returns the structural matrix A (m x n CSC).  With num_dense > 0 the first
    num_dense columns hold ~m/2 random rows each (drawn with replacement, then
    de-duplicated) -- the dense-column stress case C5."""
    rng = np.random.default_rng(seed)
    k = min(k, m)
    rows = _distinct_rows(rng, m, n, k)
    vals = rng.uniform(0.5, 4.0, size=(n, k)) * rng.choice([-1.0, 1.0], size=(n, k))
    if num_dense == 0:
        colptr = np.arange(n + 1, dtype=i64) * k
        return CscMatrix(m, n, colptr, rows.reshape(-1), vals.reshape(-1))
    cols_i, cols_x = [], []
    for j in range(num_dense):
        r = np.unique(rng.integers(0, m, size=max(m // 2, 1), dtype=i64))
        cols_i.append(r)
        cols_x.append(rng.uniform(0.5, 4.0, size=r.size) * rng.choice([-1.0, 1.0], size=r.size))
    counts = np.concatenate([[c.size for c in cols_i], np.full(n - num_dense, k)]).astype(i64)
    colptr = np.concatenate([[0], np.cumsum(counts)]).astype(i64)
    rowidx = np.concatenate(cols_i + [rows[num_dense:].reshape(-1)])
    values = np.concatenate(cols_x + [vals[num_dense:].reshape(-1)])
    return CscMatrix(m, n, colptr, rowidx, values)


def banded_lp(m, n, k=8, bandwidth=4096, seed=12345):
    """Structured counterpart of synthetic_lp: the k rows of column j are drawn from a band of
    `bandwidth` rows around j*m/n, so both gathers of A*D*A' have locality (what row/column
    orderings of real LPs provide).  Same value distribution."""
    rng = np.random.default_rng(seed)
    centre = (np.arange(n, dtype=i64) * m) // n
    lo = np.clip(centre - bandwidth // 2, 0, max(m - bandwidth, 0))
    bw = min(bandwidth, m)
    rows = np.sort(lo[:, None] + rng.integers(0, bw, size=(n, k), dtype=i64), axis=1)
    while True:
        bad = np.nonzero((rows[:, 1:] == rows[:, :-1]).any(axis=1))[0]
        if bad.size == 0:
            break
        rows[bad] = np.sort(lo[bad, None] + rng.integers(0, bw, size=(bad.size, k), dtype=i64), axis=1)
    vals = rng.uniform(0.5, 4.0, size=(n, k)) * rng.choice([-1.0, 1.0], size=(n, k))
    return CscMatrix(m, n, np.arange(n + 1, dtype=i64) * k, rows.reshape(-1), vals.reshape(-1))


def shuffled(A, seed=12345):
    """A with rows and columns in random order: (P A Q, rowperm, colperm) with row i of the result = row rowperm[i] of A.
    What banded_lp looks like after a modelling tool has emitted its rows and columns in no particular order."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rp, cp = rng.permutation(A.nrow).astype(i64), rng.permutation(A.ncol).astype(i64)
    M = sp.csc_matrix((A.x, A.i, A.p), shape=(A.nrow, A.ncol))[rp][:, cp].tocsc()
    M.sort_indices()
    return CscMatrix(A.nrow, A.ncol, M.indptr.astype(i64), M.indices.astype(i64), M.data.astype(f64)), rp, cp


def synthetic_ipm_state(m, n, spread=1.0, seed=12345):
    """xl, zl = 10^(spread*U[-1,1]) independently, xu=inf, zu=0 (W_j = xl_j/zl_j spans
    4*spread decades); a, b ~ U[-0.5,0.5).  Returns dict with xl,xu,zl,zu,mu,a,b."""
    rng = np.random.default_rng(seed + 1)
    N = n + m
    xl = 10.0 ** (spread * rng.uniform(-1.0, 1.0, N))
    zl = 10.0 ** (spread * rng.uniform(-1.0, 1.0, N))
    xu = np.full(N, np.inf)
    zu = np.zeros(N)
    mu = float(np.dot(xl, zl) / N)  # complementarity measure, all N barrier terms
    a = rng.uniform(-0.5, 0.5, N)
    b = rng.uniform(-0.5, 0.5, m)
    return dict(xl=xl, xu=xu, zl=zl, zu=zu, mu=mu, a=a, b=b)


def synthetic_basis_state(status, spread=1.0, seed=12345):
    """Column scaling factors (Iterate::ScalingFactor, reference src/iterate.cc:183-198) for the
    basis path in the regime the basis preconditioner is built for: basic variables are the
    ones with large scaling factors (what maxvolume arranges), nonbasic ones small.
    BASIC_FREE -> inf, NONBASIC_FIXED -> 0 as in the reference."""
    rng = np.random.default_rng(seed + 3)
    N = status.size
    u = rng.uniform(0.0, 1.0, N)
    d = np.where(status >= 0, 10.0 ** (spread * u), 0.3 * 10.0 ** (-spread * u))
    d[status == 1] = np.inf
    d[status == -2] = 0.0
    return d


def lp_vectors(m, n):
    """obj=1, lb=0, ub=inf, rhs=1, constr_type='<' (SURVEY 8d)."""
    return dict(obj=np.ones(n), lb=np.zeros(n), ub=np.full(n, np.inf), rhs=np.ones(m),
                constr_type="<" * m)


def planted_lu_basis(A, offdiag=3, seed=12345, band=None, num_free=0, num_fixed=0, big_rows=0.0, big=1.0):
    """Plants a basis with known LU factors into [A I] (BASICLU is not available
    offline).  Generates sparse unit-lower L0 (strictly lower part
    returned) and upper U0 with `offdiag` off-diagonal entries per column (uniform
    rows, or confined to a `band`), forms B = (L0+I) U0 and REPLACES the first m
    structural columns of A by B's columns (scrambled by random row/column
    permutations) so that variables basis[p] = p are basic.

    Returns dict(A=new structural matrix, L, U (CscMatrix, diagonal of U last in each
    column), rowperm, colperm, basis, status) with the reference's factor contract
    B[rowperm,colperm] = (L+I)U (reference src/lu_update.h:43-60).
    """
    import scipy.sparse as sp
    rng = np.random.default_rng(seed + 2)
    m, n = A.nrow, A.ncol
    assert n >= m

    def strict_tri(lower):
        cols = np.repeat(np.arange(m, dtype=i64), offdiag)
        if lower:   # rows > col
            span = (m - 1 - cols) if band is None else np.minimum(m - 1 - cols, band)
            rows = cols + 1 + (rng.random(cols.size) * span).astype(i64)
            ok = span > 0
        else:       # rows < col
            span = cols if band is None else np.minimum(cols, band)
            rows = cols - 1 - (rng.random(cols.size) * span).astype(i64)
            ok = span > 0
        rows, cols = rows[ok], cols[ok]
        vals = rng.uniform(0.1, 0.6, rows.size) * rng.choice([-1.0, 1.0], rows.size)
        if lower and big_rows > 0:
            hot = rows >= int((1.0 - big_rows) * m)
            vals[hot] = rng.uniform(big, 1.2 * big, int(hot.sum())) * rng.choice([-1.0, 1.0], int(hot.sum()))
        T = sp.coo_matrix((vals, (rows, cols)), shape=(m, m)).tocsc()
        T.sum_duplicates()
        T.sort_indices()
        return T

    L0 = strict_tri(True)
    Us = strict_tri(False)
    udiag = rng.uniform(0.5, 2.0, m) * rng.choice([-1.0, 1.0], m)
    U0 = (Us + sp.diags(udiag)).tocsc()
    U0.sort_indices()  # diagonal is the largest row index in an upper column -> last
    Bp = ((L0 + sp.identity(m, format="csc")) @ U0).tocsc()
    rowperm = rng.permutation(m).astype(i64)
    colperm = rng.permutation(m).astype(i64)
    # B[rowperm[i], colperm[k]] = Bp[i, k]
    Bcoo = Bp.tocoo()
    B = sp.coo_matrix((Bcoo.data, (rowperm[Bcoo.row], colperm[Bcoo.col])), shape=(m, m)).tocsc()
    B.sum_duplicates()
    B.sort_indices()
    # scale B's columns into the value range of A so that the equilibration stays a no-op
    rest = A.to_scipy()[:, m:]
    Anew = sp.hstack([B, rest]).tocsc()
    Anew.sort_indices()
    status = np.full(n + m, -1, dtype=i64)          # NONBASIC
    status[:m] = 0                                   # BASIC
    basis = np.arange(m, dtype=i64)
    if num_free:
        status[rng.choice(m, num_free, replace=False)] = 1     # BASIC_FREE
    if num_fixed:
        status[m + rng.choice(n, num_fixed, replace=False)] = -2  # NONBASIC_FIXED
    mk = lambda M: CscMatrix(M.shape[0], M.shape[1], M.indptr, M.indices, M.data)
    return dict(A=mk(Anew), L=mk(L0), U=mk(U0), rowperm=rowperm, colperm=colperm,
                basis=basis, status=status)


def synthetic_newton_state(m, n, seed, num_free=0, num_fixed=0, num_ub=0, num_boxed=0):
    """Iterate vectors, variable states and residuals for IPM::SolveNewtonSystem (src/ipm.cc:532-645).
    States follow Iterate's invariants (src/iterate.h:220-262): barrier-lb xu = inf, zu = 0; barrier-ub
    xl = inf, zl = 0; boxed all finite; free xl = xu = inf, zl = zu = 0; fixed all zero.
    Codes: 0 fixed, 1 free, 2 lb, 3 ub, 4 boxed.  Slack variables are barrier-lb."""
    rng = np.random.default_rng(seed)
    N = n + m
    state = np.full(N, 2, dtype=np.uint8)
    special = rng.permutation(n)[:num_free + num_fixed + num_ub + num_boxed]
    state[special[:num_free]] = 1
    state[special[num_free:num_free + num_fixed]] = 0
    state[special[num_free + num_fixed:num_free + num_fixed + num_ub]] = 3
    state[special[num_free + num_fixed + num_ub:]] = 4
    pos = lambda: 10.0 ** rng.uniform(-1, 1, N)
    xl, xu, zl, zu = pos(), pos(), pos(), pos()
    lb, ub = (state == 2) | (state == 4), (state == 3) | (state == 4)
    xl[~lb] = np.inf; zl[~lb] = 0.0
    xu[~ub] = np.inf; zu[~ub] = 0.0
    fx = state == 0
    xl[fx] = xu[fx] = zl[fx] = zu[fx] = 0.0
    mu = float((xl[lb] * zl[lb]).sum() + (xu[ub] * zu[ub]).sum()) / (lb.sum() + ub.sum())
    U = lambda k: rng.uniform(-0.5, 0.5, k)
    rl = np.where(lb, U(N), 0.0)
    ru = np.where(ub, U(N), 0.0)
    sl = np.where(lb, mu - xl_safe(xl) * zl, 0.0)
    su = np.where(ub, mu - xl_safe(xu) * zu, 0.0)
    return dict(state=state, xl=xl, xu=xu, zl=zl, zu=zu, mu=mu, rb=U(m), rc=U(N), rl=rl, ru=ru, sl=sl, su=su)


def xl_safe(v):
    """inf -> 0 so that products with zero multipliers stay finite"""
    return np.where(np.isfinite(v), v, 0.0)


def synthetic_iterate(m, n, seed, frac_special=0.15):
    """An LP with mixed bounds / constraint types plus an interior iterate and a step for it, consistent
    with Iterate::Initialize and assert_consistency (reference src/iterate.cc:60-92, 450-520).
    Returns dict(A, rhs, constr_type, obj, lb, ub  [user model];  lbs, ubs [bounds incl. slacks];
    state [IPXK_STATE_* per variable of AI];  it = dict(x, xl, xu, y, zl, zu);
    step = dict(dx, dxl, dxu, dy, dzl, dzu))."""
    rng = np.random.default_rng(seed)
    A = synthetic_lp(m, n, 8, seed)
    N = n + m
    lb, ub = np.zeros(n), np.full(n, np.inf)
    k = int(frac_special * n)
    sp = rng.permutation(n)[:3 * k]
    lb[sp[:k]] = -np.inf; ub[sp[:k]] = rng.uniform(1, 3, k)                 # upper bound only
    lb[sp[k:2 * k]] = rng.uniform(-2, 0, k); ub[sp[k:2 * k]] = rng.uniform(1, 3, k)   # boxed
    lb[sp[2 * k:]] = -np.inf                                              # free
    ct = rng.choice(list("<>="), size=m, p=[0.6, 0.25, 0.15])
    slb = np.where(ct == ">", -np.inf, 0.0)
    sub = np.where(ct == "<", np.inf, 0.0)
    lbs, ubs = np.concatenate([lb, slb]), np.concatenate([ub, sub])
    fl, fu = np.isfinite(lbs), np.isfinite(ubs)
    state = np.where(fl & fu, 4, np.where(fl, 2, np.where(fu, 3, 1))).astype(np.uint8)
    pos = lambda: 10.0 ** rng.uniform(-1, 1, N)
    it = dict(x=rng.uniform(-1, 1, N), y=rng.uniform(-1, 1, m),
              xl=np.where(fl, pos(), np.inf), zl=np.where(fl, pos(), 0.0),
              xu=np.where(fu, pos(), np.inf), zu=np.where(fu, pos(), 0.0))
    U = lambda q: rng.uniform(-1, 1, q)
    step = dict(dx=U(N), dxl=U(N), dxu=U(N), dy=U(m), dzl=U(N), dzu=U(N))
    return dict(A=A, rhs=rng.uniform(-1, 1, m), constr_type="".join(ct), obj=rng.uniform(-1, 1, n), lb=lb, ub=ub,
                lbs=lbs, ubs=ubs, state=state, it=it, step=step)


def lp_like_basis_matrix(dim, bump=64, frac_rowsing=0.2, frac_slack=0.5, offdiag=2, window=None,
                         bump_density=0.3, num_dependent=0, seed=12345, shallow=False):
    """A nearly triangular matrix of the kind LP bases are (the input of Basis::Factorize,
    reference src/basis.cc:116-156): in a hidden pivot order it is
        [ T   X   X ]   T: upper triangular, `frac_slack` of its columns unit columns, the others with up to
        [ 0   D   0 ]      `offdiag` entries above the diagonal (uniform rows, or the `window` rows above);
        [ 0   Y   R ]   R: LOWER triangular (its rows are row singletons once T is gone), entries below the diagonal;
                        D: the bump (bump x bump, `bump_density`, every row and column >= 2 entries);
    X, Y sparse.  `num_dependent` bump columns are made copies of other bump columns (a singular basis).
    shallow: the unit columns of T come first and the other columns of T only reach into THEIR rows (a slack-heavy
    basis: every structural column depends on slack rows only, so an exchanged column disturbs the triangular
    order locally instead of blocking everything downstream of its row).
    Rows and columns are scrambled.  Returns dict(Bp, Bi, Bx (CSC, unsorted), dim, plus the planted sizes)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed + 7)
    n3 = int(frac_rowsing * (dim - bump))
    n1 = dim - bump - n3
    r, c, v = [], [], []

    def vals(k):
        return rng.uniform(0.5, 4.0, k) * rng.choice([-1.0, 1.0], k)

    def add(rows, cols):
        r.append(rows.astype(i64)); c.append(cols.astype(i64)); v.append(vals(rows.size))

    # T: diagonal + entries above it in the non-slack columns
    add(np.arange(n1), np.arange(n1))
    is_slack = rng.random(n1) < frac_slack
    if shallow:
        is_slack = np.arange(n1) < int(frac_slack * n1)
    v[-1][is_slack] = 1.0                    # unit columns: slack columns of [A I]
    nonslack = np.nonzero(~is_slack)[0]
    cols = np.repeat(nonslack, offdiag)
    span = cols if window is None else np.minimum(cols, window)
    if shallow:
        span = np.full(cols.size, int(frac_slack * n1))
    ok = span > 0
    rows = cols - 1 - (rng.random(cols.size) * span).astype(i64)
    if shallow:
        rows = (rng.random(cols.size) * span).astype(i64)
    add(rows[ok], cols[ok])
    # X: bump and R columns reach into the rows of T
    if n1 > 0:
        cols = np.repeat(np.arange(n1, dim), offdiag)
        add((rng.random(cols.size) * n1).astype(i64), cols)
    # D
    if bump > 0:
        D = (rng.random((bump, bump)) < bump_density)
        D[np.arange(bump), np.arange(bump)] = True
        D[np.arange(bump), (np.arange(bump) + 1) % bump] = True if bump > 1 else D[0, 0]
        rr, cc = np.nonzero(D)
        add(n1 + rr, n1 + cc)
    # R: diagonal + entries below it; Y: R's rows reach into the bump columns
    if n3 > 0:
        base = n1 + bump
        add(base + np.arange(n3), base + np.arange(n3))
        cols = np.repeat(np.arange(n3), offdiag)
        span = n3 - 1 - cols
        ok = span > 0
        rows = cols + 1 + (rng.random(cols.size) * span).astype(i64)
        add(base + rows[ok], base + cols[ok])
        if bump > 0:
            rows = np.repeat(np.arange(n3), 1)
            add(base + rows, n1 + (rng.random(rows.size) * bump).astype(i64))
    R, Cc, V = np.concatenate(r), np.concatenate(c), np.concatenate(v)
    M = sp.coo_matrix((V, (R, Cc)), shape=(dim, dim)).tocsc()
    M.sum_duplicates()                       # duplicates add up (still nonzero with probability 1)
    if num_dependent and bump > 2:           # copies of bump columns: a numerically singular bump
        M = M.tolil()
        for t in range(num_dependent):
            M[:, n1 + bump - 1 - t] = M[:, n1 + t] * 2.0
        M = M.tocsc()
    rowperm, colperm = rng.permutation(dim), rng.permutation(dim)
    Mc = M.tocoo()
    B = sp.coo_matrix((Mc.data, (rowperm[Mc.row], colperm[Mc.col])), shape=(dim, dim)).tocsc()
    slack_cols = colperm[np.nonzero(is_slack)[0]]          # columns of B that are unit columns e_i ...
    slack_rows = rowperm[np.nonzero(is_slack)[0]]          # ... and their rows i
    return dict(dim=dim, Bp=B.indptr.astype(i64), Bi=B.indices.astype(i64), Bx=B.data.astype(f64),
                planted=dict(T=n1, bump=bump, R=n3), slack_cols=slack_cols, slack_rows=slack_rows,
                rowperm=rowperm.astype(i64), colperm=colperm.astype(i64), is_slack_stage=is_slack)


def disturbed_basis_matrix(dim, num_exchanged, entries=8, seed=12345, **kw):
    """lp_like_basis_matrix after `num_exchanged` basis exchanges: that many of its columns are replaced by random
    columns of `entries` entries (what Maxvolume / the simplex method do to a basis).  Singleton peeling is
    all-or-nothing along dependency chains, so a few dozen such columns leave a bump of a large part of the matrix
     -- sparse and nearly triangular, the input of the LU's tearing phase."""
    G = lp_like_basis_matrix(dim, seed=seed, **kw)
    rng = np.random.default_rng(seed + 99)
    import scipy.sparse as sp
    B = sp.csc_matrix((G["Bx"], G["Bi"], G["Bp"]), shape=(dim, dim)).tolil()
    cols = rng.choice(dim, num_exchanged, replace=False)
    stage_of_col = np.empty(dim, dtype=i64)
    stage_of_col[G["colperm"]] = np.arange(dim)
    for j in cols:
        # an exchange keeps the basis nonsingular (the pivot element of the update is nonzero): the entering column
        # has an entry in the row the leaving column was paired with in the hidden pivot order
        keep = G["rowperm"][stage_of_col[j]]
        rows = np.unique(np.concatenate([[keep], rng.choice(dim, entries - 1, replace=False)]))
        B[:, j] = 0
        for r in rows:
            B[r, j] = rng.uniform(0.5, 4.0) * rng.choice([-1.0, 1.0])
    B = B.tocsc()
    B.eliminate_zeros()
    B.sort_indices()
    out = dict(G)
    out.update(Bp=B.indptr.astype(i64), Bi=B.indices.astype(i64), Bx=B.data.astype(f64), exchanged=np.sort(cols))
    return out


def lp_like_basis(m, n, seed=12345, **kw):
    """An LP [A I] (m x n structural columns) with a nearly triangular basis: lp_like_basis_matrix planted into it.
    The unit columns of B are slack columns n+i, its other columns are structural columns of A (scattered),
    the remaining structural columns are random (8 entries).  Returns dict(A, basis, status, G) with basis[k] the
    variable in column k of B and status as Basis::BasicStatus (BASIC 0 / NONBASIC -1)."""
    import scipy.sparse as sp
    G = lp_like_basis_matrix(m, seed=seed, **kw)
    rng = np.random.default_rng(seed + 11)
    B = sp.csc_matrix((G["Bx"], G["Bi"], G["Bp"]), shape=(m, m))
    is_slack = np.zeros(m, dtype=bool)
    is_slack[G["slack_cols"]] = True
    struct_cols = np.nonzero(~is_slack)[0]
    nb = struct_cols.size
    assert nb <= n
    where = np.sort(rng.choice(n, nb, replace=False))       # positions of B's structural columns inside A
    rest = synthetic_lp(m, n - nb, 8, seed + 1).to_scipy() if n > nb else None
    cols = [None] * n
    A = sp.lil_matrix((m, n))
    Bs = B[:, struct_cols].tocsc()
    other = np.setdiff1d(np.arange(n), where)
    blocks = sp.hstack([Bs] + ([rest] if rest is not None else [])).tocsc()
    order = np.empty(n, dtype=i64)
    order[where] = np.arange(nb)
    order[other] = nb + np.arange(n - nb)
    Anew = blocks[:, order].tocsc()
    Anew.sort_indices()
    basis = np.empty(m, dtype=i64)
    basis[struct_cols] = where
    basis[G["slack_cols"]] = n + G["slack_rows"]
    status = np.full(n + m, -1, dtype=i64)
    status[basis] = 0
    return dict(A=CscMatrix(m, n, Anew.indptr, Anew.indices, Anew.data), basis=basis, status=status, G=G)



def synthetic_maxvolume_state(status, spread=1.0, seed=12345):
    """Scaling factors that do NOT favour the current basis (log-uniform over 2*spread decades for every variable):
    the situation Maxvolume::RunHeuristic repairs (reference src/kkt_solver_basis.cc:46-50).  BASIC_FREE -> inf,
    NONBASIC_FIXED -> 0 as Iterate::ScalingFactor gives them."""
    rng = np.random.default_rng(seed + 5)
    d = 10.0 ** (spread * rng.uniform(-1.0, 1.0, status.size))
    d[status == 1] = np.inf
    d[status == -2] = 0.0
    return d


def basis_matrix_of(A, basis):
    """B = [A I][:, basis] as dict(dim, Bp, Bi, Bx) (columns in storage order)"""
    m, n = A.nrow, A.ncol
    basis = np.asarray(basis, dtype=np.int64)
    struct = basis < n
    lens = np.where(struct, A.p[np.minimum(basis, n - 1) + 1] - A.p[np.minimum(basis, n - 1)], 1)
    Bp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    Bi = np.empty(Bp[-1], dtype=np.int64)
    Bx = np.empty(Bp[-1], dtype=np.float64)
    # slack columns
    sl = np.nonzero(~struct)[0]
    Bi[Bp[sl]] = basis[sl] - n
    Bx[Bp[sl]] = 1.0
    st = np.nonzero(struct)[0]
    if st.size:
        src0 = A.p[basis[st]]
        ln = lens[st]
        off = np.arange(ln.sum()) - np.repeat(np.cumsum(ln) - ln, ln)
        src = np.repeat(src0, ln) + off
        dst = np.repeat(Bp[st], ln) + off
        Bi[dst] = A.i[src]
        Bx[dst] = A.x[src]
    return dict(dim=m, Bp=Bp, Bi=Bi, Bx=Bx)


def misplaced_basis(m, num_misplaced, seed=12345, **kw):
    """lp_like_basis(m, 2m) after `num_misplaced` basis exchanges at random positions (a random nonbasic structural column
    each): what a crossover-free IPM's basis looks like after Maxvolume has moved it -- the singleton rounds stall on a
    nucleus of tens of thousands of columns.  Returns dict(A, basis, G) with G = the basis matrix."""
    kw.setdefault("offdiag", 3)
    P = lp_like_basis(m, 2 * m, seed=seed, **kw)
    A, basis, status = P["A"], P["basis"].copy(), P["status"]
    rng = np.random.default_rng(seed + 5)
    enter = rng.choice(np.nonzero(status[:2 * m] == -1)[0], num_misplaced, replace=False)
    basis[rng.choice(m, num_misplaced, replace=False)] = enter
    return dict(A=A, basis=basis, G=basis_matrix_of(A, basis))


def misplaced_basis_matrix(m, num_misplaced, seed=12345, **kw):
    return misplaced_basis(m, num_misplaced, seed, **kw)["G"]


def synthetic_misplaced_state(status, num_misplaced, spread=1.0, seed=12345):
    """Scaling factors of an IPM iterate whose basis is nearly the maximum volume one: synthetic_basis_state (basic
    variables large, nonbasic ones small), except that `num_misplaced` nonbasic variables have grown large and as
    many basic ones small -- the few exchanges per IPM iteration that Maxvolume::RunHeuristic performs late in a
    solve (reference src/kkt_solver_basis.cc:46-50)."""
    rng = np.random.default_rng(seed + 9)
    d = synthetic_basis_state(status, spread, seed)
    nb = rng.choice(np.nonzero(status == -1)[0], num_misplaced, replace=False)
    bs = rng.choice(np.nonzero(status == 0)[0], num_misplaced, replace=False)
    d[nb] = 10.0 ** (spread * (1.0 + rng.uniform(0.0, 1.0, num_misplaced)))
    d[bs] = 0.3 * 10.0 ** (-spread * rng.uniform(0.0, 1.0, num_misplaced))
    return d


def synthetic_slack_entering_state(P, num_entering, spread=1.0, seed=12345, gap=2.0):
    """Scaling factors for the exchanges typical late in an interior point solve: `num_entering` constraints have
    become inactive, i.e. their (nonbasic) slack variables have grown large while the structural variable that
    is pivoted on that row in the planted triangular order has become small.  Maxvolume then brings those slacks
    into the basis; a unit column entering on its own row leaves the triangular order of the basis intact, so
    refactorizations stay within the reach of a singleton-based LU.  All other variables are well separated
    (`gap` decades between basic and nonbasic scaling factors), as they are when the IPM has nearly converged.
    P: result of lp_like_basis."""
    rng = np.random.default_rng(seed + 13)
    status, basis, G = P["status"], P["basis"], P["G"]
    n = P["A"].ncol
    d = synthetic_basis_state(status, spread, seed)
    d = np.where(status >= 0, d * 10.0 ** gap, d * 10.0 ** (-gap))
    n1 = G["planted"]["T"]
    stages = np.nonzero(~G["is_slack_stage"])[0]                 # triangular stages pivoted by a structural column
    pick = rng.choice(stages, num_entering, replace=False)
    rows = G["rowperm"][pick]                                    # their pivot rows ...
    leaving = basis[G["colperm"][pick]]                          # ... and the structural variables pivoted there
    assert np.all(status[n + rows] == -1) and np.all(leaving < n)
    d[n + rows] = 10.0 ** (gap + spread * (1.0 + rng.uniform(0.0, 1.0, num_entering)))
    d[leaving] = 0.3 * 10.0 ** (-gap - spread * rng.uniform(0.0, 1.0, num_entering))
    return d


def slack_basis_crash_state(m, n, num_entering, spread=1.0, seed=12345):
    """The situation of the first Maxvolume call of a solve: the basis is the slack basis (reference
    src/lp_solver.cc builds its starting basis from the slack basis by Maxvolume) and `num_entering` structural
    variables carry large scaling factors, all others small ones.  With num_entering^2 << m the entering columns
    hardly share rows, so the basis stays triangular by singletons.  Returns (basis, status, colscale)."""
    rng = np.random.default_rng(seed + 17)
    status = np.concatenate([np.full(n, -1, dtype=i64), np.zeros(m, dtype=i64)])
    basis = n + np.arange(m, dtype=i64)
    d = np.concatenate([0.3 * 10.0 ** (-spread * rng.uniform(0.0, 1.0, n)), 10.0 ** (spread * rng.uniform(0.0, 1.0, m))])
    d[rng.choice(n, num_entering, replace=False)] = 10.0 ** (2.0 + spread * rng.uniform(0.0, 1.0, num_entering))
    return basis, status, d
