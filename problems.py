"""Synthetic QP generators for BASELINE.json's configs.  Used by bench.py and tests/.

All return (P, q, A, l, u) with P the full symmetric matrix (the front-end keeps the upper triangle,
/root/reference/src/osqp/interface.py:221-222) and A in CSC, float64.
"""
import numpy as np
import scipy.sparse as sp


def _distinct_sorted(rng, rows, k, w):
    """rows x k distinct sorted integers in [0, w) per row (bijection multiset -> set)."""
    r = np.sort(rng.integers(0, w - k + 1, size=(rows, k)), axis=1)
    return r + np.arange(k)[None, :]


def banded_qp(n, m=None, nnz_per_row=5, window=200, eq_frac=0.1, seed=12345, long_range=0.0):
    """BASELINE configs[1] ("Single QP n=100k m=200k nnz(A)=1M nnz(P)=200k"), SURVEY §8d config 2:
    A has exactly `nnz_per_row` N(0,1) entries per row in a random-within-a-band pattern (columns drawn from a window
    of `window` columns around i*n/m), P = diag(U(0.1,1.1)) + one symmetric off-diagonal pair per two rows inside the
    band, the diagonal raised by the row's off-diagonal mass (strictly diagonally dominant: eigenvalues >= 0.1; stored full nnz(P) = 2n), 10 % equality rows; feasible by construction.
    long_range > 0 (`bench.py --config mixed`): that fraction of A's entries gets its column redrawn from the WHOLE range -- a band plus a few
    long-range couplings (drawn from a generator of their own: the banded part is the same matrix as with long_range = 0)."""
    rng = np.random.default_rng(seed)
    m = 2 * n if m is None else m
    k = min(nnz_per_row, n)
    w = min(window, n)
    centre = (np.arange(m, dtype=np.int64) * n) // max(m, 1)
    lo = np.clip(centre - w // 2, 0, n - w)
    cols = lo[:, None] + _distinct_sorted(rng, m, k, w)
    vals = rng.standard_normal((m, k))
    if long_range > 0:
        r2 = np.random.default_rng(seed + 1)
        pick = r2.random((m, k)) < long_range
        far = r2.integers(0, n, size=(m, k))
        cand = np.sort(np.where(pick, far, cols), axis=1)
        dup = (np.diff(cand, axis=1) == 0).any(axis=1)          # (a redrawn column that hits another entry of its row: that row keeps its band)
        cols = np.where(dup[:, None], cols, cand)
    A = sp.csr_matrix((vals.ravel(), cols.ravel().astype(np.int32), np.arange(0, m * k + 1, k, dtype=np.int32)), shape=(m, n)).tocsc()
    A.sort_indices()
    d = rng.uniform(0.1, 1.1, n)
    i = np.arange(0, n - 1, 2)
    # (partner column inside the band; near the last column the offset wraps inside what is left of the row -- clamping to n - 1 piled
    #  every late pair onto the last column: an arrow that made P indefinite for wide windows, found with tools/robustness_grid.py)
    j = i + 1 + rng.integers(0, max(w - 1, 1), size=i.size) % np.maximum(n - 1 - i, 1)
    keep = j > i
    i, j = i[keep], j[keep]
    off = rng.uniform(-0.04, 0.04, size=i.size)
    d = d + np.bincount(i, np.abs(off), n) + np.bincount(j, np.abs(off), n)      # strictly diagonally dominant: eigenvalues >= 0.1
    P = sp.coo_matrix((np.concatenate([d, off, off]), (np.concatenate([np.arange(n), i, j]), np.concatenate([np.arange(n), j, i]))), shape=(n, n)).tocsc()
    P.sort_indices()
    q = rng.standard_normal(n)
    x0 = 0.1 * rng.standard_normal(n)
    ax0 = A @ x0
    l = ax0 - rng.uniform(0, 1, m)
    u = ax0 + rng.uniform(0, 1, m)
    eq = rng.random(m) < eq_frac
    l[eq] = ax0[eq]; u[eq] = ax0[eq]
    return P, q, A, l, u


def random_qp(n=50, m=100, density=0.15, seed=0):
    """BASELINE configs[0] (same construction as tests/golden/make_fixtures.py 'config1_random_qp')."""
    rng = np.random.default_rng(seed)
    M = sp.random(n, n, density=density, random_state=rng, data_rvs=rng.standard_normal)
    P = (M @ M.T + 1e-2 * sp.eye(n)).tocsc(); q = rng.standard_normal(n)
    A = sp.random(m, n, density=density, random_state=rng, data_rvs=rng.standard_normal, format='csc')
    return P, q, A, -rng.random(m), rng.random(m)


def lasso_qp(nf=50, ns=100, density=1.0, seed=1):
    """BASELINE configs[2] Lasso-as-QP: vars (x in R^nf, y in R^ns, t in R^nf);
    min y'y + lam 1't  s.t.  y = Ad x - b,  -t <= x <= t."""
    rng = np.random.default_rng(seed)
    Ad = sp.random(ns, nf, density=density, random_state=rng, data_rvs=rng.standard_normal, format='csc') if density < 1.0 \
        else sp.csc_matrix(rng.standard_normal((ns, nf)))
    xh = rng.standard_normal(nf) * (rng.random(nf) < 0.5) / np.sqrt(nf)
    b = Ad @ xh + 0.5 * rng.standard_normal(ns)
    lam = np.abs(Ad.T @ b).max() / 5.0
    P = sp.block_diag([sp.csc_matrix((nf, nf)), 2.0 * sp.eye(ns), sp.csc_matrix((nf, nf))], format='csc')
    q = np.concatenate([np.zeros(nf + ns), lam * np.ones(nf)])
    In = sp.eye(nf)
    A = sp.vstack([sp.hstack([Ad, -sp.eye(ns), sp.csc_matrix((ns, nf))]),
                   sp.hstack([In, sp.csc_matrix((nf, ns)), -In]),
                   sp.hstack([In, sp.csc_matrix((nf, ns)), In])], format='csc')
    l = np.concatenate([b, -np.inf * np.ones(nf), np.zeros(nf)])
    u = np.concatenate([b, np.zeros(nf), np.inf * np.ones(nf)])
    return P, q, A, l, u


def portfolio_qp(na=200, k=10, seed=1, gamma=1.0):
    """BASELINE configs[3] portfolio factor model: vars (x in R^na, y in R^k);
    min x'Dx + y'y - mu'x/gamma  s.t.  y = F'x, 1'x = 1, 0 <= x <= 1."""
    rng = np.random.default_rng(seed)
    F = sp.random(na, k, density=0.5, random_state=rng, data_rvs=rng.standard_normal, format='csc')
    D = sp.diags(rng.random(na) * np.sqrt(k))
    mu = rng.standard_normal(na)
    P = sp.block_diag([2.0 * D, 2.0 * sp.eye(k)], format='csc')
    q = np.concatenate([-mu / gamma, np.zeros(k)])
    A = sp.vstack([sp.hstack([F.T, -sp.eye(k)]), sp.hstack([sp.csc_matrix(np.ones((1, na))), sp.csc_matrix((1, k))]),
                   sp.hstack([sp.eye(na), sp.csc_matrix((na, k))])], format='csc')
    l = np.concatenate([np.zeros(k), [1.0], np.zeros(na)])
    u = np.concatenate([np.zeros(k), [1.0], np.ones(na)])
    return P, q, A, l, u


def mpc_batch(nbatch, nx=8, nu=4, N=10, seed=7):
    """BASELINE configs[4]: `nbatch` MPC QPs sharing one sparsity pattern AND one (P, A): horizon N,
    nx states, nu inputs => n = N(nx+nu) = 120 variables z = (x_1..x_N, u_0..u_{N-1});  rows: dynamics N*nx (equalities)
    + variable box n + input-rate N*nu  => m = 240.  Only the bounds of the first nx dynamics rows (A_d x0) differ per problem
    (mirrors the update-style batching of /root/reference/src/osqp/nn/torch.py:136-140).
    Returns P, q, A and arrays L, U of shape (nbatch, m)."""
    rng = np.random.default_rng(seed)
    Ad = rng.standard_normal((nx, nx)); Ad *= 0.95 / np.abs(np.linalg.eigvals(Ad)).max()
    Bd = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    n = N * (nx + nu); xo = lambda k: (k - 1) * nx; uo = lambda k: N * nx + k * nu      # offsets of x_k (k=1..N), u_k (k=0..N-1)
    rows, cols, vals = [], [], []
    def put(r, c, v):
        rows.append(r); cols.append(c); vals.append(v)
    r = 0
    for k in range(N):                                     # x_{k+1} - Ad x_k - Bd u_k = (Ad x0 if k == 0 else 0)
        for i in range(nx):
            put(r + i, xo(k + 1) + i, 1.0)
            for j in range(nu):
                put(r + i, uo(k) + j, -Bd[i, j])
            if k > 0:
                for j in range(nx):
                    put(r + i, xo(k) + j, -Ad[i, j])
        r += nx
    for j in range(n):                                     # box on every variable
        put(r + j, j, 1.0)
    r += n
    for k in range(N):                                     # input rate u_k - u_{k-1}
        for j in range(nu):
            put(r + j, uo(k) + j, 1.0)
            if k > 0:
                put(r + j, uo(k - 1) + j, -1.0)
        r += nu
    m = r
    A = sp.csc_matrix((vals, (rows, cols)), shape=(m, n)); A.sort_indices()
    P = sp.diags(np.concatenate([np.ones(N * nx), 0.1 * np.ones(N * nu)]), format='csc')
    q = np.zeros(n)
    x0 = rng.standard_normal((nbatch, nx))
    L = np.zeros((nbatch, m)); U = np.zeros((nbatch, m))
    L[:, :nx] = U[:, :nx] = x0 @ Ad.T
    L[:, N * nx:N * nx + N * nx] = -5.0; U[:, N * nx:N * nx + N * nx] = 5.0           # |x| <= 5
    L[:, 2 * N * nx:N * nx + n] = -1.0; U[:, 2 * N * nx:N * nx + n] = 1.0              # |u| <= 1
    L[:, N * nx + n:] = -0.5; U[:, N * nx + n:] = 0.5                                  # |du| <= 0.5
    return P, q, A, L, U


def kkt_certificate(P, q, A, l, u, x, y):
    """Independent optimality certificate (unscaled): primal residual, dual residual, complementarity, objective."""
    ax = A @ x
    pri = max(np.maximum(ax - u, 0).max(initial=0.0), np.maximum(l - ax, 0).max(initial=0.0))
    dua = np.abs(P @ x + q + A.T @ y).max()
    fin_u, fin_l = u < 1e20, l > -1e20
    comp = max(np.abs(np.maximum(y, 0)[fin_u] * (u - ax)[fin_u]).max(initial=0.0),
               np.abs(np.minimum(y, 0)[fin_l] * (ax - l)[fin_l]).max(initial=0.0),
               np.maximum(y, 0)[~fin_u].max(initial=0.0), np.maximum(-y, 0)[~fin_l].max(initial=0.0))
    return dict(pri=pri, dua=dua, comp=comp, obj=0.5 * x @ (P @ x) + q @ x)
