"""Synthetic problem families of BASELINE.json `configs`.

Pure numpy/scipy data generators -- no solver code.  Each returns a dict with
P (upper-triangular CSC), q, A (CSC), l, u.
"""
import numpy as np
from scipy import sparse


def demo_qp():
    """n=2, m=3 demo (reference examples/osqp_demo.c:6-18)."""
    P = sparse.triu(sparse.csc_matrix([[4., 1.], [1., 2.]]), format="csc")
    A = sparse.csc_matrix([[1., 1.], [1., 0.], [0., 1.]])
    return dict(P=P, q=np.ones(2), A=A, l=np.array([1., 0., 0.]), u=np.array([1., .7, .7]))


def random_sparse_qp(n=10000, m=20000, nnz_per_col=20, p_upper_per_col=5, seed=1):
    """C2: random sparse QP, ~0.1 % dense A, diagonally dominant P (PSD)."""
    rng = np.random.default_rng(seed)
    # A: exactly nnz_per_col distinct rows per column
    k = min(nnz_per_col, m)
    rows = np.empty((n, k), dtype=np.int64)
    for j in range(n):
        rows[j] = rng.choice(m, size=k, replace=False)
    rows.sort(axis=1)
    indptr = np.arange(0, n * k + 1, k, dtype=np.int64)
    vals = rng.uniform(-1, 1, size=n * k)
    A = sparse.csc_matrix((vals, rows.ravel(), indptr), shape=(m, n))
    # P: ~p_upper_per_col strictly-upper entries per column + dominant diagonal
    ci, ri, vv = [], [], []
    for j in range(1, n):
        kk = min(p_upper_per_col, j)
        r = rng.choice(j, size=kk, replace=False)
        ci.append(np.full(kk, j)); ri.append(r); vv.append(rng.uniform(-.1, .1, size=kk))
    ci = np.concatenate(ci); ri = np.concatenate(ri); vv = np.concatenate(vv)
    offsum = np.zeros(n)
    np.add.at(offsum, ci, np.abs(vv))
    np.add.at(offsum, ri, np.abs(vv))
    diag = 0.1 + offsum + rng.uniform(0, 1, size=n)
    P = sparse.coo_matrix((np.concatenate([vv, diag]),
                           (np.concatenate([ri, np.arange(n)]), np.concatenate([ci, np.arange(n)]))),
                          shape=(n, n)).tocsc()
    P.sort_indices()
    q = rng.uniform(-1, 1, size=n)
    l = -rng.uniform(0, 1, size=m)
    u = rng.uniform(0, 1, size=m)
    return dict(P=P, q=q, A=A, l=l, u=u)


def lasso_qp(n_feat=5000, m_data=10000, density=0.15, gamma=1.0, seed=1):
    """C3: Lasso as a QP (reference docs/examples/lasso.rst:41-63).
    Variables (x, y, t): n = 2*n_feat + m_data, m = m_data + 2*n_feat."""
    rng = np.random.default_rng(seed)
    Ad = sparse.random(m_data, n_feat, density=density, format="csc", random_state=rng,
                       data_rvs=rng.standard_normal)
    x_true = (rng.random(n_feat) > 0.8) * rng.standard_normal(n_feat) / np.sqrt(n_feat)
    b = Ad @ x_true + 0.5 * rng.standard_normal(m_data)
    Im = sparse.eye(m_data, format="csc")
    In = sparse.eye(n_feat, format="csc")
    P = sparse.block_diag([sparse.csc_matrix((n_feat, n_feat)), Im,
                           sparse.csc_matrix((n_feat, n_feat))], format="csc")
    q = np.concatenate([np.zeros(n_feat + m_data), gamma * np.ones(n_feat)])
    A = sparse.vstack([sparse.hstack([Ad, -Im, sparse.csc_matrix((m_data, n_feat))]),
                       sparse.hstack([In, sparse.csc_matrix((n_feat, m_data)), -In]),
                       sparse.hstack([In, sparse.csc_matrix((n_feat, m_data)), In])], format="csc")
    l = np.concatenate([b, -np.inf * np.ones(n_feat), np.zeros(n_feat)])
    u = np.concatenate([b, np.zeros(n_feat), np.inf * np.ones(n_feat)])
    return dict(P=sparse.triu(P, format="csc"), q=q, A=A, l=l, u=u, n_feat=n_feat,
                m_data=m_data, Ad=Ad)


def mpc_structure(N=12, dt=0.1):
    """C4: shared matrices of the horizon-12 3-D double-integrator MPC batch.

    States nx=6 (position, velocity in 3-D), inputs nu=4 (three forces plus a
    redundant vertical thruster), x0 eliminated: variables per stage (x_{k+1},
    u_k) -> n = 12*(6+4) = 120.  Rows: 72 dynamics equalities, 120 variable
    boxes, 48 input-rate rows (u_k - u_{k-1}; first row is u_0 alone) -> m = 240.
    (formulation after docs/examples/mpc.rst:81-101.)
    Returns P (triu), A, and a function building (q, l, u) from x0.
    """
    nx, nu = 6, 4
    Ad = np.eye(nx)
    Ad[:3, 3:] = dt * np.eye(3)
    Bd = np.zeros((nx, nu))
    Bd[:3, :3] = 0.5 * dt * dt * np.eye(3)
    Bd[3:, :3] = dt * np.eye(3)
    Bd[2, 3] = 0.5 * dt * dt * 0.5
    Bd[5, 3] = dt * 0.5
    Q = np.diag([10., 10., 10., 1., 1., 1.])
    QN = 5 * Q
    R = 0.1 * np.eye(nu)
    nvar = N * (nx + nu)
    xi = lambda k: slice(k * (nx + nu), k * (nx + nu) + nx)          # x_{k+1}
    ui = lambda k: slice(k * (nx + nu) + nx, (k + 1) * (nx + nu))    # u_k
    Pd = np.zeros((nvar, nvar))
    for k in range(N):
        Pd[xi(k), xi(k)] = QN if k == N - 1 else Q
        Pd[ui(k), ui(k)] = R
    Aeq = np.zeros((N * nx, nvar))
    for k in range(N):
        r = slice(k * nx, (k + 1) * nx)
        Aeq[r, xi(k)] = -np.eye(nx)
        Aeq[r, ui(k)] = Bd
        if k > 0:
            Aeq[r, xi(k - 1)] = Ad
    Abox = np.eye(nvar)
    Arate = np.zeros((N * nu, nvar))
    for k in range(N):
        r = slice(k * nu, (k + 1) * nu)
        Arate[r, ui(k)] = np.eye(nu)
        if k > 0:
            Arate[r, ui(k - 1)] = -np.eye(nu)
    A = sparse.csc_matrix(np.vstack([Aeq, Abox, Arate]))
    P = sparse.triu(sparse.csc_matrix(Pd), format="csc")
    xmax = np.array([5., 5., 5., 3., 3., 3.])
    umax = np.array([2., 2., 2., 1.])
    box_hi = np.tile(np.concatenate([xmax, umax]), N)
    rate = 1.5 * np.ones(N * nu)

    def vectors(x0):
        q = np.zeros(nvar)
        beq = np.zeros(N * nx)
        beq[:nx] = -Ad @ x0
        l = np.concatenate([beq, -box_hi, -rate])
        u = np.concatenate([beq, box_hi, rate])
        return q, l, u

    return dict(P=P, A=A, vectors=vectors, n=nvar, m=A.shape[0], nx=nx, nu=nu, N=N)


def mpc_batch(batch=1024, N=12, seed0=0):
    """q, l, u stacked per problem (seed = problem index) for the shared (P, A)."""
    s = mpc_structure(N)
    Q = np.zeros((batch, s["n"]))
    L = np.zeros((batch, s["m"]))
    U = np.zeros((batch, s["m"]))
    for b in range(batch):
        rng = np.random.default_rng(seed0 + b)
        x0 = 0.5 * rng.standard_normal(s["nx"])
        Q[b], L[b], U[b] = s["vectors"](x0)
    return s, Q, L, U


def portfolio_qp(n_blocks=400, block=125, sector_rows=0, seed=1):
    """C5: block-diagonal dense PSD P (triu CSC), A = [1'; I] (+ optional sparse
    sector rows); bounds after docs/examples/portfolio.rst:64-65."""
    rng = np.random.default_rng(seed)
    n = n_blocks * block
    blocks = []
    for _ in range(n_blocks):
        G = rng.standard_normal((block, block))
        blocks.append(np.triu(G @ G.T / block + 0.1 * np.eye(block)))
    P = sparse.block_diag([sparse.csc_matrix(b) for b in blocks], format="csc")
    mats = [sparse.csc_matrix(np.ones((1, n))), sparse.eye(n, format="csc")]
    l = [np.array([1.]), np.zeros(n)]
    u = [np.array([1.]), np.ones(n)]
    if sector_rows:
        S = sparse.random(sector_rows, n, density=0.01, format="csc", random_state=rng)
        mats.append(S); l.append(np.zeros(sector_rows)); u.append(0.3 * np.ones(sector_rows))
    A = sparse.vstack(mats, format="csc")
    q = -rng.standard_normal(n) * 0.1
    return dict(P=P, q=q, A=A, l=np.concatenate(l), u=np.concatenate(u))
