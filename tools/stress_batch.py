"""Randomised parity sweep of the batched engine (one workgroup per QP) against the CPU checker:
random shared patterns with n <= 128 (both tile shapes), ragged rows/columns, dense rows,
equalities, infinite bounds, infeasible members, several settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import sparse
import osqp_amd, oracle.oracle as orc
orc.build()


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    only = set(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else None
    rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", 99)))
    bad = 0
    for case in range(cases):
        n = int(rng.integers(3, 129)); m = int(rng.integers(1, min(400, 3 * n + 5)))
        dens = rng.uniform(1.5, 10) / n
        A = sparse.random(m, n, density=min(1.0, dens), format="lil", random_state=rng)
        if rng.random() < 0.4: A[int(rng.integers(0, m)), :] = rng.uniform(0.5, 1.5, n)      # a dense row
        if rng.random() < 0.3: A[:, int(rng.integers(0, n))] = rng.uniform(-1, 1, (m, 1))   # a dense column
        A = A.tocsc()
        Ph = sparse.random(n, n, density=min(1.0, rng.uniform(1, 4) / n), format="csc", random_state=rng)
        P = sparse.triu((Ph @ Ph.T + sparse.diags(rng.uniform(0.01, 1.0, n))).tocsc()).tocsc()
        B = 12
        Q = rng.standard_normal((B, n)); L = np.zeros((B, m)); U = np.zeros((B, m))
        for b in range(B):
            ax = A @ (0.3 * rng.standard_normal(n))
            L[b] = ax - rng.uniform(0, 1, m); U[b] = ax + rng.uniform(0, 1, m)
            eq = rng.random(m) < 0.15; L[b, eq] = U[b, eq] = ax[eq]
            inf = rng.random(m) < 0.1; U[b, inf] = 1e30
        if m >= 2 and rng.random() < 0.3:        # an infeasible member: two contradictory copies of one row
            A = sparse.vstack([A, A[0]], format="csc"); m += 1
            L = np.hstack([L, L[:, :1]]); U = np.hstack([U, U[:, :1]])
            L[3, -1] = U[3, 0] + 5.0; U[3, -1] = U[3, 0] + 6.0
        kw = dict(eps_abs=1e-4, eps_rel=1e-4)
        if rng.random() < 0.3: kw["scaling"] = 0
        if rng.random() < 0.3: kw["alpha"] = float(rng.uniform(1.0, 1.8))
        if rng.random() < 0.3: kw["rho"] = float(10 ** rng.uniform(-2, 1))
        if rng.random() < 0.2: kw["check_termination"] = int(rng.integers(1, 40))
        kw["max_iter"] = 1000
        if only is not None and case not in only: continue
        try:
            r = osqp_amd.BatchOSQP().setup(P, A, Q, L, U, **kw).solve()
        except Exception as e:
            print("case %d n=%d m=%d: setup/solve raised %s" % (case, n, m, e)); bad += 1; continue
        msgs = []
        for b in range(B):
            ro = orc.OracleOSQP().setup(P=P, q=Q[b], A=A, l=L[b], u=U[b], **kw).solve()
            if r.status_val[b] != ro.info.status_val: msgs.append("qp%d status %d vs %d" % (b, r.status_val[b], ro.info.status_val))
            elif r.iter[b] != ro.info.iter: msgs.append("qp%d iter %d vs %d" % (b, r.iter[b], ro.info.iter))
            elif ro.info.status == "solved":
                ex, ey = rel(r.x[b], ro.x), rel(r.y[b], ro.y)
                if ex > 1e-5 or ey > 1e-5: msgs.append("qp%d x %.1e y %.1e" % (b, ex, ey))
        if msgs:
            bad += 1
            print("case %d (n=%d m=%d nnzA=%d %s): %s" % (case, n, m, A.nnz, kw, "; ".join(msgs[:4])), flush=True)
    print("done: %d cases, %d flagged" % (cases, bad))


if __name__ == "__main__":
    main()
