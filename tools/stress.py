"""Randomised parity sweep of the HIP engine against the CPU checker (not part of the test
suite: a few minutes on the GPU box).  Mixes structure the unit tests hold one at a time:
short rows, long rows (>= 512 entries, one wavefront), huge rows (>= 8192, sliced over the
grid), dense diagonal blocks of P (with and without stray off-block entries), equality rows,
infinite bounds, scaling on/off, alpha/rho/sigma variations, updates and warm starts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import sparse
import osqp_amd
import oracle.oracle as orc

orc.build()


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def make(rng, kind):
    if kind == "sparse":
        n = int(rng.integers(20, 400)); m = int(rng.integers(1, 2 * n))
        A = sparse.random(m, n, density=min(1.0, rng.uniform(2, 12) / n), format="csc", random_state=rng)
        Ph = sparse.random(n, n, density=min(1.0, rng.uniform(1, 5) / n), format="csc", random_state=rng)
        P = (Ph @ Ph.T + sparse.diags(rng.uniform(0.0, 1.0, n))).tocsc()
    elif kind == "longrows":
        n = int(rng.integers(700, 1500)); m = int(rng.integers(5, 40))
        A = sparse.vstack([sparse.random(m, n, density=0.7, format="csc", random_state=rng),
                           sparse.random(30, n, density=5.0 / n, format="csc", random_state=rng)], format="csc")
        m = A.shape[0]
        P = sparse.diags(rng.uniform(0.1, 2.0, n)).tocsc()
    elif kind == "huge":
        n = int(rng.integers(8300, 9500))
        rows = [sparse.csc_matrix(np.ones((1, n)))]
        if rng.random() < 0.5: rows.append(sparse.csc_matrix(rng.uniform(0.5, 1.5, (1, n))))
        rows.append(sparse.eye(n, format="csc"))
        A = sparse.vstack(rows, format="csc"); m = A.shape[0]
        P = sparse.diags(rng.uniform(0.5, 2.0, n)).tocsc()
    elif kind == "blocks":
        nb = int(rng.integers(2, 6)); b = int(rng.integers(32, 100))
        n = nb * b + int(rng.integers(0, 20))
        blocks = []
        for _ in range(nb):
            G = rng.standard_normal((b, b)); blocks.append(G @ G.T / b + 0.1 * np.eye(b))
        tail = n - nb * b
        if tail: blocks.append(np.diag(rng.uniform(0.1, 1.0, tail)))
        P = sparse.block_diag(blocks, format="lil")
        if rng.random() < 0.5 and nb > 1:      # a stray coupling between two blocks: they must merge or fall back
            P[3, b + 4] = 0.01; P[b + 4, 3] = 0.01
            P[3, 3] += 1.0; P[b + 4, b + 4] += 1.0
        P = P.tocsc()
        m = int(rng.integers(2, n // 2)) + n
        A = sparse.vstack([sparse.random(m - n, n, density=6.0 / n, format="csc", random_state=rng), sparse.eye(n, format="csc")], format="csc")
    q = rng.standard_normal(n)
    x0 = rng.standard_normal(n) * 0.3
    Ax = A @ x0
    l = Ax - rng.uniform(0.0, 1.0, m); u = Ax + rng.uniform(0.0, 1.0, m)
    eq = rng.random(m) < 0.15; l[eq] = Ax[eq]; u[eq] = Ax[eq]
    inf = rng.random(m) < 0.1; u[inf] = 1e30
    inf2 = rng.random(m) < 0.05; l[inf2] = -1e30
    return dict(P=sparse.triu(P).tocsc(), q=q, A=A, l=l, u=u)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    only = set(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else None
    rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", 2026)))
    kinds = ["sparse"] * 6 + ["longrows"] * 2 + ["blocks"] * 3 + ["huge"]
    bad = 0; t0 = time.time()
    for case in range(n_cases):
        kind = kinds[case % len(kinds)]
        pb = make(rng, kind)
        kw = dict(eps_abs=1e-4, eps_rel=1e-4)
        if rng.random() < 0.3: kw["scaling"] = 0
        if rng.random() < 0.3: kw["alpha"] = float(rng.uniform(1.0, 1.8))
        if rng.random() < 0.3: kw["rho"] = float(10 ** rng.uniform(-2, 1))
        if rng.random() < 0.2: kw["adaptive_rho_interval"] = int(rng.integers(10, 60))
        if rng.random() < 0.2: kw["check_termination"] = int(rng.integers(1, 40))
        if kind == "huge": kw["max_iter"] = 300
        steps = ["solve"]
        if rng.random() < 0.4: steps += ["update_q", "solve"]
        if rng.random() < 0.3 and kind != "huge": steps += ["update_A", "solve"]
        if only is not None and case not in only:
            for st in steps:     # keep the random stream aligned with a full run
                if st == "update_q": rng.standard_normal(pb["q"].size)
                elif st == "update_A": rng.standard_normal(pb["A"].nnz)
            continue
        sg = osqp_amd.OSQP().setup(**pb, **kw); so = orc.OracleOSQP().setup(**pb, **kw)
        msgs = []
        for st in steps:
            if st == "update_q":
                q2 = pb["q"] + 0.1 * rng.standard_normal(pb["q"].size); sg.update(q=q2); so.update(q=q2)
            elif st == "update_A":
                A = sparse.csc_matrix(pb["A"]); A.sort_indices()
                Ax2 = A.data * (1.0 + 0.01 * rng.standard_normal(A.nnz)); sg.update(Ax=Ax2); so.update(Ax=Ax2)
            else:
                rg, ro = sg.solve(), so.solve()
                if only is not None:
                    print("   case %d %s: gpu %s it %d obj %.10g pri %.3e dua %.3e rho_upd %d | cpu %s it %d obj %.10g pri %.3e dua %.3e rho_upd %d | stats %s" % (
                        case, st, rg.info.status, rg.info.iter, rg.info.obj_val, rg.info.pri_res, rg.info.dua_res, rg.info.rho_updates,
                        ro.info.status, ro.info.iter, ro.info.obj_val, ro.info.pri_res, ro.info.dua_res, ro.info.rho_updates, sg.stats()))
                if rg.info.status != ro.info.status: msgs.append("status %s vs %s" % (rg.info.status, ro.info.status))
                elif rg.info.iter != ro.info.iter: msgs.append("iter %d vs %d" % (rg.info.iter, ro.info.iter))
                elif ro.info.status in ("solved", "maximum iterations reached"):
                    ex, ey = rel(rg.x, ro.x), rel(rg.y, ro.y)
                    if ex > 1e-5 or ey > 1e-5: msgs.append("x %.1e y %.1e" % (ex, ey))
        stt = sg.stats()
        if stt["pcg_forced"]: msgs.append("pcg_forced %d" % stt["pcg_forced"])
        if msgs:
            bad += 1
            print("case %d (%s, n=%d m=%d, %s): %s" % (case, kind, pb["q"].size, pb["l"].size, kw, "; ".join(msgs)), flush=True)
        if case % 20 == 19: print("... %d cases, %d flagged, %.0fs" % (case + 1, bad, time.time() - t0), flush=True)
    print("done: %d cases, %d flagged" % (n_cases, bad))
    sys.exit(1 if bad else 0)        # (round 3: 0 flagged with the default PCG cap of max(20000, 10 n); max(1000, 2 n) left 10 cases with capped solves)


if __name__ == "__main__":
    main()
