"""One QP per stream: B resident engines of one process on one GPU, solved one after the other and from a thread pool.
With grids sized to the problem (8-60 workgroups each) their resident windows share the device's CU budget instead of
taking turns.   usage: python tools/multi_stream_probe.py [B] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, osqp_amd
from osqp_amd.problems import random_sparse_qp
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
ss = [osqp_amd.OSQP().setup(**random_sparse_qp(n, 2 * n, seed=50 + k), warm_start=0) for k in range(B)]
ref = [s.solve() for s in ss]
t0 = time.perf_counter()
for _ in range(5):
    seq = [s.solve() for s in ss]
t_seq = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for _ in range(5):
    par = osqp_amd.solve_many(ss, max_workers=B)
t_par = (time.perf_counter() - t0) / 5
same = all(np.array_equal(a.x, b.x) and a.info.iter == b.info.iter for a, b in zip(seq, par))
it = sum(r.info.iter for r in seq)
print("%d QPs n=%d (resident=%d each): one after the other %.2f ms (%.0f it/s), %d threads %.2f ms (%.0f it/s): x%.2f; bit-identical: %s" % (
    B, n, ss[0].stats()["resident"], 1e3 * t_seq, it / t_seq, B, 1e3 * t_par, it / t_par, t_seq / t_par, same))
