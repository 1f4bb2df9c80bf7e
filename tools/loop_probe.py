"""Steady-state cost of the ADMM loop on config 2: long windows, no termination checks, no rho updates.
usage: python tools/loop_probe.py [n m]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, osqp_amd
from osqp_amd.problems import random_sparse_qp
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 20000)
pb = random_sparse_qp(n, m, seed=1)
s = osqp_amd.OSQP().setup(**pb, eps_abs=1e-13, eps_rel=1e-13, max_iter=600, check_termination=300, adaptive_rho=0, warm_start=0)
s.solve()
st0 = s.stats()
t0 = time.perf_counter(); r = s.solve(); dt = time.perf_counter() - t0
st = s.stats()
pcg = st["pcg_iters_total"] - st0["pcg_iters_total"]
L = osqp_amd.lib()
L.hipeng_time_kernel.restype = C.c_int; L.hipeng_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
us = [C.c_double() for _ in range(5)]
for k, w in enumerate((0, 1, 5, 6, 7)):
    L.hipeng_time_kernel(s.engine(), w, 200, C.byref(us[k]))
print("iters %d in %.3f ms -> %.1f us per ADMM iteration; %.2f PCG iterations each; launches %d syncs %d" % (
    r.info.iter, 1e3 * dt, 1e6 * dt / r.info.iter, pcg / r.info.iter, st["graph_launches"] - st0["graph_launches"], st["host_syncs"] - st0["host_syncs"]))
print("kernel periods: A %.2f us, B %.2f us, init %.2f us -> PCG kernels alone %.1f us per ADMM iteration" % (
    us[0].value, us[1].value, us[2].value, (pcg / r.info.iter + 1) * (us[0].value + us[1].value)))
print("one PCG iteration in loop order (A then B): %.2f us; empty dependent launch: %.2f us" % (us[3].value, us[4].value))

if hasattr(L, "hipeng_timeline"):      # make TIMELINE=1 build: where the time of one ADMM iteration goes
    L.hipeng_timeline.restype = C.c_longlong; L.hipeng_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    buf = np.zeros(1 << 20, dtype=np.uint64)
    L.hipeng_timeline(s.engine(), buf.ctypes.data, buf.size)          # drop what the runs above left
    r = s.solve()
    k = L.hipeng_timeline(s.engine(), buf.ctypes.data, buf.size)
    ids = (buf[:k] >> np.uint64(56)).astype(int); ts = (buf[:k] & np.uint64((1 << 56) - 1)).astype(np.int64) * 10e-3   # us
    wsel = (ids >= 20) & (ids < 60)
    if wsel.any():           # per-wavefront stamps of one pipelined iteration (relative to wavefront 0 after the exchange)
        print("  one pipelined iteration, workgroup 0, us after the exchange (rows: after-exchange, products issued, scalars, barrier, update; columns: wavefronts 0..7)")
        for ph in range(5):
            row = []
            for wv in range(8):
                v = ts[ids == 20 + ph * 8 + wv]
                v = v[v < 1e4]
                row.append(v.mean() if v.size else float("nan"))
            print("   phase %d: " % ph + " ".join("%6.2f" % x for x in row))
    w1 = (ids >= 60) & (ids < 100)
    if w1.any():           # the FIRST trip of a launch (u0 read from memory): us since kernel entry, per wavefront
        print("  first trip of a launch, workgroup 0, us since kernel entry (rows: u0 in LDS, products issued, scalars, barrier passed, update; columns: wavefronts 0..7)")
        for ph in range(5):
            row = []
            for wv in range(8):
                v = ts[ids == 60 + ph * 8 + wv]
                v = v[v < 1e4]
                row.append(v.mean() if v.size else float("nan"))
            print("   phase %d: " % ph + " ".join("%6.2f" % x for x in row))
    wsel = wsel | w1
    ids, ts = ids[~wsel], ts[~wsel]
    order = np.argsort(ts, kind="stable"); ids, ts = ids[order], ts[order]
    per = np.diff(ts)
    names = {1: "k_pcg_init", 2: "k_cg_A", 3: "k_cg_B", 4: "k_admm_finalize"}
    print("timeline: %d kernel starts over %.1f us; %d ADMM iterations" % (k, ts[-1] - ts[0], r.info.iter))
    tot = 0.0
    for i in (1, 2, 3, 4):
        p = per[ids[:-1] == i]
        if p.size == 0: continue
        act = p[p >= 2.6]; noop = p[p < 2.6]
        print("  %-16s %6d launches: active %6d mean %5.2f us (sum %8.1f) | early-exit %6d mean %5.2f us (sum %8.1f) | > 30 us: %d (sum %.1f)" % (
            names[i], p.size, act.size, act[act < 30].mean() if (act < 30).any() else 0, act[act < 30].sum(), noop.size, noop.mean() if noop.size else 0, noop.sum(),
            (p >= 30).sum(), p[p >= 30].sum()))
    print("  per ADMM iteration: %.1f us total" % ((ts[-1] - ts[0]) / r.info.iter))
    if (ids == 5).any():       # resident PCG launches: phases of workgroup 0
        names = {5: "resident start", 10: "iteration top", 11: "flags seen", 12: "vector in LDS", 13: "products done", 14: "scalars in", 15: "kernel end"}
        nxt = {}
        for a, b, d in zip(ids[:-1], ids[1:], per):
            nxt.setdefault((a, b), []).append(d)
        for (a, b), v in sorted(nxt.items()):
            v = np.array(v)
            print("  %-16s -> %-16s %7d x mean %6.2f us (min %5.2f, max %6.2f)" % (names.get(a, a), names.get(b, b), v.size, v.mean(), v.min(), v.max()))
