"""Config 5 (portfolio, n=50000, 400 dense 125x125 blocks) and config 3 (Lasso) kernel timings."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, osqp_amd
from osqp_amd.problems import portfolio_qp, lasso_qp
L = osqp_amd.lib()
L.hipeng_time_kernel.restype = C.c_int; L.hipeng_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
L.hipeng_kernel_bytes.restype = C.c_int; L.hipeng_kernel_bytes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
which = sys.argv[1:] or ["portfolio", "lasso"]
for name in which:
    pb = portfolio_qp() if name == "portfolio" else {k: v for k, v in lasso_qp().items() if k in "PqAlu"}
    t0 = time.perf_counter()
    s = osqp_amd.OSQP().setup(**pb, eps_abs=1e-4, eps_rel=1e-4, max_iter=int(os.environ.get('PROBE_MAX_ITER', 4000)))
    ts = time.perf_counter() - t0
    t0 = time.perf_counter(); r = s.solve(); tv = time.perf_counter() - t0
    st = s.stats()
    print("%s: setup %.2fs; %s in %d iters, %.3fs -> %.1f it/s; pcg/it %.1f; %.1f us per PCG iteration" % (
        name, ts, r.info.status, r.info.iter, tv, r.info.iter / tv, st["pcg_iters_total"] / r.info.iter, 1e6 * tv / st["pcg_iters_total"]))
    us = C.c_double(); L.hipeng_time_kernel(s.engine(), 5, 50, C.byref(us)); print("   k_pcg_init: %.1f us" % us.value)
    if st.get("resident"):
        L.hipeng_resident_info.restype = C.c_int; L.hipeng_resident_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        info = (C.c_longlong * 16)(); up = C.c_double()
        s.update_settings(max_iter=100); s.solve(); s.update_settings(max_iter=4000)      # a state in the middle of a solve
        L.hipeng_time_kernel(s.engine(), 8, 50, C.byref(up)); L.hipeng_resident_info(s.engine(), info)
        print("   resident launch (form %d): %.1f us for %d PCG iterations -> %.2f us each" % (info[9], up.value - us.value, info[6], (up.value - us.value) / max(1, info[6])))
    for k, nm in ((3, "k_cg_A update-only (split mode)"), (4, "k_cg_A apply-only (split mode)")):
        us = C.c_double(); L.hipeng_time_kernel(s.engine(), k, 100, C.byref(us)); print("   %s: %.1f us" % (nm, us.value))
    for k, nm in enumerate(("k_cg_A", "k_cg_B")):
        us = C.c_double(); by = C.c_double()
        L.hipeng_time_kernel(s.engine(), k, 100, C.byref(us)); L.hipeng_kernel_bytes(s.engine(), k, C.byref(by))
        print("   %s: %.1f us, %.1f MB -> %.0f GB/s" % (nm, us.value, by.value / 1e6, by.value / us.value / 1e3))
    if hasattr(L, "hipeng_timeline") and st.get("resident") and name == "portfolio":     # make TIMELINE=1: phases of k_pcg_blockres, workgroup 0
        L.hipeng_timeline.restype = C.c_longlong; L.hipeng_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
        buf = np.zeros(1 << 20, dtype=np.uint64)
        L.hipeng_timeline(s.engine(), buf.ctypes.data, buf.size)
        s.update_settings(max_iter=200); s.solve(); s.update_settings(max_iter=4000)
        k = L.hipeng_timeline(s.engine(), buf.ctypes.data, buf.size)
        ids = (buf[:k] >> np.uint64(56)).astype(int); ts = (buf[:k] & np.uint64((1 << 56) - 1)).astype(np.int64) * 10e-3
        sel = (ids >= 40) & (ids <= 46)
        ii, tt = ids[sel], ts[sel]
        names = {40: "iteration top", 41: "product done", 42: "partials in LDS", 43: "granules in", 44: "totals", 45: "update done", 46: "kernel end"}
        nxt = {}
        for a, b, d in zip(ii[:-1], ii[1:], np.diff(tt)):
            if 0 <= d < 1e4: nxt.setdefault((a, b), []).append(d)
        for (a, b), v in sorted(nxt.items()):
            v = np.array(v)
            print("   %-16s -> %-16s %6d x mean %6.2f us (median %5.2f, p90 %5.2f, max %6.2f)" % (names[a], names[b], v.size, v.mean(), np.median(v), np.percentile(v, 90), v.max()))
        o = np.argsort(ts, kind="stable"); ids2, ts2 = ids[o], ts[o]
        for a, b, nm in ((1, 6, "k_pcg_init start -> k_pcg_blockres start"), (46, 4, "k_pcg_blockres end -> k_admm_finalize start"), (4, 1, "k_admm_finalize start -> k_pcg_init start")):
            v = np.array([d for x, y, d in zip(ids2[:-1], ids2[1:], np.diff(ts2)) if x == a and y == b and d < 1e4])
            if v.size: print("   %-48s %6d x mean %6.2f us (median %5.2f)" % (nm, v.size, v.mean(), np.median(v)))
