"""Per-workgroup start/end times of one steady-state batch solve (debug build: make BATCH_DEBUG=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OSQP_AMD_BATCH_PROFILE"] = "1"
import numpy as np, osqp_amd
from osqp_amd.problems import mpc_batch
s, Q, L, U = mpc_batch(1024)
bs = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U, warm_start=0)
for _ in range(3): bs.solve(fetch=False)
r = bs.solve()
t0 = r.dual_inf_cert[:, 13] / 100.0; t1 = r.dual_inf_cert[:, 14] / 100.0   # us
base = t0.min(); t0 -= base; t1 -= base
dur = t1 - t0
print("kernel span %.1f us; sum of WG durations / 256 = %.1f us; longest WG %.1f us" % (t1.max(), dur.sum() / 256, dur.max()))
print("WG start times: first 256 by %.1f us; last start %.1f us" % (np.sort(t0)[255], t0.max()))
it = r.iter
for lo, hi in ((25, 25), (50, 50), (75, 75), (100, 125), (200, 200)):
    m = (it >= lo) & (it <= hi)
    if m.any(): print("  iters %d..%d: n=%d, duration mean %.1f us (%.2f us/iter incl. everything)" % (lo, hi, m.sum(), dur[m].mean(), (dur[m] / it[m]).mean()))
ends = np.sort(t1)
print("end-time percentiles (us): 50%% %.0f, 90%% %.0f, 99%% %.0f, max %.0f" % (ends[512], ends[921], ends[1013], ends[-1]))
# idle estimate: per-CU unknown, but total busy vs span
print("busy fraction = %.2f" % (dur.sum() / (256 * t1.max())))
