"""Analyse a rocprofv3 kernel trace of one engine run: busy time, gaps, no-op launches.
usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# keep only the last third of the single-QP engine kernels (steady-state steps)
eng = [r for r in rows if r[2].startswith(("k_cg", "k_pcg", "k_admm", "k_resid", "k_final", "k_precond", "k_refresh"))]
print("engine kernels:", len(eng))
busy = collections.Counter(); cnt = collections.Counter(); noop = collections.Counter(); noop_t = collections.Counter()
gaps = []
for i, (s, e, n) in enumerate(eng):
    d = e - s
    busy[n] += d; cnt[n] += 1
    if n.startswith("k_cg") and d < 2600: noop[n] += 1; noop_t[n] += d
    if i: gaps.append(s - eng[i - 1][1])
span = eng[-1][1] - eng[0][0]
print("span %.3f ms  busy %.3f ms" % (span / 1e6, sum(busy.values()) / 1e6))
for n in busy: print("  %-18s n=%6d busy=%8.3f ms avg=%6.2f us noop=%d (%.3f ms)" % (n, cnt[n], busy[n] / 1e6, busy[n] / cnt[n] / 1e3, noop[n], noop_t[n] / 1e6))
import statistics
small = [g for g in gaps if g < 5000]; big = [g for g in gaps if g >= 5000]
print("gaps<5us: n=%d sum=%.3f ms median=%.2f us;  gaps>=5us: n=%d sum=%.3f ms" % (len(small), sum(small) / 1e6, statistics.median(small) / 1e3, len(big), sum(big) / 1e6))
big.sort(reverse=True); print("largest gaps (us):", [round(g / 1e3) for g in big[:12]])
