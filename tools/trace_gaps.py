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
eng = [r for r in rows if r[2].replace("void ", "").startswith(("k_cg", "k_pcg", "k_admm", "k_resid", "k_final", "k_precond", "k_refresh", "k_form", "__amd_rocclr"))]
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
# what precedes the gaps of 5 us and more
prev = collections.Counter(); prev_t = collections.Counter()
for i in range(1, len(eng)):
    g = eng[i][0] - eng[i - 1][1]
    if g >= 5000: key = eng[i - 1][2].replace("void ", "") + " -> " + eng[i][2].replace("void ", ""); prev[key] += 1; prev_t[key] += g
for k, v in sorted(prev_t.items(), key=lambda kv: -kv[1])[:12]: print("  gap after/before %-60s n=%4d sum=%7.3f ms avg=%6.1f us" % (k, prev[k], v / 1e6, v / prev[k] / 1e3))

# one bench step of config 2 (osqp_update_rho + cold-started solve: two k_form_K launches, the second at the rho update
# inside the solve): the last complete one
fk = [i for i, r in enumerate(eng) if r[2].replace("void ", "").startswith("k_form_K")]
if len(fk) >= 4:
    a, b = fk[-4], fk[-2]            # second-to-last step: from its update_rho to the next step's
    step = eng[a:b]
    span = step[-1][1] - step[0][0]
    busy = sum(e - s for s, e, _ in step)
    print("last complete step: %d launches, span %.3f ms, busy %.3f ms, idle %.3f ms" % (len(step), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    cat = collections.Counter(); catn = collections.Counter()
    for i in range(1, len(step)):
        g = step[i][0] - step[i - 1][1]
        if g > 0: key = step[i - 1][2].replace("void ", "") + " -> " + step[i][2].replace("void ", ""); cat[key] += g; catn[key] += 1
    for k, v in sorted(cat.items(), key=lambda kv: -kv[1])[:14]: print("    idle %-62s n=%4d sum=%7.1f us avg=%6.1f us" % (k, catn[k], v / 1e3, v / catn[k] / 1e3))
