"""Condense three rocprofv3 passes over the same bench command into one JSON for profiles/:
    python tools/profile_summary.py <kernel-trace dir> <pmc FETCH_SIZE dir> <pmc WRITE_SIZE dir> <out.json>
Durations come from the --kernel-trace pass, FETCH_SIZE / WRITE_SIZE from their own --pmc passes
(never combined with a trace, as the pool requires)."""
import csv, glob, json, statistics, sys, collections


def find(d, pat):
    f = sorted(glob.glob(d + "/**/" + pat, recursive=True))
    return f[0] if f else None


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def durations(d):
    out = collections.defaultdict(list)
    f = find(d, "*kernel_trace.csv")
    if not f: return out
    for r in csv.DictReader(open(f)):
        out[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out


def counters(d, counter):
    out = collections.defaultdict(list)
    f = find(d, "*counter_collection.csv")
    if not f: return out
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter: out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out


def pct(v, q):
    v = sorted(v); return v[min(len(v) - 1, int(q * len(v)))]


def main():
    tr, fe, wr, outp = sys.argv[1:5]
    dur, fetch, write = durations(tr), counters(fe, "FETCH_SIZE"), counters(wr, "WRITE_SIZE")
    res = {"note": "rocprofv3 on MI355X; separate passes: --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE. "
                   "FETCH/WRITE in KB as reported by the counters. Calibration (profiles/r02_fetch_calibration.json): FETCH_SIZE "
                   "reports 1/2 of the bytes of a coalesced streaming read at 4, 8 and 16 B per lane, WRITE_SIZE is exact, so "
                   "HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE ('traffic_bytes_corrected'). Launch counts include "
                   "early-exit launches of converged PCG iterations; FETCH/WRITE medians are over active launches (> 1 KB moved).",
           "kernels": {}}
    for k in sorted(set(dur) | set(fetch) | set(write)):
        e = {}
        for nm, src in (("FETCH_SIZE_KB", fetch), ("WRITE_SIZE_KB", write)):
            if src.get(k):
                act = [v for v in src[k] if v > 1.0] or src[k]
                e[nm] = {"median": statistics.median(act), "max": max(act), "launches": len(src[k]), "active_launches": len(act)}
        if dur.get(k):
            v = dur[k]
            e["duration_ns"] = {"median": statistics.median(v), "p10": pct(v, 0.1), "p90": pct(v, 0.9), "mean": sum(v) / len(v), "launches": len(v)}
        if "FETCH_SIZE_KB" in e and "WRITE_SIZE_KB" in e:
            e["traffic_bytes_corrected"] = round(1024.0 * (2.0 * e["FETCH_SIZE_KB"]["median"] + e["WRITE_SIZE_KB"]["median"]))
        res["kernels"][k] = e
    # which build the passes ran on: bench.py flags a `traffic` figure taken from another build of the engine as stale
    import hashlib, os
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "osqp_amd", "csrc", "engine.hip")
    try: res["engine_hip_sha256"] = hashlib.sha256(open(src, "rb").read()).hexdigest()
    except OSError: res["engine_hip_sha256"] = None
    json.dump(res, open(outp, "w"), indent=1)
    for k, e in res["kernels"].items():
        print(k, {kk: (vv["median"] if isinstance(vv, dict) else vv) for kk, vv in e.items()})


if __name__ == "__main__":
    main()
