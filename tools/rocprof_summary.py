#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (csv or rocpd sqlite) per kernel:
calls, avg/min/max/total duration.  usage: rocprof_summary.py <dir-or-file> [--timeline N]"""
import csv, glob, os, sqlite3, sys, collections

def short(n):
    n = n.replace("acx::", "")
    if "rocprim" in n:
        for k in ("onesweep_iteration", "onesweep_global_offsets", "scan_impl", "lookback", "radix_sort"):
            if k in n: return "rocprim::" + k
        return "rocprim::other"
    return n.split("(")[0][:70]

def load(path):
    rows = []
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True) + glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    for f in files:
        if f.endswith(".csv"):
            for r in csv.DictReader(open(f)):
                rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        else:
            db = sqlite3.connect(f)
            for n, s, e in db.execute("select name, start, end from kernels"):
                rows.append((n, s, e))
    return sorted(rows, key=lambda r: r[1])

if __name__ == "__main__":
    rows = load(sys.argv[1])
    agg = collections.OrderedDict()
    for n, s, e in rows:
        agg.setdefault(short(n), []).append((e - s) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_us':>11s} {'%':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:72s} {len(v):6d} {sum(v)/len(v):10.2f} {min(v):10.2f} {max(v):10.2f} {sum(v):11.2f} {100*sum(v)/tot:6.2f}")
    if "--timeline" in sys.argv:
        n = int(sys.argv[sys.argv.index("--timeline") + 1])
        t0 = rows[-n][1]
        print("\nlast", n, "dispatches (start_us relative, dur_us, gap_us):")
        prev = None
        for name, s, e in rows[-n:]:
            gap = (s - prev) / 1e3 if prev else 0
            print(f"  {(s - t0)/1e3:10.1f} {(e - s)/1e3:9.1f} {gap:8.1f}  {short(name)}")
            prev = e
