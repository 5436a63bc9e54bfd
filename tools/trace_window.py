#!/usr/bin/env python3
"""Developer tool: print a window of a rocprofv3 kernel_trace.csv as (start, end, duration, queue, kernel) relative
to the window start — to see which kernels of neighbouring batches actually overlap."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-38:], r.get("Queue_Id", "?")) for r in rows)
scans = [i for i, e in enumerate(ev) if "k_scan<false" in e[2]]
mid = scans[len(scans) * 2 // 3]
t0 = ev[mid][0]
for s, e, n, q in ev:
    if t0 - 60_000 <= s <= t0 + 900_000:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}us q{q} {n}")
