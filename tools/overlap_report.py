#!/usr/bin/env python3
"""Developer tool: from a rocprofv3 kernel_trace.csv, report how much of the small kernels' time overlaps k_scan."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
ev.sort()
scans = [(s, e) for s, e, n, q in ev if "k_scan<false" in n]
tail = scans[len(scans)//2:]
t0, t1 = tail[0][0], tail[-1][1]
print("scan launches in window:", len(tail), "window ms:", (t1 - t0) / 1e6, "per-step ms:", (t1 - t0) / 1e6 / max(1, len(tail) - 1) if len(tail) > 1 else 0)
busy = sum(e - s for s, e in tail)
print("scan busy fraction:", busy / (t1 - t0))
def overlap(s, e):
    return sum(max(0, min(e, b) - max(s, a)) for a, b in tail)
agg = {}
for s, e, n, q in ev:
    if s < t0 or e > t1 or "k_scan<false" in n: continue
    key = n.split("(")[0][:40]
    a = agg.setdefault(key, [0, 0, 0, set()])
    a[0] += 1; a[1] += e - s; a[2] += overlap(s, e); a[3].add(q)
for k, (c, d, o, qs) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:42s} n={c:4d} avg={d/c/1e3:8.1f}us  overlapped_with_scan={o/max(1,d):5.2f} queues={sorted(qs)}")
print("scan queues:", sorted({q for s, e, n, q in ev if 'k_scan<false' in n}))
