#!/usr/bin/env python3
"""Developer tool: from a rocprofv3 kernel_trace.csv of a pipelined search loop, report what sits between consecutive
main scans (k_scan<false,...>): the scan-to-scan period, the gap from one scan's end to the next one's start, and for
each side kernel when it started / ended relative to the previous scan's end (steady-state half of the trace)."""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void anr::", "").replace("anr::", "")) for r in rows)
scans = [e for e in ev if e[2].startswith("k_scan<false")]
scans = scans[len(scans) // 3:]
per = [(b[0] - a[0]) / 1e3 for a, b in zip(scans[:-1], scans[1:])]
gap = [(b[0] - a[1]) / 1e3 for a, b in zip(scans[:-1], scans[1:])]
dur = [(a[1] - a[0]) / 1e3 for a in scans]
q = lambda v, p: sorted(v)[min(len(v) - 1, int(p * len(v)))]
print(f"main scans: {len(scans)}  duration us: median {st.median(dur):.1f} p10 {q(dur, .1):.1f} p90 {q(dur, .9):.1f}")
print(f"scan start -> next scan start us: median {st.median(per):.1f} mean {st.mean(per):.1f} p10 {q(per, .1):.1f} p90 {q(per, .9):.1f} max {max(per):.1f}")
print(f"scan end -> next scan start us:   median {st.median(gap):.1f} mean {st.mean(gap):.1f} p10 {q(gap, .1):.1f} p90 {q(gap, .9):.1f} max {max(gap):.1f}")
rel = {}
for a, b in zip(scans[:-1], scans[1:]):
    for s, e, n in ev:
        if n.startswith("k_scan<false") or e < a[1] - 400_000 or s > b[0]:
            continue
        if a[0] < e and s < b[0] and e > a[1] - 150_000:  # kernels alive near the boundary
            rel.setdefault(n, []).append(((s - a[1]) / 1e3, (e - a[1]) / 1e3))
print("side kernels relative to the previous main scan's END (us): start / end medians, n per boundary")
for n, v in sorted(rel.items(), key=lambda kv: st.median(x[1] for x in kv[1])):
    print(f"  {n[:44]:44s} start {st.median(x[0] for x in v):8.1f}  end {st.median(x[1] for x in v):8.1f}  dur {st.median(x[1]-x[0] for x in v):7.1f}  n/boundary {len(v)/max(1,len(scans)-1):.2f}")
