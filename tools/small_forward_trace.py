#!/usr/bin/env python3
"""Developer tool: per-kernel durations and the gaps between them for ONE small forward (a query at a time), from a
rocprofv3 kernel trace of `tools/enc_perf.py 1 32` (or any B L): the last forward of the trace, kernel by kernel."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# forwards start at k_embed_ln*; take the last complete one
starts = [i for i, r in enumerate(rows) if "k_embed_ln" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
seq = rows[a:b]
t_first, t_last = int(seq[0]["Start_Timestamp"]), int(seq[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq)
print(f"{len(seq)} kernels, span {(t_last - t_first) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, gaps {(t_last - t_first - busy) / 1e3:.1f} us")
agg = {}
prev_end = None
for r in seq:
    n = r["Kernel_Name"].split("(")[0].replace("void anr::", "").replace("anr::", "")[:40]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    g = int(r["Start_Timestamp"]) - prev_end if prev_end is not None else 0
    prev_end = int(r["End_Timestamp"])
    x = agg.setdefault(n, [0, 0, 0])
    x[0] += 1; x[1] += d; x[2] += g
for n, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:42s} n={c:3d} dur avg {d / c / 1e3:6.2f} us  gap before avg {g / c / 1e3:6.2f} us  total {(d + g) / 1e3:7.1f} us")
