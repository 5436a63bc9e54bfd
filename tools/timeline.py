#!/usr/bin/env python3
"""Developer tool: from a rocprofv3 kernel_trace.csv print the per-step kernel timeline (durations and gaps)
averaged over the steady-state steps.  A step starts at each k_prepq launch."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
starts = [i for i, e in enumerate(ev) if "k_prepq" in e[2]]
steps = [ev[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = steps[len(steps) // 2:]
sig = collections.Counter(tuple(e[2] for e in s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if tuple(e[2] for e in s) == sig]
print("steps:", len(steps), "kernels per step:", len(sig))
n = len(steps)
tot = 0.0
for i, name in enumerate(sig):
    dur = sum(s[i][1] - s[i][0] for s in steps) / n / 1e3
    gap = sum((s[i][0] - s[i - 1][1]) for s in steps) / n / 1e3 if i else 0.0
    tot += dur + gap
    print(f"{name[:60]:60s} gap={gap:7.1f}us dur={dur:8.1f}us")
span = sum(s[-1][1] - s[0][0] for s in steps) / n / 1e3
period = (steps[-1][0][0] - steps[0][0][0]) / max(1, n - 1) / 1e3
print(f"span first-start..last-end = {span:.1f}us; step period = {period:.1f}us")
