#!/usr/bin/env python3
"""Developer tool: the last N kernel dispatches of a rocprofv3 --kernel-trace run, with start offsets and durations (us)."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%9.1f %8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]))
