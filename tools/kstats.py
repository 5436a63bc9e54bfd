#!/usr/bin/env python3
"""Developer tool: print the per-kernel rows of a rocprofv3 --kernel-trace --stats --output-format csv directory."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:10.1f} us min {float(r["MinNs"])/1e3:10.1f} us {r["Percentage"]:>6s} %')
