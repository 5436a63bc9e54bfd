#!/bin/bash
# HBM traffic of k_scan from the PMC counters (separate passes, MI355X_MICROARCH.md §HBM): run on the GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/tools/scan_perf.py --rows ${ROWS:-10000000} --steps 4 --mode sync > $R/gpurun_out/pmc_$c.log 2>&1
done
python3 - <<PY
import csv, glob
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/gpurun_out/pmc_%s/*/*counter_collection.csv" % c)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_scan<false" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c] = vals
    print(c, "launches", len(vals), "mean per launch (KiB):", sum(vals) / max(1, len(vals)))
fetch = sum(out["FETCH_SIZE"]) / len(out["FETCH_SIZE"]); write = sum(out["WRITE_SIZE"]) / len(out["WRITE_SIZE"])
print("traffic bytes per launch (2*FETCH+WRITE)*1024 =", (2 * fetch + write) * 1024)
PY
