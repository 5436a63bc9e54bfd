#!/bin/bash
# Run on the GPU box: headline bench + rocprofv3 kernel stats of the same command + PMC traffic of k_scan.
# Outputs land in gpurun_out/; copy the summaries into profiles/ afterwards (tools/collect_profiles.py).
set -e
R=$GRAFT_REPO_ROOT
python3 $R/bench.py > $R/gpurun_out/bench_final.json 2> $R/gpurun_out/bench_final.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats -- python3 $R/bench.py --no-cpu --recall-queries 0 > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/bench_prof.err
ROWS=10000000 bash $R/tools/pmc_traffic.sh > $R/gpurun_out/pmc_summary.txt 2>&1
cd $R
cp $(ls gpurun_out/kstats/*/*kernel_stats.csv | head -1) gpurun_out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  python3 - "$c" <<'PY'
import csv, glob, sys
c = sys.argv[1]
f = glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % c)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_scan<false" in r["Kernel_Name"] and r["Counter_Name"] == c]
with open("gpurun_out/pmc_%s_k_scan.csv" % c, "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
done
rm -rf gpurun_out/kstats gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
tail -3 gpurun_out/pmc_summary.txt
