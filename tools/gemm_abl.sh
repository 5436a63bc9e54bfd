#!/bin/bash
# Developer tool (GPU box): per-kernel times of the encoder GEMMs for the schedule ablations of k_gemm_pp
# (build csrc with `make EXTRA=-DANR_GEMM_ABLATIONS` first; ablated builds compute wrong results).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for abl in ${ABLS:-0 1 2 3}; do
  export ANORAG_GEMM_ABL=$abl
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl$abl -- python3 $R/tools/enc_perf.py ${ENC_B:-256} ${ENC_L:-64} > $R/gpurun_out/abl$abl.log 2>&1
  python3 - $abl <<'PY'
import csv, glob, os, sys
abl = sys.argv[1]
f = max(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/abl%s/*/*kernel_stats.csv" % abl), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if "gemm" in r["Name"]:
        print("ABL", abl, r["Name"][:44].ljust(44), "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
