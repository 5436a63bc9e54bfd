#!/bin/bash
# usage (on the GPU box): tools/pmc_kernel.sh <kernel-substring> <counters...> -- <python script and args>
# one rocprofv3 --pmc pass per counter (with --kernel-trace only), prints the per-launch mean of each counter
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift
CS=()
while [ "$1" != "--" ]; do CS+=("$1"); shift; done
shift
for c in "${CS[@]}"; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmck_$c -- python3 "$@" > /dev/null 2>&1
  python3 - "$c" "$K" "$R" <<'PY'
import csv, glob, sys
c, k, R = sys.argv[1:4]
f = glob.glob(f"{R}/gpurun_out/pmck_{c}/*/*counter_collection.csv")[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if k in r["Kernel_Name"] and r["Counter_Name"] == c]
print(f"{c:32s} launches={len(v):4d} mean={sum(v)/max(1,len(v)):.4g}")
PY
  rm -rf $R/gpurun_out/pmck_$c
done
