cd /tmp && export TMPDIR=/tmp
for st in 5 6 7 4; do
  export ANORAG_SEL_STOP=$st
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sel$st -- python3 $GRAFT_REPO_ROOT/tools/scan_perf.py --rows 1250000 --steps 10 --mode sync > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/sel$st/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in rows)
sel=[(e[1]-e[0])/1e3 for e in ev if 'k_select' in e[2]]
print("stop=$st", sorted(set(round(x) for x in sel if x < 300)))
PY
done
