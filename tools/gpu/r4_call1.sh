set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/microbench/ticket_atomic > $O/ticket_atomic.txt 2>&1
python3 $R/tools/scan_perf.py --rows 1250000 --steps 60 > $O/shard_base.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/scan_perf.py --rows 1250000 --steps 60 --mode async > $O/trace_run.txt 2>&1
T=$(ls $O/trace/*/*kernel_trace.csv | head -1)
python3 $R/tools/timeline.py $T > $O/timeline.txt 2>&1 || true
python3 $R/tools/overlap_report.py $T > $O/overlap.txt 2>&1 || true
python3 $R/tools/trace_window.py $T > $O/window.txt 2>&1 || true
rm -rf $O/trace
cat $O/ticket_atomic.txt $O/shard_base.txt $O/overlap.txt
