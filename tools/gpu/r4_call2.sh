set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $R/tools/microbench/coresident > $O/coresident.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/scan_perf.py --rows 1250000 --steps 90 --mode async > $O/trace_run.txt 2>&1
T=$(ls $O/trace/*/*kernel_trace.csv | head -1)
python3 $R/tools/boundary_report.py $T > $O/boundary.txt 2>&1 || true
rm -rf $O/trace
cat $O/coresident.txt $O/boundary.txt
