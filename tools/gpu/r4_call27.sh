set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c27
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --force-dist --rows 1250000 --exchange-group 2 --no-legs --no-cpu --no-facade --recall-queries 0 --steps 150 > $O/run.json 2> $O/run.err
T=$(ls $O/trace/*/*kernel_trace.csv | head -1)
python3 $R/tools/boundary_report.py $T > $O/boundary.txt 2>&1 || true
python3 $R/tools/trace_window.py $T > $O/window.txt 2>&1 || true
rm -rf $O/trace
cat $O/boundary.txt
head -60 $O/window.txt
