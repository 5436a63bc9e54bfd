set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c42
mkdir -p $O
cd $R
timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err || (tail -n 20 $O/bench.err; exit 1)
python3 tools/show_bench.py $O/bench.json
