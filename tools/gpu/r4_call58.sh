set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c58
mkdir -p $O
cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || (tail -n 60 $O/tests.txt; exit 1)
grep -E "passed|failed" $O/tests.txt | tail -n 1
timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err || (tail -n 20 $O/bench.err; exit 1)
python3 tools/show_bench.py $O/bench.json 2>/dev/null | cut -c1-330 | head -n 12
