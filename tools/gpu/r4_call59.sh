set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c59
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_index_gpu.py tests/test_sharded_gpu.py -x -q -m gpu > $O/tests.txt 2>&1 || (tail -n 60 $O/tests.txt; exit 1)
grep -E "passed|failed" $O/tests.txt | tail -n 1
timeout -k 10 900 python3 bench.py --no-cpu --recall-queries 8 --no-facade > $O/bench.json 2> $O/bench.err || (tail -n 20 $O/bench.err; exit 1)
python3 tools/show_bench.py $O/bench.json 2>/dev/null | cut -c1-200 | head -n 12
