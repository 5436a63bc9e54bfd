set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/suite
mkdir -p $O
cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || (tail -n 60 $O/tests.txt; exit 1)
grep -E "passed|failed" $O/tests.txt | tail -n 1
STEPS=13 bash tools/refresh_profiles.sh > $O/refresh13.log 2>&1 || (tail -n 30 $O/refresh13.log; exit 1)
python3 tools/show_bench.py gpurun_out/prof/r04_bench_1gpu.json 2>/dev/null | cut -c1-200 | head -n 11
cat gpurun_out/prof/r04_shard_size_lines.txt | cut -c1-220
