set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c47
mkdir -p $O
cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu --durations=25 > $O/tests.txt 2>&1 || (tail -n 60 $O/tests.txt; exit 1)
grep -E "passed|failed" $O/tests.txt | tail -n 2
grep -E "^[0-9.]+s (call|setup)" $O/tests.txt | head -n 25
