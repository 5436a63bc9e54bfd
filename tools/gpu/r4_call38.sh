set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c38
mkdir -p $O
cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || (tail -n 60 $O/tests.txt; exit 1)
tail -n 3 $O/tests.txt
