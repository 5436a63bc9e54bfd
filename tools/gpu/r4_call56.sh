set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_index_gpu.py -x -q -m gpu -k "twelve or tight" 2>&1 | tail -n 12
for rep in 1 2; do timeout -k 10 300 python3 tools/scan_perf.py --rows 1250000 --steps 400 --mode async --no-timing --clustered 2>&1 | grep -E "^\[async" | sed "s/^/clustered auto /"; done
timeout -k 10 300 python3 tools/scan_perf.py --rows 1250000 --steps 200 --mode async --no-timing 2>&1 | grep -E "^\[async" | sed "s/^/gaussian auto /"
timeout -k 10 300 python3 tools/scan_perf.py --rows 1250000 --steps 60 --mode async --no-timing --clustered 2>&1 | grep -E "^\[async" | sed "s/^/clustered auto 60 steps /"
SIGMA=0.02 timeout -k 10 300 python3 tools/scan_perf.py --rows 1250000 --steps 60 --mode async --no-timing --clustered 2>&1 | grep -E "^\[async" | sed "s/^/tight auto /"
