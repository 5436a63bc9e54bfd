set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c34
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_encoder_gpu.py -x -q -m gpu -s -k "concurrent" > $O/tests2.txt 2>&1 || (grep -E "serial .* ms per query" $O/tests2.txt; tail -n 25 $O/tests2.txt; exit 1)
grep -E "serial .* ms per query|passed|failed" $O/tests2.txt | tail -n 5
ANORAG_ENC_LANES=1 timeout -k 10 600 python3 -m pytest tests/test_encoder_gpu.py -x -q -m gpu -s -k "concurrent" 2>&1 | grep -E "serial .* ms per query" || true
timeout -k 10 600 python3 tools/query_latency.py > $O/query_latency.txt 2>&1 || (tail -n 20 $O/query_latency.txt; exit 1)
grep -v "amdgpu\|Warning\|it/s" $O/query_latency.txt
timeout -k 10 900 python3 -m pytest tests/test_encoder_gpu.py tests/test_dropin_gpu.py -x -q -m gpu > $O/tests3.txt 2>&1 || (tail -n 30 $O/tests3.txt; exit 1)
tail -n 2 $O/tests3.txt
