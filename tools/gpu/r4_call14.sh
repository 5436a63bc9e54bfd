set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c14
mkdir -p $O
cd $R
python3 -m pytest tests/test_bm25_gpu.py tests/test_fusion_sparse_gpu.py tests/test_fusion_dense_gpu.py -x -q -m gpu > $O/tests1.txt 2>&1 || (tail -30 $O/tests1.txt; exit 1)
tail -2 $O/tests1.txt
python3 -m pytest tests/test_encoder_gpu.py -x -q -m gpu -s -k "concurrent" > $O/tests2.txt 2>&1 || (grep -E "serial .* ms per query" $O/tests2.txt; tail -5 $O/tests2.txt; exit 1)
grep -E "serial .* ms per query|passed|failed" $O/tests2.txt | tail -5
python3 -m pytest tests/test_dropin_gpu.py -x -q -m gpu > $O/tests3.txt 2>&1 || (tail -30 $O/tests3.txt; exit 1)
tail -2 $O/tests3.txt
python3 tools/query_latency.py > $O/query_latency.txt 2>&1 || (tail -20 $O/query_latency.txt; exit 1)
grep -v amdgpu $O/query_latency.txt
QLO=1000 QHI=30000 PROFILE=1 python3 tools/bm25_fuse_perf.py > $O/bm25_rare.txt 2>&1 || (tail -20 $O/bm25_rare.txt; exit 1)
grep -v amdgpu $O/bm25_rare.txt | head -60
python3 tools/bm25_fuse_perf.py > $O/bm25_common.txt 2>&1 || (tail -20 $O/bm25_common.txt; exit 1)
grep -v amdgpu $O/bm25_common.txt | head -10
