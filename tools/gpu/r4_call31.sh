set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c31
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_fusion_sparse_gpu.py -x -q -m gpu -k "sliced or heavy or refuse" > $O/tests.txt 2>&1 || (tail -n 60 $O/tests.txt; exit 1)
tail -n 3 $O/tests.txt
timeout -k 10 300 python3 tools/bm25_fuse_perf.py > $O/bm25_common.txt 2>&1 || (tail -n 20 $O/bm25_common.txt; exit 1)
grep -v amdgpu $O/bm25_common.txt | head -n 10
QLO=1000 QHI=30000 timeout -k 10 300 python3 tools/bm25_fuse_perf.py > $O/bm25_rare.txt 2>&1 || (tail -n 20 $O/bm25_rare.txt; exit 1)
grep -v amdgpu $O/bm25_rare.txt | head -n 10
