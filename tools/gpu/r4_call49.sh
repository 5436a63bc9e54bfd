set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c49
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_fusion_dense_gpu.py -x -q -m gpu > $O/tests.txt 2>&1 || (tail -n 40 $O/tests.txt; exit 1)
tail -n 2 $O/tests.txt
timeout -k 10 300 python3 tools/fuse_dense_perf.py 2>&1 | grep "queries x" | tee $O/fuse.txt | cut -c1-260
NZ=10000 timeout -k 10 300 python3 tools/fuse_dense_perf.py 2>&1 | grep "queries x" | sed "s/^/NZ=10000 /" | cut -c1-260
