set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c37
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_fusion_sparse_gpu.py tests/test_fusion_dense_gpu.py -x -q -m gpu > $O/tests.txt 2>&1 || (tail -n 60 $O/tests.txt; exit 1)
tail -n 3 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bm25 -- python3 $R/tools/bm25_fuse_perf.py > $O/bm25_common.txt 2>&1 || (tail -n 20 $O/bm25_common.txt; exit 1)
grep -v amdgpu $O/bm25_common.txt | head -n 6 | cut -c1-420
find $O/prof -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} $O/bm25_kernel_stats.csv
head -n 16 $O/bm25_kernel_stats.csv | cut -c1-160
rm -rf $O/prof
