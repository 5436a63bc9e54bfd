set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c43
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_index_gpu.py tests/test_fusion_sparse_gpu.py -x -q -m gpu -k "twelve or tight or sparse_rows_fuse or golden" > $O/tests.txt 2>&1 || (tail -n 40 $O/tests.txt; exit 1)
tail -n 2 $O/tests.txt
STEPS=12 bash tools/refresh_profiles.sh > $O/refresh12.log 2>&1 || (tail -n 30 $O/refresh12.log; exit 1)
tail -n 5 $O/refresh12.log
python3 tools/show_bench.py gpurun_out/prof/r04_bench_1gpu.json | head -n 8 | cut -c1-300
