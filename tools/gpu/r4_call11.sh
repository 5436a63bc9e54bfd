set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c11
mkdir -p $O
cd $R
python3 -m pytest tests/test_fusion_sparse_gpu.py tests/test_sharded_gpu.py tests/test_index_gpu.py -x -q -m gpu > $O/tests.txt 2>&1 || (tail -40 $O/tests.txt; exit 1)
tail -3 $O/tests.txt
python3 bench.py > $O/bench.json 2> $O/bench.err || (tail -20 $O/bench.err; exit 1)
python3 tools/show_bench.py $O/bench.json
