set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c20
mkdir -p $O
cd $R
python3 -m pytest tests/test_index_gpu.py tests/test_sharded_gpu.py -x -q -m gpu > $O/tests.txt 2>&1 || (tail -40 $O/tests.txt; exit 1)
tail -2 $O/tests.txt
python3 tools/pipeline_log.py --rows 1250000 --steps 90 --clustered --sigma 0.02 2>&1 | grep -v amdgpu | head -16 > $O/tight.txt
cat $O/tight.txt
python3 tools/pipeline_log.py --rows 1250000 --steps 200 2>&1 | grep -v amdgpu | head -5
