set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 -m pytest $R/tests/test_index_gpu.py -x -q -m gpu > $O/test_index.txt 2>&1 || (tail -30 $O/test_index.txt; exit 1)
tail -3 $O/test_index.txt
for cfg in "0 1 0" "0 0 0" "1 0 0" "1 0 2" "1 1 2" "1 0 1"; do set -- $cfg; python3 $R/tools/pipeline_log.py --rows 1250000 --steps 150 --schedule $1 --stream-wait $2 --shadow $3 --dump 4 2>&1 | grep -v amdgpu.ids; done > $O/pipeline.txt
cat $O/pipeline.txt
