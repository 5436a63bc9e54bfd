set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c21
mkdir -p $O
cd $R
python3 tools/pipeline_log.py --rows 1250000 --steps 90 --clustered --sigma 0.02 --dump 5 2>&1 | grep -v amdgpu | head -24 > $O/tight.txt
cat $O/tight.txt
python3 tools/pipeline_log.py --rows 1250000 --steps 200 2>&1 | grep -v amdgpu | head -4
