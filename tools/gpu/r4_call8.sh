set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c8
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/pipeline_log.py --rows 10000000 --steps 40 --schedule 1 --stream-wait 0 --shadow 2 --dump 6 2>&1 | grep -v amdgpu.ids > $O/pipeline10m.txt
cat $O/pipeline10m.txt
