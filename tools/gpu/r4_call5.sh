set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for hq in 4 8 16; do for sh in 0 2; do echo "== GPU_MAX_HW_QUEUES=$hq"; GPU_MAX_HW_QUEUES=$hq python3 $R/tools/pipeline_log.py --rows 1250000 --steps 150 --shadow $sh --dump 5 2>&1 | grep -v amdgpu.ids; done; done > $O/pipeline.txt
cat $O/pipeline.txt
