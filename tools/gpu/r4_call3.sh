set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $R/tools/microbench/coresident > $O/coresident.txt 2>&1
grep k_prepq $O/coresident.txt
