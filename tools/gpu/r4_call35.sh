R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c35
mkdir -p $O
cd $R
for lanes in 1 2; do ANORAG_ENC_LANES=$lanes timeout -k 10 300 python3 tools/shared_forward_perf.py 2>&1 | grep -E "lanes|threads"; done | tee $O/shared_forward.txt
