set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c45
mkdir -p $O
cd $R
STEPS=45 bash tools/refresh_profiles.sh > $O/refresh45.log 2>&1 || (tail -n 30 $O/refresh45.log; exit 1)
tail -n 3 $O/refresh45.log
cat gpurun_out/prof/r04_shared_forward_lines.txt gpurun_out/prof/r04_query_latency.txt gpurun_out/prof/r04_encoder_forward_lines.txt | cut -c1-250
cut -c1-330 gpurun_out/prof/r04_bm25_fuse_pipeline_lines.txt
