set -e
cd $GRAFT_REPO_ROOT
STEPS=45 bash tools/refresh_profiles.sh
cat gpurun_out/prof/r04_encoder_forward_lines.txt
cat gpurun_out/prof/r04_fuse_dense_c5.txt
cat gpurun_out/prof/r04_bm25_fuse_pipeline_lines.txt
