set -e
cd $GRAFT_REPO_ROOT
STEPS=3 bash tools/refresh_profiles.sh
cat gpurun_out/prof/r04_shard_size_lines.txt
cat gpurun_out/prof/r04_pipeline_log_shard_1250k.txt | grep -E "^rows|scan end|scan start" 
