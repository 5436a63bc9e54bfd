set -e
R=$GRAFT_REPO_ROOT
cd $R
{ python3 tools/bm25_fuse_perf.py; echo "--- rare-term queries (vocabulary ranks 1000..30000)"; QLO=1000 QHI=30000 python3 tools/bm25_fuse_perf.py; } 2>&1 | grep -v amdgpu > gpurun_out/r04_bm25_fuse_pipeline_lines.txt
cut -c1-150 gpurun_out/r04_bm25_fuse_pipeline_lines.txt; grep -o "median of 7 calls [0-9.]* ms): [0-9.]* ms" gpurun_out/r04_bm25_fuse_pipeline_lines.txt
