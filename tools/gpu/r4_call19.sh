set -e
cd $GRAFT_REPO_ROOT
STEPS=12 bash tools/refresh_profiles.sh
python3 tools/show_bench.py gpurun_out/prof/r04_bench_1gpu.json
cat gpurun_out/prof/r04_pmc_traffic_k_scan.json | head -20
