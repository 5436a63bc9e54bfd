set -e
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -x -q -m gpu > gpurun_out/r4c24_tests.txt 2>&1 || (tail -40 gpurun_out/r4c24_tests.txt; exit 1)
tail -2 gpurun_out/r4c24_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
STEPS=1 bash tools/refresh_profiles.sh
python3 tools/show_bench.py gpurun_out/prof/r04_bench_1gpu.json
head -4 gpurun_out/prof/r04_bench_10m_1gpu_in_flight_1_kernel_stats.csv
python3 tools/show_bench.py gpurun_out/prof/bench_prof_serial.json 2>/dev/null | head -1 || true
