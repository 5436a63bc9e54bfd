set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c40
mkdir -p $O
cd $R
for of in 256 320 384 512; do timeout -k 10 300 python3 tools/scan_perf.py --rows 10000000 --steps 40 --scan-bits 12 --overfetch $of 2>&1 | grep -E "^\[(sync|async)" | sed "s/^/of $of /" | tee -a $O/scan10m.txt; done
for of in 256 384; do timeout -k 10 300 python3 tools/scan_perf.py --rows 1250000 --steps 80 --scan-bits 12 --overfetch $of 2>&1 | grep -E "^\[(sync|async)" | sed "s/^/of $of /" | tee -a $O/scan1250k.txt; done
timeout -k 10 300 python3 tools/scan_perf.py --rows 1250000 --steps 80 --scan-bits 12 --clustered 2>&1 | grep -E "^\[(sync|async)" | sed "s/^/clustered 12 /" | tee -a $O/scan1250k.txt
timeout -k 10 300 python3 tools/scan_perf.py --rows 1250000 --steps 80 --scan-bits 16 --clustered 2>&1 | grep -E "^\[(sync|async)" | sed "s/^/clustered 16 /" | tee -a $O/scan1250k.txt
