set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c39
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_index_gpu.py -x -q -m gpu -k "twelve" > $O/tests12.txt 2>&1 || (tail -n 60 $O/tests12.txt; exit 1)
tail -n 3 $O/tests12.txt
for bits in 16 12; do timeout -k 10 300 python3 tools/scan_perf.py --rows 10000000 --steps 30 --scan-bits $bits 2>&1 | grep -E "^\[(sync|async)" | sed "s/^/bits $bits /" | tee -a $O/scan10m.txt; done
for bits in 16 12; do timeout -k 10 300 python3 tools/scan_perf.py --rows 1250000 --steps 60 --scan-bits $bits 2>&1 | grep -E "^\[(sync|async)" | sed "s/^/bits $bits /" | tee -a $O/scan1250k.txt; done
