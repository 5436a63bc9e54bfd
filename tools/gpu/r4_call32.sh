set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c32
mkdir -p $O
cd $R
PROFILE=1 timeout -k 10 300 python3 tools/bm25_fuse_perf.py > $O/bm25_common.txt 2>&1 || (tail -n 20 $O/bm25_common.txt; exit 1)
grep -v amdgpu $O/bm25_common.txt | head -n 40
QLO=1000 QHI=30000 PROFILE=1 timeout -k 10 300 python3 tools/bm25_fuse_perf.py > $O/bm25_rare.txt 2>&1 || (tail -n 20 $O/bm25_rare.txt; exit 1)
grep -v amdgpu $O/bm25_rare.txt | head -n 40
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bm25 -- python3 $R/tools/bm25_fuse_perf.py > $O/prof.log 2>&1 || (tail -n 20 $O/prof.log; exit 1)
find $O/prof -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} $O/bm25_kernel_stats.csv
head -n 30 $O/bm25_kernel_stats.csv | cut -c1-200
