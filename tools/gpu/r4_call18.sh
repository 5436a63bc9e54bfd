set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c18
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bm25 -- python3 $R/tools/bm25_fuse_perf.py > $O/bm25_common.txt 2>&1
python3 $R/tools/kstats.py $O/bm25 > $O/bm25_kstats.txt 2>&1 || true
rm -rf $O/bm25
head -16 $O/bm25_kstats.txt
QLO=1000 QHI=30000 PROFILE=1 python3 $R/tools/bm25_fuse_perf.py > $O/bm25_rare_profile.txt 2>&1
grep -v amdgpu $O/bm25_rare_profile.txt | tail -n 30
