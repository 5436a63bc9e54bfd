set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c17
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  ANORAG_ENC_FOLD=$f rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_fold$f -- python3 $R/tools/enc_perf.py > $O/enc_fold$f.txt 2>&1
  python3 $R/tools/kstats.py $O/enc_fold$f > $O/enc_fold${f}_kstats.txt 2>&1 || true
  rm -rf $O/enc_fold$f
done
tail -2 $O/enc_fold1.txt $O/enc_fold0.txt
head -14 $O/enc_fold1_kstats.txt; head -14 $O/enc_fold0_kstats.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bm25 -- python3 $R/tools/bm25_fuse_perf.py > $O/bm25_common.txt 2>&1
python3 $R/tools/kstats.py $O/bm25 > $O/bm25_kstats.txt 2>&1 || true
rm -rf $O/bm25
head -16 $O/bm25_kstats.txt
QLO=1000 QHI=30000 PROFILE=1 python3 $R/tools/bm25_fuse_perf.py 2>&1 | grep -v amdgpu | tail -28 > $O/bm25_rare_profile.txt
cat $O/bm25_rare_profile.txt
