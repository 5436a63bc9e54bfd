set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c33
mkdir -p $O
cd $R
for f in 0 2 0 2; do ANORAG_ENC_FOLD=$f python3 tools/enc_perf.py 2>&1 | grep -v amdgpu | tail -n 1; done > $O/enc_perf.txt
cat $O/enc_perf.txt
timeout -k 10 600 python3 -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "folded" > $O/tests.txt 2>&1 || (tail -n 40 $O/tests.txt; exit 1)
tail -n 3 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
for f in 0 2; do
  ANORAG_ENC_FOLD=$f rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_fold$f -- python3 $R/tools/enc_perf.py > $O/enc_fold$f.txt 2>&1
  python3 $R/tools/kstats.py $O/enc_fold$f > $O/enc_fold${f}_kstats.txt 2>&1 || true
  rm -rf $O/enc_fold$f
done
head -n 14 $O/enc_fold0_kstats.txt; head -n 14 $O/enc_fold2_kstats.txt
