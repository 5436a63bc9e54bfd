set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c16
mkdir -p $O
cd $R
python3 -m pytest tests/test_encoder_gpu.py -x -q -m gpu > $O/tests_enc.txt 2>&1 || (tail -40 $O/tests_enc.txt; exit 1)
tail -2 $O/tests_enc.txt
for f in 1 0; do ANORAG_ENC_FOLD=$f python3 tools/enc_perf.py 2>&1 | grep -v amdgpu | tail -3; done > $O/enc_perf.txt
cat $O/enc_perf.txt
QLO=1000 QHI=30000 python3 tools/bm25_fuse_perf.py > $O/bm25_rare.txt 2>&1 || (tail -20 $O/bm25_rare.txt; exit 1)
grep -v amdgpu $O/bm25_rare.txt | head -8
python3 tools/bm25_fuse_perf.py > $O/bm25_common.txt 2>&1 || (tail -20 $O/bm25_common.txt; exit 1)
grep -v amdgpu $O/bm25_common.txt | head -8
