set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c9
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in 0 8 12; do echo "== ANORAG_SAMPLE_WAVES=$w"; ANORAG_SAMPLE_WAVES=$w python3 $R/tools/pipeline_log.py --rows 1250000 --steps 200 --schedule 0 --stream-wait 0 --shadow 0 2>&1 | grep -v amdgpu.ids | head -4; done > $O/pipeline.txt
echo "== shadow 2 schedule 0" >> $O/pipeline.txt
python3 $R/tools/pipeline_log.py --rows 1250000 --steps 200 --schedule 0 --stream-wait 0 --shadow 2 2>&1 | grep -v amdgpu.ids | head -12 >> $O/pipeline.txt
echo "== 1M rows" >> $O/pipeline.txt
for w in 0 12; do ANORAG_SAMPLE_WAVES=$w python3 $R/tools/pipeline_log.py --rows 1000000 --steps 200 --schedule 0 --stream-wait 0 --shadow 0 2>&1 | grep -v amdgpu.ids | head -4; done >> $O/pipeline.txt
echo "== 10M rows" >> $O/pipeline.txt
for w in 0 12; do ANORAG_SAMPLE_WAVES=$w python3 $R/tools/pipeline_log.py --rows 10000000 --steps 60 --schedule 0 --stream-wait 0 --shadow 0 2>&1 | grep -v amdgpu.ids | head -4; done >> $O/pipeline.txt
cat $O/pipeline.txt
