set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c54
mkdir -p $O
cd $R
STEPS=15 bash tools/refresh_profiles.sh > $O/refresh15.log 2>&1 || (tail -n 30 $O/refresh15.log; exit 1)
python3 -c "
import json;d=json.load(open('gpurun_out/prof/r04_bench_1gpu.json'));r=d['roofline'];print(round(d['value']), d['ms_per_step'], r['frac'], r['traffic'], r['ms_per_launch'], d['recall_at_k'], d['same_results_from_the_f16_image'])"
cat gpurun_out/prof/r04_fuse_dense_c5.txt | cut -c1-220
timeout -k 10 300 python3 bench.py --scan-bits 16 --no-legs --no-cpu --no-facade --recall-queries 8 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scan-bits 16:', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['bytes_per_stored_value'], d['roofline']['traffic_note'][:120])" | tee $O/bits16.txt
