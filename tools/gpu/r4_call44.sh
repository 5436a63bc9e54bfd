set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c44
mkdir -p $O
cd $R
STEPS=13 bash tools/refresh_profiles.sh > $O/refresh13.log 2>&1 || (tail -n 30 $O/refresh13.log; exit 1)
tail -n 3 $O/refresh13.log
python3 -c "
import json;d=json.load(open('gpurun_out/prof/r04_bench_1gpu.json'));r=d['roofline'];print(round(d['value']), d['ms_per_step'], r['frac'], r['traffic'], r['ms_per_launch'])"
cat gpurun_out/prof/r04_shard_size_lines.txt | cut -c1-250
