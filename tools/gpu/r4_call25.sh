set -e
cd $GRAFT_REPO_ROOT
python3 bench.py --force-dist --no-legs --no-cpu --no-facade --recall-queries 8 --steps 60 > gpurun_out/r4c25_dist.json 2> gpurun_out/r4c25_dist.err || (tail -20 gpurun_out/r4c25_dist.err; exit 1)
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4c25_dist.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['recall_at_k'], d['config']['parallelism'], d['roofline']['frac'], d['roofline']['in_timed_region'])
PY
for rows in 1250000 2500000 5000000; do python3 bench.py --force-dist --rows $rows --no-legs --no-cpu --no-facade --recall-queries 0 --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['rows_total'], round(d['ms_per_step'],4), round(d['value']))"; done
