set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exchange
python3 -m pytest tests/test_sharded_gpu.py -x -q -m gpu > gpurun_out/exchange/tests.txt 2>&1 || { tail -n 40 gpurun_out/exchange/tests.txt; exit 1; }
tail -n 3 gpurun_out/exchange/tests.txt
for g in 1 2 4; do python3 bench.py --force-dist --rows 1250000 --exchange-group $g --no-legs --no-cpu --no-facade --recall-queries 0 --steps 120 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group', $g, d['config']['rows_total'], round(d['ms_per_step'],4), round(d['value']), d['roofline']['in_timed_region']['scan_to_scan_ms_median'])"; done
python3 bench.py --rows 1250000 --no-legs --no-cpu --no-facade --recall-queries 0 --steps 120 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no exchange', round(d['ms_per_step'],4))"
python3 bench.py --force-dist --no-legs --no-cpu --no-facade --recall-queries 8 --steps 60 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('10M dist', round(d['ms_per_step'],4), d['recall_at_k'])"
