set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --one-device --rows 2500000 --steps 20 --warmup 5 --no-cpu --recall-queries 8 2>gpurun_out/r4c50.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['n_gpus'], round(d['value']), d['ms_per_step'], d['recall_at_k'], d['config']['parallelism'], d['scaling'])"
