import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.argv = ["bench.py"]
import torch, bench
dev = torch.device("cuda", 0)
for i in range(2):
    r = bench.c5_leg(dev)
    print({k: round(v.get("streaming_kernels_ms", v.get("staging_kernels_ms")), 3) for k, v in r.items() if isinstance(v, dict)})
