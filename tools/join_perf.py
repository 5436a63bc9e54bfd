#!/usr/bin/env python3
"""Developer tool: time anr_index_self_join (semantic-similarity pairs) on synthetic clustered embeddings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd"))
import numpy as np
import torch
from anorag_hip import FlatIndex, METRIC_IP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.7
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
cent = torch.randn((max(1, n // 50), d), generator=g, device=dev)
idx = FlatIndex(d, METRIC_IP, normalize=True); idx.reserve(n)
done = 0
while done < n:
    m = min(262144, n - done)
    x = cent[torch.randint(0, cent.shape[0], (m,), generator=g, device=dev)] + 0.6 * torch.randn((m, d), generator=g, device=dev)
    torch.cuda.synchronize(); idx.add_device(x.data_ptr(), m); done += m
I, J, S = idx.self_join(thr)            # warm-up (also sizes the lists)
t0 = time.perf_counter(); I, J, S = idx.self_join(thr, cap_hint=len(I) + 1024, sort=False); dt = time.perf_counter() - t0
t0 = time.perf_counter(); idx.self_join(thr, cap_hint=len(I) + 1024); dts = time.perf_counter() - t0
flops = n * (n + 256) / 2 * 2 * d
print(f"n={n} d={d} thr={thr}: {len(I)} pairs, {dt*1e3:.1f} ms ({dts*1e3:.1f} ms with the host-side (i, j) sort), {flops/dt/1e12:.0f} TFLOP/s (upper triangle in 256x256 blocks), "
      f"numpy would need a {n*n*4/1e9:.1f} GB matrix")
