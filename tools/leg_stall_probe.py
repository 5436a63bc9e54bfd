#!/usr/bin/env python3
"""Developer tool: does the shard-size pipeline stall once in a while?  Per-batch host timestamps of the 3-deep asynchronous
loop at 1.25 M rows, repeated; prints every gap over 5 ms (round 3: a one-off ~65 ms stall hit bench.py's shard leg in 2 of
9 runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd")); sys.path.insert(0, ROOT)
import gc
import torch
from anorag_hip import FlatIndex, METRIC_IP
gc.collect(); gc.disable()
dev = torch.device("cuda", 0)
rows, dim, batch, k = 1_250_000, 768, 64, 100
big = None
if os.environ.get("WITH_BIG"):   # the bench's situation: a 10 M-row index alive beside this one
    big = FlatIndex(dim, METRIC_IP, normalize=True); big.reserve(10_000_000)
    g0 = torch.Generator(device=dev); g0.manual_seed(1)
    for _ in range(38):
        x = torch.randn((262144, dim), generator=g0, device=dev); torch.cuda.synchronize(); big.add_device(x.data_ptr(), x.shape[0])
    del x
for rep in range(int(os.environ.get("REPS", 6))):
    idx = FlatIndex(dim, METRIC_IP, normalize=True); idx.reserve(rows)
    g = torch.Generator(device=dev); g.manual_seed(11 + rep)
    done = 0
    while done < rows:
        m = min(262144, rows - done)
        x = torch.randn((m, dim), generator=g, device=dev); torch.cuda.synchronize(); idx.add_device(x.data_ptr(), m); done += m
    del x
    Q = torch.randn((90, batch, dim), generator=g, device=dev)
    S = [torch.cuda.Stream(device=dev) for _ in range(3)]
    D = [torch.empty((batch, k), device=dev) for _ in range(3)]; I = [torch.empty((batch, k), device=dev, dtype=torch.int64) for _ in range(3)]
    ts = []
    for i in range(90):
        t0 = time.perf_counter()
        idx.search_device_async(Q[i].data_ptr(), batch, k, D[i % 3].data_ptr(), I[i % 3].data_ptr(), S[i % 3].cuda_stream)
        ts.append(time.perf_counter() - t0)
    idx.sync(); torch.cuda.synchronize()
    slow = [(i, round(t * 1e3, 1)) for i, t in enumerate(ts) if t > 5e-3]
    print(f"rep {rep}: mean call {1e3 * sum(ts[10:]) / 80:.3f} ms; calls over 5 ms: {slow}", flush=True)
    idx.close()
