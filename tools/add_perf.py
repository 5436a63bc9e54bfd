#!/usr/bin/env python3
"""Developer tool: index build rate from host memory (anr_index_add) and from device memory (anr_index_add_dev)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd"))
import numpy as np
import torch
from anorag_hip import FlatIndex, METRIC_IP
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 768
x = np.random.default_rng(0).standard_normal((n, d), dtype=np.float32)
idx = FlatIndex(d, METRIC_IP, normalize=True); idx.reserve(n)
idx.add(x[:1000]); idx.reset()
t0 = time.perf_counter(); idx.add(x); dt = time.perf_counter() - t0
print(f"host add: {n} x {d}: {dt*1e3:.0f} ms = {x.nbytes/dt/1e9:.1f} GB/s of float32 input, {n/dt/1e6:.2f} M rows/s")
idx.reset()
xt = torch.from_numpy(x[: n // 2]).cuda(); torch.cuda.synchronize()
t0 = time.perf_counter(); idx.add_device(xt.data_ptr(), xt.shape[0]); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"device add: {xt.shape[0]} x {d}: {dt*1e3:.1f} ms = {xt.numel()*4/dt/1e9:.0f} GB/s of float32 input")
