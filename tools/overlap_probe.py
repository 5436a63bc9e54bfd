#!/usr/bin/env python3
"""Developer probe: do small kernels on another stream run while k_scan is resident?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd"))
import torch
from anorag_hip import FlatIndex, METRIC_IP
from anorag_hip._lib import OPT_RESERVE_CUS
dev = torch.device("cuda", 0)
rows = 4_000_000
idx = FlatIndex(768, METRIC_IP, normalize=True); idx.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
done = 0
while done < rows:
    x = torch.randn((262144, 768), generator=g, device=dev); torch.cuda.synchronize()
    idx.add_device(x.data_ptr(), min(262144, rows - done)); done += 262144
from anorag_hip._lib import OPT_TIMING
idx.set_option(OPT_TIMING, 1)
idx.set_option(OPT_RESERVE_CUS, int(sys.argv[1]) if len(sys.argv) > 1 else 16)
Q = torch.randn((8, 64, 768), generator=g, device=dev)
D = [torch.empty((64, 100), device=dev) for _ in range(3)]; I = [torch.empty((64, 100), device=dev, dtype=torch.int64) for _ in range(3)]
ss = [torch.cuda.Stream() for _ in range(3)]
probe = torch.cuda.Stream()
y = torch.zeros(1 << 16, device=dev)
for i in range(3):
    idx.search_device_async(Q[i].data_ptr(), 64, 100, D[i].data_ptr(), I[i].data_ptr(), ss[i].cuda_stream)
with torch.cuda.stream(probe):
    for j in range(5): y.add_(1.0)
idx.sync(); torch.cuda.synchronize(); idx.reset_stats()
# one long scan (~1.1 ms at 4M rows), probe kernels issued while it runs
evs = []
t0 = torch.cuda.Event(enable_timing=True); t0.record(probe)
idx.search_device_async(Q[3].data_ptr(), 64, 100, D[0].data_ptr(), I[0].data_ptr(), ss[0].cuda_stream)
with torch.cuda.stream(probe):
    for j in range(10):
        y.add_(1.0)
        e = torch.cuda.Event(enable_timing=True); e.record(probe); evs.append(e)
idx.sync(); torch.cuda.synchronize()
print("probe completion times (ms after t0):", [round(t0.elapsed_time(e), 3) for e in evs])
print("scan ms:", idx.last_stats()["scan_ms"])
