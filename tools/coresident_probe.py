#!/usr/bin/env python3
"""Developer tool: do small kernels run BESIDE the scan's resident workgroups?  Stream A runs searches back to back
(asynchronously, 3 in flight); stream B launches a tiny elementwise kernel every ~50 us and records how long each
took from launch to completion (events).  If the small kernel co-resides, its latency stays ~10 us; if it has to
wait for a scan to retire, it shows ~a scan time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd"))
import torch
from anorag_hip import FlatIndex, METRIC_IP
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
dev = torch.device("cuda", 0)
idx = FlatIndex(768, METRIC_IP, normalize=True); idx.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
done = 0
while done < rows:
    m = min(262144, rows - done)
    x = torch.randn((m, 768), generator=g, device=dev); torch.cuda.synchronize(); idx.add_device(x.data_ptr(), m); done += m
Q = torch.randn((200, 64, 768), generator=g, device=dev)
S = [torch.cuda.Stream() for _ in range(3)]
Ds = [torch.empty((64, 100), device=dev) for _ in range(3)]; Is = [torch.empty((64, 100), device=dev, dtype=torch.int64) for _ in range(3)]
side = torch.cuda.Stream()
y = torch.zeros((64, 100), device=dev)
# warm up
for i in range(6): idx.search_device_async(Q[i].data_ptr(), 64, 100, Ds[i % 3].data_ptr(), Is[i % 3].data_ptr(), S[i % 3].cuda_stream)
idx.sync(); torch.cuda.synchronize()
with torch.cuda.stream(side):
    y.add_(1.0)
torch.cuda.synchronize()
evs = []
t0 = time.perf_counter()
for i in range(6, 200):
    idx.search_device_async(Q[i].data_ptr(), 64, 100, Ds[i % 3].data_ptr(), Is[i % 3].data_ptr(), S[i % 3].cuda_stream)
    if i % 2 == 0 and i > 20:
        with torch.cuda.stream(side):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); y.add_(1.0); b.record()
            evs.append((a, b, time.perf_counter()))
idx.sync(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 194
lat = sorted(a.elapsed_time(b) * 1e3 for a, b, _ in evs)
print(f"rows={rows}: {dt*1e3:.3f} ms per search step; tiny kernel on a side stream: GPU time between its two events "
      f"median {lat[len(lat)//2]:.1f} us, p10 {lat[len(lat)//10]:.1f}, p90 {lat[len(lat)*9//10]:.1f}, max {lat[-1]:.1f} (n={len(lat)})")
