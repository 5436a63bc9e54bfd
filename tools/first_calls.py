#!/usr/bin/env python3
"""Developer tool: per-call latency of the first searches of a process (batch 1, 10 k x 384, top-10): which calls are
slow, and whether it is the library or the interpreter (NOGC=1 switches Python's collector off)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd"))
import numpy as np
import gc
if os.environ.get("NOGC"): gc.disable()
from anorag_hip import FlatIndex, METRIC_IP
n, d, k = 10000, 384, 10
x = np.random.default_rng(1234).standard_normal((n, d), dtype=np.float32)
q = np.random.default_rng(4321).standard_normal((512, d), dtype=np.float32)
t0 = time.perf_counter(); idx = FlatIndex(d, METRIC_IP, normalize=True); idx.add(x); t1 = time.perf_counter()
print("create+add ms", (t1 - t0) * 1e3)
ts = []
for i in range(1000):
    a = time.perf_counter(); idx.search(q[i % 512:i % 512 + 1], k); ts.append((time.perf_counter() - a) * 1e6)
print("first 12:", [round(t) for t in ts[:12]])
for lo, hi in ((12, 50), (50, 100), (100, 200), (200, 400), (400, 700), (700, 1000)):
    print(lo, hi, "mean us", round(sum(ts[lo:hi]) / (hi - lo), 1), "max", round(max(ts[lo:hi])))
big = [(i, round(t)) for i, t in enumerate(ts) if t > 150]
print("calls over 150 us:", big[:20])
ts2 = []
for i in range(20000):
    a = time.perf_counter(); idx.search(q[i % 512:i % 512 + 1], k); ts2.append((time.perf_counter() - a) * 1e6)
print("next 20000: mean", round(sum(ts2) / len(ts2), 1), "calls over 150 us:", [(i, round(t)) for i, t in enumerate(ts2) if t > 150][:20])
# a second index in the same process: are the stalls per process or per stream?
idx2 = FlatIndex(d, METRIC_IP, normalize=True); idx2.add(x[:5000])
ts3 = []
for i in range(1000):
    a = time.perf_counter(); idx2.search(q[i % 512:i % 512 + 1], k); ts3.append((time.perf_counter() - a) * 1e6)
print("second index: first call", round(ts3[0]), "mean of the rest", round(sum(ts3[1:]) / 999, 1), "calls over 150 us:", [(i, round(t)) for i, t in enumerate(ts3) if t > 150][:10])
