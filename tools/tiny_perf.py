#!/usr/bin/env python3
"""Developer tool: where the time of a batch-1 search on a tiny corpus goes (C1: 10 k x 384, top-10).
Compares the single-launch path (ANR_OPT_TINY = 1) with the five-kernel pipeline, through the Python wrapper and
through the bare ctypes call with preallocated buffers, beside numpy on the host."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd")); sys.path.insert(0, ROOT)
import numpy as np
try:
    from threadpoolctl import threadpool_limits
    threadpool_limits(int(os.environ.get("ANORAG_BENCH_THREADS", "16")))
except Exception:
    pass
from anorag_hip import FlatIndex, METRIC_IP, _lib
from anorag_hip._lib import OPT_TINY

n, d, k = int(os.environ.get("ROWS", 10_000)), int(os.environ.get("DIM", 384)), int(os.environ.get("K", 10))
x = np.random.default_rng(1234).standard_normal((n, d), dtype=np.float32)
q = np.random.default_rng(4321).standard_normal((512, d), dtype=np.float32)
idx = FlatIndex(d, METRIC_IP, normalize=True); idx.add(x)
lib = _lib.load()
D = np.empty((1, k), dtype=np.float32); I = np.empty((1, k), dtype=np.int64)
def wrapper(i): return idx.search(q[i:i + 1], k)
def bare(i):
    lib.anr_index_search(idx._h, C.c_void_p(q.ctypes.data + i * d * 4), 1, k, C.c_void_p(D.ctypes.data), C.c_void_p(I.ctypes.data))
def timeit(f, reps=400):
    for i in range(20): f(i)
    t0 = time.perf_counter()
    for i in range(reps): f(i % 512)
    return (time.perf_counter() - t0) / reps * 1e6
res = {}
# the first 400 calls of a process, timed on their own (rounds 1-2: 131-171 us per call — two stalls of ~1 ms and ~38 ms
# in calls 188 and 317 that were Python's collector making its first full passes, tools/first_calls.py; FlatIndex now
# runs one full collection when the first index of a process is created)
idx.set_option(OPT_TINY, int(os.environ.get("TINY_MODE", "1")))  # 2 = wherever the path is able
t0 = time.perf_counter()
for i in range(400): wrapper(i % 512)
res["first 400 calls of the process, wrapper_us"] = (time.perf_counter() - t0) / 400 * 1e6
for i in range(1200): bare(i % 512)
for tiny in (1, 0):
    idx.set_option(OPT_TINY, int(os.environ.get("TINY_MODE", "1")) if tiny else 0)
    res[f"tiny={tiny} wrapper_us"] = timeit(wrapper)
    res[f"tiny={tiny} bare_ctypes_us"] = timeit(bare)
xn = x / np.linalg.norm(x, axis=1, keepdims=True)
def cpu(i):
    qq = q[i:i + 1]; qq = qq / np.linalg.norm(qq, axis=1, keepdims=True); s = qq @ xn.T
    part = np.argpartition(-s, k - 1, axis=1)[:, :k]; np.argsort(-np.take_along_axis(s, part, 1), axis=1)
res["numpy_us"] = timeit(cpu)
def noop(i): lib.anr_index_ntotal(idx._h)
res["ctypes_noop_us"] = timeit(noop)
for key, v in res.items(): print(f"{key:44s} {v:8.2f}")
