#!/usr/bin/env python3
"""Developer tool: the pipelined search loop WITHOUT a profiler, explained by the library's own batch log
(anr_index_batch_log: host enqueue / retire times and the device clock stamps the batch's kernels leave).
  python tools/pipeline_log.py [--rows 1250000] [--steps 120] [--shadow 0|1|2] [--clustered] [--dump N]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd"))
import numpy as np
import torch
from anorag_hip import FlatIndex, METRIC_IP
from anorag_hip._lib import OPT_TIMING, OPT_SHADOW, OPT_SCHEDULE, OPT_STREAM_WAIT

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_250_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--steps", type=int, default=120)
ap.add_argument("--shadow", type=int, default=0)
ap.add_argument("--timing", type=int, default=0)
ap.add_argument("--schedule", type=int, default=0)
ap.add_argument("--stream-wait", type=int, default=0)
ap.add_argument("--clustered", action="store_true")
ap.add_argument("--sigma", type=float, default=0.3)
ap.add_argument("--dump", type=int, default=0, help="print the last N records relative to the first one's scan start")
a = ap.parse_args()
dev = torch.device("cuda", 0)
idx = FlatIndex(a.dim, METRIC_IP, normalize=True)
idx.reserve(a.rows)
g = torch.Generator(device=dev); g.manual_seed(1)
done = 0
while done < a.rows:
    m = min(262144, a.rows - done)
    x = torch.randn((m, a.dim), generator=g, device=dev)
    if a.clustered:
        if done == 0:
            cent = torch.randn((1024, a.dim), generator=g, device=dev)
        x = cent[torch.randint(0, 1024, (m,), generator=g, device=dev)] + a.sigma * x
    torch.cuda.synchronize(); idx.add_device(x.data_ptr(), m); done += m
idx.set_option(OPT_TIMING, a.timing)
idx.set_option(OPT_SHADOW, a.shadow)
idx.set_option(OPT_SCHEDULE, a.schedule)
idx.set_option(OPT_STREAM_WAIT, a.stream_wait)
Q = torch.randn((a.steps + 8, a.batch, a.dim), generator=g, device=dev)
if a.clustered:
    Q = cent[torch.randint(0, 1024, (a.steps + 8, a.batch), generator=g, device=dev)] + a.sigma * Q
NS = 3
S = [torch.cuda.Stream() for _ in range(NS)]
D = [torch.empty((a.batch, a.k), device=dev) for _ in range(NS)]
I = [torch.empty((a.batch, a.k), device=dev, dtype=torch.int64) for _ in range(NS)]
def run(lo, hi):
    t0 = time.perf_counter()
    for i in range(lo, hi):
        idx.search_device_async(Q[i].data_ptr(), a.batch, a.k, D[i % NS].data_ptr(), I[i % NS].data_ptr(), S[i % NS].cuda_stream)
    idx.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (hi - lo)
run(0, 8)
idx.reset_stats()
dt = run(8, a.steps + 8)
st = idx.last_stats()
rec, off = idx.batch_log(a.steps)
rec = rec[4:]  # the first batches of the burst start from an idle device
rec = rec[np.argsort(rec[:, 7], kind='stable')]  # streams are independent: device order, not submission order
n = rec.shape[0]
us = lambda v: v / 1e3
first, last, end = rec[:, 7], rec[:, 8], rec[:, 9]
per = us(np.diff(first)); gap = us(first[1:] - end[:-1]); dur = us(end - first)
q = lambda v, p: float(np.percentile(v, p))
print(f"rows={a.rows} schedule={a.schedule} stream_wait={a.stream_wait} shadow={a.shadow}{' clustered sigma=' + str(a.sigma) if a.clustered else ''}: wall {dt*1e3:.4f} ms per batch ({a.batch/dt:.0f} q/s), "
      f"fallback={st['n_fallback']} from_lists={st['n_from_lists']} dense_exact={st['n_dense_exact']} overfetch={st['overfetch']} cand/q={st['n_candidates']/a.steps/a.batch:.0f}; "
      f"shadow batches {int((rec[:, 12] & 1).sum())}/{n}")
print(f"  main scan (first workgroup start -> end) us: median {q(dur,50):.1f} p10 {q(dur,10):.1f} p90 {q(dur,90):.1f}; placed (first -> last workgroup start) median {q(us(last-first),50):.1f}")
print(f"  scan start -> next scan start us: median {q(per,50):.1f} mean {per.mean():.1f} p90 {q(per,90):.1f} max {per.max():.1f}")
print(f"  scan end   -> next scan start us: median {q(gap,50):.1f} mean {gap.mean():.1f} p90 {q(gap,90):.1f} max {gap.max():.1f}")
# where each side kernel of batch i ends relative to the END of the scan in front of it (batch i-1's)
prev_end = end[:-1]
for name, col in (("prep end", 4), ("sample end", 5), ("ladder end", 6), ("own scan first wg", 7)):
    v = us(rec[1:, col] - prev_end)
    print(f"  {name:18s} relative to the previous scan's end us: median {q(v,50):8.1f} p10 {q(v,10):8.1f} p90 {q(v,90):8.1f}")
for name, col in (("select end", 10), ("post end", 11)):
    v = us(rec[:, col] - end)
    print(f"  {name:18s} relative to the batch's own scan end us: median {q(v,50):8.1f} p10 {q(v,10):8.1f} p90 {q(v,90):8.1f}")
henq = us(rec[:, 2] - rec[:, 1])
lead = us(first - (rec[:, 2] + off))  # how long before its scan started the batch was fully enqueued
ret = us((rec[:, 3] + off) - rec[:, 11])
print(f"  host: enqueue call us median {q(henq,50):.1f} p90 {q(henq,90):.1f} max {henq.max():.1f}; enqueued before its scan started by us median {q(lead,50):.1f} p10 {q(lead,10):.1f} min {lead.min():.1f}; "
      f"retired after its last kernel by us median {q(ret,50):.1f}")
hgap = us(np.diff(rec[:, 1]))
print(f"  host: enqueue -> next enqueue us median {q(hgap,50):.1f} p90 {q(hgap,90):.1f} max {hgap.max():.1f}")
if a.dump:
    r = rec[-a.dump - 8:-8]
    t0 = r[0, 7]
    for row in r:
        h = [(row[c] + off - t0) / 1e3 for c in (1, 2, 3)]
        d = [(row[c] - t0) / 1e3 if row[c] else float('nan') for c in range(4, 12)]
        print(f"  seq {row[0]:4d} host enq {h[0]:8.1f}..{h[1]:8.1f} retire {h[2]:8.1f} | prep {d[0]:8.1f} sample {d[1]:8.1f} ladder {d[2]:8.1f} scan {d[3]:8.1f}/{d[4]:8.1f}..{d[5]:8.1f} select {d[6]:8.1f} post {d[7]:8.1f}")
