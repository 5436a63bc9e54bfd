#!/usr/bin/env python3
"""Developer tool: time the search pipeline on synthetic data (not part of the product or the tests)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd"))
import torch
from anorag_hip import FlatIndex, METRIC_IP
from anorag_hip._lib import OPT_TIMING, OPT_SAMPLE_ROWS, OPT_OVERFETCH

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--sample", type=int, default=0)
ap.add_argument("--overfetch", type=int, default=0)
ap.add_argument("--streams", type=int, default=0)
ap.add_argument("--no-timing", action="store_true", help="do not record the per-scan HIP events (ANR_OPT_TIMING)")
ap.add_argument("--mode", default="both")
ap.add_argument("--scan-bits", type=int, default=0)
ap.add_argument("--fused-post", type=int, default=-1, help="ANR_OPT_FUSED_POST: 0 three launches, 1 automatic (default), 2 always fused")
ap.add_argument("--clustered", action="store_true", help="1024 Gaussian centroids, sigma 0.3 (SURVEY 8d)")
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
        x = cent[torch.randint(0, 1024, (m,), generator=g, device=dev)] + float(os.environ.get("SIGMA", "0.3")) * x
    torch.cuda.synchronize()
    idx.add_device(x.data_ptr(), m)
    done += m
if a.scan_bits != 0:
    from anorag_hip._lib import OPT_SCAN_BITS
    idx.set_option(OPT_SCAN_BITS, a.scan_bits)
if a.fused_post >= 0:
    from anorag_hip._lib import OPT_FUSED_POST
    idx.set_option(OPT_FUSED_POST, a.fused_post)
idx.set_option(OPT_TIMING, 0 if a.no_timing else 1)
if a.sample: idx.set_option(OPT_SAMPLE_ROWS, a.sample)
if a.overfetch: idx.set_option(OPT_OVERFETCH, a.overfetch)
Q = torch.randn((a.steps + 2, a.batch, a.dim), generator=g, device=dev)
if a.clustered:
    Q = cent[torch.randint(0, 1024, (a.steps + 2, a.batch), generator=g, device=dev)] + 0.3 * Q
D = torch.empty((a.batch, a.k), device=dev); I = torch.empty((a.batch, a.k), device=dev, dtype=torch.int64)
for i in range(2):
    idx.search_device(Q[i].data_ptr(), a.batch, a.k, D.data_ptr(), I.data_ptr())
torch.cuda.synchronize()
tot_scan = tot_all = 0.0; cand = fb = 0
t0 = time.perf_counter()
for i in range(2, a.steps + 2):
    idx.search_device(Q[i].data_ptr(), a.batch, a.k, D.data_ptr(), I.data_ptr())
    st = idx.last_stats()
    tot_scan += st["scan_ms"]; tot_all += st["total_ms"]; cand += st["n_candidates"]; fb += st["n_fallback"]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
byt = st["scan_bytes"]
if a.mode in ("both", "sync"):
  print(f"[sync ] rows={a.rows} wall/step={dt*1e3:.3f} ms  scan={tot_scan/a.steps:.3f} ms  gpu-total={tot_all/a.steps:.3f} ms  "
      f"scan GB/s={byt/1e9/(tot_scan/a.steps/1e3):.0f}  qps={a.batch/dt:.0f}  cand/q={cand/a.steps/a.batch:.0f} fallback={fb} "
      f"sample={st['sample_rows']} overfetch={st['overfetch']}")
if a.mode in ("both", "async"):
    from anorag_hip._lib import OPT_STREAMS
    if a.streams: idx.set_option(OPT_STREAMS, a.streams)
    NS = 3
    strs = [torch.cuda.Stream() for _ in range(NS)]
    Ds = [torch.empty_like(D) for _ in range(NS)]; Is = [torch.empty_like(I) for _ in range(NS)]
    for rep in range(2):
        idx.sync(); idx.reset_stats(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2, a.steps + 2):
            s = i % NS
            idx.search_device_async(Q[i].data_ptr(), a.batch, a.k, Ds[s].data_ptr(), Is[s].data_ptr(), strs[s].cuda_stream)
        idx.sync(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
    st = idx.last_stats()
    print(f"[async] rows={a.rows} wall/step={dt*1e3:.3f} ms  scan={st['scan_ms']/a.steps:.3f} ms  "
          f"scan GB/s={(st['scan_bytes']/1e9/(st['scan_ms']/1e3)) if st['scan_ms'] > 0 else 0:.0f}  qps={a.batch/dt:.0f}  cand/q={st['n_candidates']/a.steps/a.batch:.0f} "
          f"fallback={st['n_fallback']}")
    # same answers as the synchronous path
    idx.search_device(Q[a.steps + 1].data_ptr(), a.batch, a.k, D.data_ptr(), I.data_ptr())
    torch.cuda.synchronize()
    s = (a.steps + 1) % NS
    assert torch.equal(I, Is[s]) and torch.equal(D, Ds[s]), "async result differs from sync"
