#!/usr/bin/env python3
"""Developer tool: A/B the scan kernel under environment switches, interleaved in one process so that
box-to-box and thermal drift cancel.  usage: scan_ab.py --rows N --variants "A=1,B=2;A=0" (';' separates variants,
',' separates NAME=VALUE pairs; an empty variant is the default build)."""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd"))
import torch
from anorag_hip import FlatIndex, METRIC_IP
from anorag_hip._lib import OPT_TIMING

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_250_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--variants", default="")
ap.add_argument("--sample", type=int, default=0)
a = ap.parse_args()
variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in a.variants.split(";")]
names = sorted({k for v in variants for k in v})
dev = torch.device("cuda", 0)
idx = FlatIndex(a.dim, METRIC_IP, normalize=True)
idx.reserve(a.rows)
g = torch.Generator(device=dev); g.manual_seed(1)
done = 0
while done < a.rows:
    m = min(262144, a.rows - done)
    x = torch.randn((m, a.dim), generator=g, device=dev)
    torch.cuda.synchronize()
    idx.add_device(x.data_ptr(), m)
    done += m
idx.set_option(OPT_TIMING, 1)
if a.sample:
    from anorag_hip._lib import OPT_SAMPLE_ROWS
    idx.set_option(OPT_SAMPLE_ROWS, a.sample)
Q = torch.randn((a.steps, 64, a.dim), generator=g, device=dev)
D = torch.empty((64, 100), device=dev); I = torch.empty((64, 100), device=dev, dtype=torch.int64)
res = [[] for _ in variants]
tot = [[] for _ in variants]
for r in range(a.rounds + 1):
    for vi, v in enumerate(variants):
        for n in names:
            os.environ.pop(n, None)
        os.environ.update(v)
        s = t = 0.0
        for i in range(a.steps):
            idx.search_device(Q[i].data_ptr(), 64, 100, D.data_ptr(), I.data_ptr())
            st = idx.last_stats()
            s += st["scan_ms"]; t += st["total_ms"]
        if r:  # round 0 warms up
            res[vi].append(s / a.steps); tot[vi].append(t / a.steps)
for vi, v in enumerate(variants):
    print(f"rows={a.rows} {str(v):40s} scan median={statistics.median(res[vi]):.4f} min={min(res[vi]):.4f} max={max(res[vi]):.4f} ms"
          f"  total median={statistics.median(tot[vi]):.4f}  cand/q={st['n_candidates']/64:.0f}", flush=True)
