#!/usr/bin/env python3
"""Developer tool: soak test — thousands of asynchronous and synchronous searches interleaved with adds, resets,
option changes and host-buffer calls, every result compared with an exact torch reference on the device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd"))
import numpy as np
import torch
from anorag_hip import FlatIndex, METRIC_IP
from anorag_hip._lib import OPT_STREAMS
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
d, k = 128, 20
idx = FlatIndex(d, METRIC_IP, normalize=True)
X = torch.empty((0, d), device=dev)
S = [torch.cuda.Stream() for _ in range(3)]
bad = 0; n_search = 0
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 60
rnd = np.random.default_rng(3)
def check(Q, D, I):
    global bad
    Xn = torch.nn.functional.normalize(X.double(), dim=1); Qn = torch.nn.functional.normalize(Q.double(), dim=1)
    s = Qn @ Xn.T
    kth = torch.topk(s, min(k, X.shape[0]), dim=1).values[:, -1:]
    got = torch.gather(s, 1, I[:, : min(k, X.shape[0])].clamp(min=0))
    if not bool(torch.all(got >= kth - 1e-6)) or float((D[:, : min(k, X.shape[0])].double() - got).abs().max()) > 1e-4:
        bad += 1
it = 0
while time.time() < t_end:
    it += 1
    op = rnd.integers(0, 10)
    if op == 0 and X.shape[0] < 600_000:
        m = int(rnd.integers(1, 90_000))
        x = torch.randn((m, d), generator=g, device=dev); torch.cuda.synchronize()
        idx.add_device(x.data_ptr(), m); X = torch.cat([X, x])
    elif op == 1 and rnd.integers(0, 8) == 0:
        idx.reset(); X = torch.empty((0, d), device=dev)
    elif op == 2:
        idx.set_option(OPT_STREAMS, int(rnd.integers(1, 4)))
    elif X.shape[0] == 0:
        continue
    elif op in (3, 4):                                  # host buffers, odd batch sizes
        nq = int(rnd.integers(1, 150))
        Q = torch.randn((nq, d), generator=g, device=dev)
        D, I = idx.search(Q.cpu().numpy(), k)
        check(Q, torch.from_numpy(D).to(dev), torch.from_numpy(I).to(dev)); n_search += nq
    else:                                               # a burst of asynchronous batches
        nb = int(rnd.integers(1, 9))
        Qs = torch.randn((nb, 64, d), generator=g, device=dev); torch.cuda.synchronize()
        outs = []
        for b in range(nb):
            D = torch.empty((64, k), device=dev); I = torch.empty((64, k), device=dev, dtype=torch.int64)
            idx.search_device_async(Qs[b].data_ptr(), 64, k, D.data_ptr(), I.data_ptr(), S[b % 3].cuda_stream)
            outs.append((D, I))
        idx.sync(); torch.cuda.synchronize()
        for b, (D, I) in enumerate(outs):
            check(Qs[b], D, I); n_search += 64
    if it % 200 == 0:
        print(f"iter {it}: rows={X.shape[0]} searches={n_search} mismatches={bad}", flush=True)
print(f"done: iterations={it} searches={n_search} mismatches={bad}")
sys.exit(1 if bad else 0)
