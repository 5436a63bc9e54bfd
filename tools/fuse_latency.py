#!/usr/bin/env python3
"""Developer tool: latency of ONE HybridSearcher.fuse call the way the reference makes it — string note ids, a dense
top-100 list, 1000 BM25 hits, 15 graph hits, pool 80 — beside the reference algorithm in Python (oracle restatement)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd")); sys.path.insert(0, ROOT)
import numpy as np
from retrieval.hybrid_search import HybridSearcher
from oracle import fusion as ofu
rng = np.random.default_rng(0)
W = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
for method in ("linear", "rrf"):
    hs = HybridSearcher({"retrieval": {"candidate_pool": 80, "hybrid": {"fusion_method": method, "rrf_k": 60, "weights": W}}})
    ids = [f"note_{i:06d}" for i in range(20000)]
    dense = [(ids[i], float(s)) for i, s in zip(rng.choice(20000, 100, replace=False), np.sort(rng.random(100))[::-1])]
    bm25 = [(ids[i], float(s)) for i, s in zip(rng.choice(20000, 1000, replace=False), rng.random(1000))]
    graph = [(ids[i], float(s)) for i, s in zip(rng.choice(20000, 15, replace=False), rng.random(15))]
    for _ in range(600): hs.fuse(dense, bm25, graph, None)
    t0 = time.perf_counter()
    for _ in range(300): r = hs.fuse(dense, bm25, graph, None)
    t1 = time.perf_counter()
    for _ in range(20): e = ofu.fuse(dense, bm25, graph, None, candidate_pool=80, fusion_method=method, weights=W, rrf_k=60)
    t2 = time.perf_counter()
    by_id = {x["note_id"]: x for x in e}   # (equal finals may come in another order: the reference iterates a set)
    same = [x["final_similarity"] for x in r] == [x["final_similarity"] for x in e] and \
        all(x["scores"] == by_id[x["note_id"]]["scores"] for x in r if x["note_id"] in by_id)
    print(f"{method}: HybridSearcher.fuse {1e6*(t1-t0)/300:.0f} us per query; the reference algorithm in Python {1e6*(t2-t1)/20:.0f} us; same results {same}")
