"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of the reference's score fusion, retrieval/hybrid_search.py:
  * ``normalize``      follows ``HybridSearcher._normalize`` (:26-32): divide by the MAX (not min-max);
                       max == 0 -> all 0.0; a negative max flips signs (kept: it is the reference's behaviour);
  * ``fuse``           follows ``HybridSearcher.fuse`` (:34-103) for both methods:
      - sources are dicts built from the (id, score) lists, later duplicates overwrite (:54-59);
      - ``rrf`` (:64-82): per source dense/bm25/graph, stable sort by score descending (ties keep dict
        insertion order), rank from 1, ``score[id] += w_src / (rrf_k + rank)``; then
        ``final = score + w_path * path.get(id, 0)``; ids that occur only in ``path`` are dropped;
      - ``linear`` (:83-100): ``final = sum_src w_src * (s / max_src)`` over dense/bm25/graph plus
        ``w_path * path_raw``, over the union of all ids (a Python ``set``: the order among exactly equal
        finals is hash-seed dependent in the reference);
      - stable sort by ``final_similarity`` descending, cut to ``candidate_pool`` (:102-103).
Pinned by tests/golden/fusion_cases.json, produced by running the reference file itself
(tests/golden/make_golden.py).  ``fuse_arrays`` is the same arithmetic on integer ids / numpy arrays, the form
the device kernel consumes; it is checked against ``fuse`` in the tests.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

SOURCES = ("dense", "bm25", "graph")


def normalize(scores: Dict[Any, float]) -> Dict[Any, float]:
    if not scores:
        return {}
    m = max(scores.values())
    if m == 0:
        return {k: 0.0 for k in scores}
    return {k: v / m for k, v in scores.items()}


def fuse(dense=None, bm25=None, graph=None, path=None, *, candidate_pool=50, enabled=True, fusion_method="linear",
         weights: Optional[Dict[str, float]] = None, rrf_k=60) -> List[Dict[str, Any]]:
    if not enabled:
        return []
    weights = weights or {}
    src = {
        "dense": {n: s for n, s in (dense or [])},
        "bm25": {n: s for n, s in (bm25 or [])},
        "graph": {n: s for n, s in (graph or [])},
        "path": {n: s for n, s in (path or [])},
    }
    results = []

    def entry(nid, final):
        return {
            "note_id": nid,
            "scores": {k: src[k].get(nid) for k in ("dense", "bm25", "graph", "path")},
            "final_similarity": final,
            "tags": {"source": "graph" if nid in src["graph"] else "semantic", "is_bridge": nid in src["path"]},
        }

    if fusion_method == "rrf":
        acc: Dict[Any, float] = {}
        for key in SOURCES:
            ordered = sorted(src[key].items(), key=lambda kv: kv[1], reverse=True)
            w = weights.get(key, 0.0)
            for rank, (nid, _) in enumerate(ordered, start=1):
                acc.setdefault(nid, 0.0)
                acc[nid] += w / (rrf_k + rank)
        for nid, s in acc.items():
            results.append(entry(nid, s + weights.get("path", 0.0) * src["path"].get(nid, 0.0)))
    else:
        normed = {k: (normalize(v) if k != "path" else v) for k, v in src.items()}
        ids = set().union(*[set(d) for d in src.values()])
        for nid in ids:
            final = (weights.get("dense", 0.0) * normed["dense"].get(nid, 0.0)
                     + weights.get("bm25", 0.0) * normed["bm25"].get(nid, 0.0)
                     + weights.get("graph", 0.0) * normed["graph"].get(nid, 0.0)
                     + weights.get("path", 0.0) * normed["path"].get(nid, 0.0))
            results.append(entry(nid, final))
    results.sort(key=lambda r: r["final_similarity"], reverse=True)
    return results[:candidate_pool]


def fuse_arrays(n: int, lists: Sequence[Optional[Tuple[np.ndarray, np.ndarray]]], weights: Sequence[float],
                method: str, rrf_k: float, pool: int):
    """Array form over integer ids in [0, n): lists = (dense, bm25, graph, path), each (ids int64, scores f64)
    with unique ids, in the caller's list order.  Returns (ids, finals) best-first.

    Order among exactly equal finals: ``rrf`` -> the reference's (deterministic) order, i.e. insertion order
    of its ``ranks`` dict = (first source holding the id, rank inside that source); ``linear`` -> ascending
    id (the reference iterates a set there, its order is hash-seed dependent)."""
    present = np.zeros(n, dtype=bool)
    final = np.zeros(n, dtype=np.float64)
    tie = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
    for si in range(3):
        if lists[si] is None or len(lists[si][0]) == 0:
            continue
        ids, sc = lists[si]
        sc = sc.astype(np.float64)
        if method == "rrf":
            order = np.argsort(-sc, kind="stable")  # ties keep list order
            rank = np.empty(len(ids), dtype=np.int64)
            rank[order] = np.arange(1, len(ids) + 1)
            final[ids] += weights[si] / (rrf_k + rank)
            tie[ids] = np.minimum(tie[ids], (np.int64(si) << 40) | rank)
        else:
            m = sc.max()
            final[ids] += weights[si] * (np.zeros_like(sc) if m == 0 else sc / m)
        present[ids] = True
    if lists[3] is not None and len(lists[3][0]):
        pid, ps = lists[3]
        if method == "rrf":
            keep = present[pid]
            final[pid[keep]] += weights[3] * ps[keep].astype(np.float64)
        else:
            final[pid] += weights[3] * ps.astype(np.float64)
            present[pid] = True
    cand = np.nonzero(present)[0]
    second = tie[cand] if method == "rrf" else cand
    order = np.lexsort((second, -final[cand]))[:pool]
    sel = cand[order]
    return sel, final[sel]
