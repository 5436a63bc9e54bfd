"""MI355X drop-in for the reference's ``retrieval/hybrid_search.py``.

Same class, constructor, attributes and ``fuse`` contract as the reference (hybrid_search.py:13-115): the
final similarity of every note is the fusion of the dense, bm25 and graph retriever scores (``linear``:
max-normalised weighted sum, ``rrf``: weighted reciprocal rank) plus an additive path score; the result is
the list of ``{note_id, scores, final_similarity, tags}`` dicts, best first, cut to ``candidate_pool``.

The arithmetic runs on the device (``anr_fuse_lists`` of libanorag_hip.so, float64 in the reference's order of
operations, so ``final_similarity`` is bit-identical); this module only maps note ids to integers and back.
There is no CPU fallback: without the HIP library ``fuse`` raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Sequence, Tuple

import numpy as np

from anorag_hip import _lib

_SOURCES = ("dense", "bm25", "graph", "path")


class HybridSearcher:
    """Fuse scores from multiple retrievers according to configuration (reference hybrid_search.py:13)."""

    def __init__(self, config: Dict[str, Any]):
        cfg = config if isinstance(config, dict) else config.load_config()
        r_cfg = cfg.get("retrieval", {})
        h_cfg = r_cfg.get("hybrid", {})
        self.candidate_pool = r_cfg.get("candidate_pool", 50)
        self.enabled = h_cfg.get("enabled", True)
        self.fusion_method = h_cfg.get("fusion_method", "linear")
        self.weights = h_cfg.get("weights", {})
        self.rrf_k = h_cfg.get("rrf_k", 60)
        self.device = int(cfg.get("anorag_hip", {}).get("device", 0)) if isinstance(cfg, dict) else 0

    # kept for API parity (reference hybrid_search.py:26-32); the device applies the same rule
    def _normalize(self, scores: Dict[str, float]) -> Dict[str, float]:
        if not scores:
            return {}
        max_score = max(scores.values())
        if max_score == 0:
            return {k: 0.0 for k in scores}
        return {k: v / max_score for k, v in scores.items()}

    def fuse(
        self,
        dense: List[Tuple[str, float]] | None = None,
        bm25: List[Tuple[str, float]] | None = None,
        graph: List[Tuple[str, float]] | None = None,
        path: List[Tuple[str, float]] | None = None,
    ) -> List[Dict[str, Any]]:
        """Fuse scores and produce final similarity (reference hybrid_search.py:34-103)."""
        if not self.enabled:
            return []
        out = self.fuse_batch([(dense, bm25, graph, path)])
        return out[0]

    def fuse_batch(self, queries: Sequence[Tuple[Any, Any, Any, Any]]) -> List[List[Dict[str, Any]]]:
        """Several independent ``fuse`` calls in one device launch (one workgroup per query)."""
        if not self.enabled:
            return [[] for _ in queries]
        nq = len(queries)
        if nq == 0:
            return []
        pool = int(self.candidate_pool)
        if pool <= 0:
            return [[] for _ in queries]
        all_ids: List[int] = []
        all_sc: List[float] = []
        offs = np.zeros((nq, 5), dtype=np.int64)
        id_maps: List[List[Any]] = []
        src_dicts: List[List[Dict[Any, float]]] = []
        for qi, lists in enumerate(queries):
            # sources as dicts: later duplicates overwrite, first position kept (hybrid_search.py:54-59)
            dicts = [{nid: s for nid, s in (lst or [])} for lst in lists]
            src_dicts.append(dicts)
            to_int: Dict[Any, int] = {}
            names: List[Any] = []
            for si, d in enumerate(dicts):
                offs[qi, si] = len(all_ids)
                for nid, s in d.items():
                    k = to_int.get(nid)
                    if k is None:
                        k = len(names)
                        to_int[nid] = k
                        names.append(nid)
                    all_ids.append(k)
                    all_sc.append(float(s))
            offs[qi, 4] = len(all_ids)
            id_maps.append(names)
        ids = np.asarray(all_ids, dtype=np.int64)
        sc = np.asarray(all_sc, dtype=np.float64)
        w = np.asarray([float(self.weights.get(k, 0.0)) for k in _SOURCES], dtype=np.float64)
        method = 1 if self.fusion_method == "rrf" else 0
        o_ids = np.empty((nq, pool), dtype=np.int64)
        o_fin = np.empty((nq, pool), dtype=np.float64)
        o_src = np.empty((nq, pool, 4), dtype=np.float64)
        o_cnt = np.empty((nq,), dtype=np.int32)
        lib = _lib.load()
        _lib.check(
            lib.anr_fuse_lists(self.device, method, nq, ids.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
                               offs.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), float(self.rrf_k), pool,
                               o_ids.ctypes.data_as(C.c_void_p), o_fin.ctypes.data_as(C.c_void_p),
                               o_src.ctypes.data_as(C.c_void_p), o_cnt.ctypes.data_as(C.c_void_p)),
            "anr_fuse_lists",
        )
        results: List[List[Dict[str, Any]]] = []
        for qi in range(nq):
            names = id_maps[qi]
            dicts = src_dicts[qi]
            res = []
            for j in range(int(o_cnt[qi])):
                nid = names[int(o_ids[qi, j])]
                res.append({
                    "note_id": nid,
                    # the caller's own score objects (ints stay ints), None when absent (hybrid_search.py:74)
                    "scores": {k: dicts[si].get(nid) for si, k in enumerate(_SOURCES)},
                    "final_similarity": float(o_fin[qi, j]),
                    "tags": {
                        "source": "graph" if nid in dicts[2] else "semantic",
                        "is_bridge": nid in dicts[3],
                    },
                })
            results.append(res)
        return results


def create_hybrid_searcher(config: Dict[str, Any]) -> HybridSearcher:
    """Create a HybridSearcher instance with the given configuration (reference hybrid_search.py:106-115)."""
    return HybridSearcher(config)
