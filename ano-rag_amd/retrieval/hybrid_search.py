"""MI355X drop-in for the reference's ``retrieval/hybrid_search.py``.

Same class, constructor, attributes and ``fuse`` contract as the reference (hybrid_search.py:13-115): the
final similarity of every note is the fusion of the dense, bm25 and graph retriever scores (``linear``:
max-normalised weighted sum, ``rrf``: weighted reciprocal rank) plus an additive path score; the result is
the list of ``{note_id, scores, final_similarity, tags}`` dicts, best first, cut to ``candidate_pool``.

The arithmetic runs on the device (libanorag_hip.so, float64 in the reference's order of operations, so
``final_similarity`` is bit-identical); this module only maps note ids to integers and back.  Short inputs (at most
4096 list entries per query) take ``anr_fuse_lists`` (everything LDS-resident); longer ones — e.g. the N-length
``bm25_scores`` vector zipped with the note ids — take ``anr_fuse_dense``: the long source becomes a device array
that is streamed once (fused elementwise + arg-k), so ``fuse`` accepts lists of any length, as the reference does.
``fuse_arrays`` is the same kernel for callers that already hold integer ids and device-resident score vectors
(``DeviceBM25.scores_device``): nothing but the pool-sized result leaves the device.
There is no CPU fallback: without the HIP library ``fuse`` raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Sequence, Tuple

import numpy as np

from anorag_hip import _lib
from anorag_hip.fusion import DeviceArray, fuse_dense

try:  # C shaping of the fused arrays into the reference's result dicts (csrc/pyshape.c)
    from anorag_hip import _pyshape
except ImportError:  # pragma: no cover - the extension is built by `make` next to libanorag_hip.so
    _pyshape = None

_SOURCES = ("dense", "bm25", "graph", "path")
_LIST_MAX = 4096    # entries per query of the LDS-resident list kernel (csrc/fusion_kernels.hpp kFuseMax)
_SHORT_MAX = 1024   # short-list entries per query beside the arrays of anr_fuse_dense


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
        big = [qi for qi, lists in enumerate(queries) if sum(len(lst or ()) for lst in lists) > _LIST_MAX]
        if big:  # long inputs: one anr_fuse_dense call each (every such query has its own id universe)
            results: List[Any] = [None] * nq
            for qi in big:
                results[qi] = self._fuse_long(queries[qi], pool)
            rest = [qi for qi in range(nq) if results[qi] is None]
            for qi, r in zip(rest, self.fuse_batch([queries[qi] for qi in rest]) if rest else []):
                results[qi] = r
            return results
        all_ids: List[int] = []
        all_sc: List[Any] = []
        offs = np.zeros((nq, 5), dtype=np.int64)
        id_maps: List[List[Any]] = []
        src_dicts: List[List[Dict[Any, float]]] = []
        for qi, lists in enumerate(queries):
            # sources as dicts: later duplicates overwrite, first position kept (hybrid_search.py:54-59)
            dicts = [dict(lst) if lst else {} for lst in lists]
            src_dicts.append(dicts)
            to_int: Dict[Any, int] = {}
            intern = to_int.setdefault  # id -> dense integer, in order of first appearance
            for si, d in enumerate(dicts):
                offs[qi, si] = len(all_ids)
                all_ids += [intern(nid, len(to_int)) for nid in d]
                all_sc += d.values()
            offs[qi, 4] = len(all_ids)
            id_maps.append(list(to_int))
        ids = np.fromiter(all_ids, dtype=np.int64, count=len(all_ids))
        sc = np.fromiter(all_sc, dtype=np.float64, count=len(all_sc))
        w = np.asarray([float(self.weights.get(k, 0.0)) for k in _SOURCES], dtype=np.float64)
        method = 1 if self.fusion_method == "rrf" else 0
        o_ids = np.empty((nq, pool), dtype=np.int64)
        o_fin = np.empty((nq, pool), dtype=np.float64)
        o_src = np.empty((nq, pool, 4), dtype=np.float64)
        o_cnt = np.empty((nq,), dtype=np.int32)
        lib = _lib.load()
        _lib.check(
            lib.anr_fuse_lists(self.device, method, nq, ids.ctypes.data, sc.ctypes.data,
                               offs.ctypes.data, w.ctypes.data, float(self.rrf_k), pool,
                               o_ids.ctypes.data, o_fin.ctypes.data,
                               o_src.ctypes.data, o_cnt.ctypes.data),
            "anr_fuse_lists",
        )
        results: List[List[Dict[str, Any]]] = []
        ids_l, fin_l, cnt_l = o_ids.tolist(), o_fin.tolist(), o_cnt.tolist()
        for qi in range(nq):
            names = id_maps[qi]
            d0, d1, d2, d3 = src_dicts[qi]
            g0, g1, g2, g3 = d0.get, d1.get, d2.get, d3.get
            fin_q = fin_l[qi]
            res = []
            for j, k in enumerate(ids_l[qi][:cnt_l[qi]]):
                nid = names[k]
                res.append({
                    "note_id": nid,
                    # the caller's own score objects (ints stay ints), None when absent (hybrid_search.py:74)
                    "scores": {"dense": g0(nid), "bm25": g1(nid), "graph": g2(nid), "path": g3(nid)},
                    "final_similarity": fin_q[j],
                    "tags": {"source": "graph" if nid in d2 else "semantic", "is_bridge": nid in d3},
                })
            results.append(res)
        return results


    # -- long inputs -------------------------------------------------------------------------------------------
    def _fuse_long(self, lists, pool: int) -> List[Dict[str, Any]]:
        """one query whose lists exceed the LDS kernel: the long sources become device arrays over this query's id
        universe (NaN = the id is absent from that source), the rest stay short lists"""
        dicts = [{nid: s for nid, s in (lst or [])} for lst in lists]
        lens = [len(d) for d in dicts]
        rrf = self.fusion_method == "rrf"
        if pool > _SHORT_MAX:  # beyond the streaming kernel's result capacity (anr_fuse_dense: pool <= 1024)
            return self._fuse_rrf_long(dicts, pool) if rrf else self._fuse_linear_rounds(dicts, pool)
        order = sorted(range(3 if rrf else 4), key=lambda s: -lens[s])
        arrays = [order[0]]

        def fits(arr):  # the limits anr_fuse_dense checks (csrc/fusion_dense.hip): short-list entries beside the arrays,
            m = sum(lens[t] for t in range(4) if t not in arr)  # and the entries of the composed lists of its last kernel
            return m <= _SHORT_MAX and len(arr) * (pool + 2 * m) + m <= _LIST_MAX

        if rrf:
            # (a negative weight on the streamed source turns its ranking round: the general path ranks by sorting)
            if not fits(arrays) or not float(self.weights.get(_SOURCES[arrays[0]], 0.0)) >= 0.0:
                return self._fuse_rrf_long(dicts, pool)  # two or three long lists: every one ranked by a device sort
        else:
            for s in order[1:]:  # linear takes any source as an array: promote the next longest until the rest fits
                if fits(arrays) or not lens[s]:
                    break
                arrays.append(s)
        # id universe: the first array's ids in ITS list order (rrf ranks ties in list order: array order == list order)
        to_int: Dict[Any, int] = {}
        names: List[Any] = []
        for s in arrays + [t for t in range(4) if t not in arrays]:
            for nid in dicts[s]:
                if nid not in to_int:
                    to_int[nid] = len(names)
                    names.append(nid)
        sources: Dict[str, Any] = {}
        held = []
        for s in range(4):
            if not dicts[s]:
                continue
            if s in arrays:
                n = lens[s] if s == arrays[0] else len(names)
                a = np.full((1, n), np.nan, dtype=np.float64)
                a[0, [to_int[nid] for nid in dicts[s]]] = [float(v) for v in dicts[s].values()]
                arr = DeviceArray.from_numpy(a, self.device, with_max=self.fusion_method != "rrf")
                held.append(arr)
                sources[_SOURCES[s]] = arr
            else:
                sources[_SOURCES[s]] = [(np.fromiter((to_int[nid] for nid in dicts[s]), dtype=np.int64, count=lens[s]),
                                         np.fromiter((float(v) for v in dicts[s].values()), dtype=np.float64,
                                                     count=lens[s]))]
        try:
            o_ids, o_fin, _, o_cnt = fuse_dense(self.fusion_method, self.weights, float(self.rrf_k), pool, 1, sources,
                                                device=self.device)
        finally:
            for arr in held:
                arr.free()
        res = []
        for j in range(int(o_cnt[0])):
            nid = names[int(o_ids[0, j])]
            res.append({
                "note_id": nid,
                "scores": {k: dicts[si].get(nid) for si, k in enumerate(_SOURCES)},
                "final_similarity": float(o_fin[0, j]),
                "tags": {"source": "graph" if nid in dicts[2] else "semantic", "is_bridge": nid in dicts[3]},
            })
        return res

    def _fuse_linear_rounds(self, dicts, pool: int) -> List[Dict[str, Any]]:
        """linear fusion of long lists with ``candidate_pool`` beyond the streaming kernel's 1024 results: every source
        becomes an array over the query's id universe and the kernel runs in rounds of 1024 — the ids a round returned
        are marked absent (NaN) for the next one, while each source keeps its ORIGINAL maximum as the normaliser
        (``row_max``), so every final is the one a single call would give; the order (final, then lower id) is total,
        so the rounds concatenate to the reference's list.  A rare configuration: one upload per round."""
        to_int: Dict[Any, int] = {}
        names: List[Any] = []
        for d in dicts:
            for nid in d:
                if nid not in to_int:
                    to_int[nid] = len(names)
                    names.append(nid)
        if not names:
            return []
        host, maxes = {}, {}
        for s in range(4):
            if dicts[s]:
                a = np.full((1, len(names)), np.nan, dtype=np.float64)
                a[0, [to_int[nid] for nid in dicts[s]]] = [float(v) for v in dicts[s].values()]
                host[s] = a
                maxes[s] = DeviceArray.from_numpy(np.fmax.reduce(a, axis=1).reshape(1, 1), self.device)
        res: List[Dict[str, Any]] = []
        try:
            while len(res) < pool:
                held = []
                try:
                    sources = {}
                    for s, a in host.items():
                        arr = DeviceArray.from_numpy(a, self.device)
                        held.append(arr)
                        arr.row_max = maxes[s]
                        sources[_SOURCES[s]] = arr
                    want = min(_SHORT_MAX, pool - len(res))
                    o_ids, o_fin, _, o_cnt = fuse_dense("linear", self.weights, float(self.rrf_k), want, 1, sources,
                                                        device=self.device)
                finally:
                    for arr in held:
                        arr.row_max = None  # shared: freed once, below
                        arr.free()
                n = int(o_cnt[0])
                for j in range(n):
                    i = int(o_ids[0, j])
                    nid = names[i]
                    res.append({
                        "note_id": nid,
                        "scores": {k: dicts[si].get(nid) for si, k in enumerate(_SOURCES)},
                        "final_similarity": float(o_fin[0, j]),
                        "tags": {"source": "graph" if nid in dicts[2] else "semantic", "is_bridge": nid in dicts[3]},
                    })
                    for a in host.values():
                        a[0, i] = np.nan
                if n < want:
                    break
        finally:
            for m in maxes.values():
                m.free()
        return res

    def _fuse_rrf_long(self, dicts, pool: int) -> List[Dict[str, Any]]:
        """rrf over lists of any length (``anr_fuse_rrf_long``): each ranked list is sorted on the device in its own
        order — the general form; one long list takes the streaming path above, short lists the LDS kernel"""
        to_int: Dict[Any, int] = {}
        intern = to_int.setdefault
        all_ids: List[int] = []
        all_sc: List[Any] = []
        offs = np.zeros((1, 5), dtype=np.int64)
        for si, d in enumerate(dicts):
            offs[0, si] = len(all_ids)
            all_ids += [intern(nid, len(to_int)) for nid in d]
            all_sc += d.values()
        offs[0, 4] = len(all_ids)
        names = list(to_int)
        if not names:
            return []
        ids = np.fromiter(all_ids, dtype=np.int64, count=len(all_ids))
        sc = np.fromiter(all_sc, dtype=np.float64, count=len(all_sc))
        w = np.asarray([float(self.weights.get(k, 0.0)) for k in _SOURCES], dtype=np.float64)
        o_ids = np.empty((1, pool), dtype=np.int64)
        o_fin = np.empty((1, pool), dtype=np.float64)
        o_src = np.empty((1, pool, 4), dtype=np.float64)
        o_cnt = np.empty((1,), dtype=np.int32)
        _lib.check(_lib.load().anr_fuse_rrf_long(self.device, 1, ids.ctypes.data, sc.ctypes.data,
                                                 offs.ctypes.data, len(names), w.ctypes.data,
                                                 float(self.rrf_k), pool, o_ids.ctypes.data,
                                                 o_fin.ctypes.data, o_src.ctypes.data,
                                                 o_cnt.ctypes.data), "anr_fuse_rrf_long")
        d0, d1, d2, d3 = dicts
        res = []
        for j in range(int(o_cnt[0])):
            nid = names[int(o_ids[0, j])]
            res.append({"note_id": nid,
                        "scores": {"dense": d0.get(nid), "bm25": d1.get(nid), "graph": d2.get(nid), "path": d3.get(nid)},
                        "final_similarity": float(o_fin[0, j]),
                        "tags": {"source": "graph" if nid in d2 else "semantic", "is_bridge": nid in d3}})
        return res

    def fuse_arrays(self, nq: int, dense=None, bm25=None, graph=None, path=None, note_ids: Sequence[Any] | None = None,
                    want_stats: bool = False):
        """EXTENSION (same arithmetic, integer ids): batch fusion where a source is either a ``DeviceArray`` [nq, N]
        (the score of EVERY note 0..N-1, e.g. ``DeviceBM25.scores_device``), the same rows by their non-zero entries
        (``SparseRows``, e.g. ``DeviceBM25.scores_sparse_device``) or, per query, a short
        ``(ids, scores)`` pair (e.g. the dense top-k of ``VectorIndex``).  Returns the reference's list of result
        dicts per query (``note_id`` = ``note_ids[i]`` when given, else the integer id)."""
        if not self.enabled:
            return [[] for _ in range(nq)]
        pool = int(self.candidate_pool)
        if pool <= 0 or nq == 0:
            return [[] for _ in range(nq)]
        out = fuse_dense(self.fusion_method, self.weights, float(self.rrf_k), pool, nq,
                         {"dense": dense, "bm25": bm25, "graph": graph, "path": path}, device=self.device,
                         want_stats=want_stats)
        o_ids, o_fin, o_src, o_cnt = out[:4]
        results = _shape_fused(o_ids, o_fin, o_src, o_cnt, note_ids)
        return (results, out[4]) if want_stats else results


    def fuse_bm25(self, corpus, queries: Sequence[Sequence[str]], dense=None, graph=None, path=None,
                  note_ids: Sequence[Any] | None = None):
        """EXTENSION: BM25 scoring and fusion back to back on the device — ``bm25_scores`` (utils/bm25_search.py:286-340)
        feeding ``fuse`` (hybrid_search.py:34-103) for a batch of tokenised queries, same results as
        ``fuse_arrays(bm25=corpus.scores_device(queries))``.  A query whose postings touch at most
        ``corpus.SPARSE_CAP_MAX`` (65 536) documents is handed over in sparse form (its N-vector is never formed); the
        others take the N-vector path.  ``corpus``: a ``DeviceBM25``; dense / graph / path: per query ``(ids, scores)`` or None."""
        nq = len(queries)
        if not self.enabled or int(self.candidate_pool) <= 0 or nq == 0:
            return [[] for _ in range(nq)]
        rows = corpus.scores_sparse_device(queries, normalize=True, allow_overflow=True)
        try:
            heavy = np.flatnonzero(rows.counts < 0).tolist()
            if 2 * len(heavy) > nq:  # mostly frequent-word queries: one pass down the N-vector path for all of them
                heavy, results = list(range(nq)), [None] * nq
            else:
                if heavy:
                    # the overflow-marked rows are answered below from their N-vectors; here they are fused as EMPTY rows
                    # (the fusion refuses a count of -1 — it would otherwise read "every BM25 score is 0.0" silently)
                    rows.counts = np.maximum(rows.counts, 0).astype(np.int32)
                    _lib.check(_lib.load().anr_device_copy(rows.device, C.c_void_p(rows.count_ptr),
                                                           rows.counts.ctypes.data, rows.counts.nbytes, 0),
                               "anr_device_copy")
                results = self.fuse_arrays(nq, dense=dense, bm25=rows, graph=graph, path=path, note_ids=note_ids)
        finally:
            rows.free()
        if heavy:
            sub = lambda lists: None if lists is None else [lists[i] for i in heavy]
            vec = corpus.scores_device([queries[i] for i in heavy], normalize=True)
            try:
                again = self.fuse_arrays(len(heavy), dense=sub(dense), bm25=vec, graph=sub(graph), path=sub(path),
                                         note_ids=note_ids)
            finally:
                vec.free()
            for i, r in zip(heavy, again):
                results[i] = r
        return results


def _shape_fused_py(o_ids, o_fin, o_src, o_cnt, note_ids=None):
    """the result dicts of the reference's fuse (hybrid_search.py:85-103) from the anr_fuse_dense output arrays"""
    results = []
    ids_l, fin_l = o_ids.tolist(), o_fin.tolist()
    src_l = np.where(np.isnan(o_src), None, o_src.astype(object)).tolist()
    for qi in range(o_ids.shape[0]):
        res = []
        for j in range(int(o_cnt[qi])):
            i = ids_l[qi][j]
            sc = src_l[qi][j]
            res.append({"note_id": note_ids[i] if note_ids is not None else i,
                        "scores": dict(zip(_SOURCES, sc)),
                        "final_similarity": fin_l[qi][j],
                        "tags": {"source": "graph" if sc[2] is not None else "semantic",
                                 "is_bridge": sc[3] is not None}})
        results.append(res)
    return results


def _shape_fused(o_ids, o_fin, o_src, o_cnt, note_ids=None):
    if _pyshape is not None:  # the same dicts, built in C: 16 000 results took 9 ms in the Python loop
        nq, pool = o_ids.shape
        return _pyshape.shape_fused(np.ascontiguousarray(o_ids), np.ascontiguousarray(o_fin), np.ascontiguousarray(o_src),
                                    np.ascontiguousarray(o_cnt, dtype=np.int32), nq, pool, note_ids)
    return _shape_fused_py(o_ids, o_fin, o_src, o_cnt, note_ids)


def create_hybrid_searcher(config: Dict[str, Any]) -> HybridSearcher:
    """Create a HybridSearcher instance with the given configuration (reference hybrid_search.py:106-115)."""
    return HybridSearcher(config)
