"""The graph layer's two dense scans on the device.

* ``semantic_similarity_relations`` replaces the body of ``RelationExtractor.extract_semantic_similarity_relations``
  (graph/relation_extractor.py:591-629): the reference forms the full N x N cosine matrix (:769-782, N^2 floats —
  40 GB at 100 k notes) and walks its upper triangle in Python; here the stored rows are multiplied block
  against block on the MFMA units (``anr_index_self_join``) and only the pairs at or above the threshold ever
  leave the device.  Output: the same list of relation dicts, in the same (i, j) order.
* ``find_embedding_candidates`` replaces the arithmetic of ``GraphRetriever._find_embedding_candidates``
  (graph/graph_retriever.py:153-170): un-normalised inner product of every stored embedding with the query and
  the top_k indices, best first — one search on an inner-product index.

Both fail loudly when the HIP library is missing; there is no CPU path here (the oracle lives in oracle/).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .flat_index import METRIC_IP, FlatIndex


def similarity_pairs(embeddings: np.ndarray, threshold: float, device: int = 0, index: Optional[FlatIndex] = None):
    """(i, j, cosine) of every pair i < j with cosine >= threshold, sorted by (i, j); plus each pair's rank in
    row i of the (never formed) similarity matrix: 1 + the number of entries of that row that are larger."""
    own = index is None
    if own:
        emb = np.ascontiguousarray(embeddings, dtype=np.float32)
        index = FlatIndex(emb.shape[1], METRIC_IP, normalize=True, device=device)
        index.add(emb)
    try:
        I, J, S = index.self_join(float(threshold))
    finally:
        if own:
            index.close()
    # rank of j in row i: every entry of row i above S[i, j] >= threshold is itself a reported pair (with i on
    # either side), so the symmetric pair list is enough
    src = np.concatenate([I, J])
    dst = np.concatenate([J, I])
    val = np.concatenate([S, S])
    order = np.lexsort((dst, -val.astype(np.float64), src))
    src_o, val_o = src[order], val[order]
    start = np.r_[0, np.flatnonzero(np.diff(src_o)) + 1] if len(src_o) else np.zeros(0, dtype=np.int64)
    group_start = np.repeat(start, np.diff(np.r_[start, len(src_o)])) if len(src_o) else start
    pos = np.arange(len(src_o)) - group_start
    # strictly-greater count: equal values share the rank of the first of them
    idx = np.arange(len(src_o))
    new_run = np.r_[True, (src_o[1:] != src_o[:-1]) | (val_o[1:] != val_o[:-1])] if len(src_o) else np.zeros(0, dtype=bool)
    run_first = np.maximum.accumulate(np.where(new_run, idx, 0)) if len(src_o) else idx
    first_equal = pos[run_first] if len(src_o) else pos
    rank_sorted = first_equal + 1
    rank_all = np.empty(len(src_o), dtype=np.int64)
    rank_all[order] = rank_sorted
    return I, J, S, rank_all[: len(I)]


def semantic_similarity_relations(atomic_notes: Sequence[Dict[str, Any]], embeddings: np.ndarray, threshold: float = 0.7,
                                  weight: float = 0.5, device: int = 0) -> List[Dict[str, Any]]:
    """Same list as graph/relation_extractor.py:591-629 returns (threshold = graph.similarity_threshold,
    weight = graph.weights.semantic_similarity)."""
    relations: List[Dict[str, Any]] = []
    if embeddings.shape[0] != len(atomic_notes):
        return relations  # the reference logs a warning and returns nothing (:596-598)
    if len(atomic_notes) < 2:
        return relations
    I, J, S, R = similarity_pairs(embeddings, threshold, device)
    for i, j, s, r in zip(I.tolist(), J.tolist(), S, R.tolist()):
        relations.append({
            "source_id": atomic_notes[i].get("note_id"),
            "target_id": atomic_notes[j].get("note_id"),
            "relation_type": "semantic_similarity",
            "weight": weight * s,
            "metadata": {"cosine_similarity": float(s), "similarity_rank": int(r)},
        })
    return relations


def find_embedding_candidates(embeddings, query_embedding: np.ndarray, top_k: int = 15, device: int = 0) -> np.ndarray:
    """indices of the top_k stored embeddings by un-normalised inner product with the query, best first
    (graph_retriever.py:158-162).  ``embeddings`` is an [N, d] array or an inner-product FlatIndex built from it."""
    own = not isinstance(embeddings, FlatIndex)
    if own:
        emb = np.ascontiguousarray(embeddings, dtype=np.float32)
        index = FlatIndex(emb.shape[1], METRIC_IP, normalize=False, device=device)
        index.add(emb)
    else:
        index = embeddings
    try:
        k = min(int(top_k), index.ntotal)
        if k <= 0:
            return np.zeros((0,), dtype=np.int64)
        _, I = index.search(np.asarray(query_embedding, dtype=np.float32).reshape(1, -1), k)
    finally:
        if own:
            index.close()
    return I[0]
