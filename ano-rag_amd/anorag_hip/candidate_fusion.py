"""Candidate-level score fusion of the reference's ``QueryProcessor`` on the device.

``QueryProcessor._hybrid_search`` (query/query_processor.py:3680-3768) and ``_enhanced_hybrid_search_v2``
(:1088-1143) fuse, per candidate, a vector similarity with a BM25 score — linear or RRF with 0-based ranks — and
apply guardrail multipliers (must-have terms, boost entities / predicates, section / lexical penalties, a noise
floor), then sort the candidates by the fused score.  The text matching that decides the multipliers is host work
(Python string search, as there); the arithmetic and the ordering run in ``anr_fuse_candidates`` for a whole batch
of queries, float64 in the reference's order of operations (bit-identical scores, same stable order).

OPT-IN: nothing in the reference's call path is replaced by importing this module — wiring it in is the
maintainer's choice, because the reference obtains the vector scores by re-encoding every
candidate, which ``VectorRetriever.score_candidates`` replaces with stored embeddings.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

MODES = {"linear": 0, "rrf": 1, "v2": 2}


def _flat(seqs, dtype, total):
    if seqs is None:
        return None
    out = np.empty(total, dtype=dtype)
    pos = 0
    for s in seqs:
        out[pos:pos + len(s)] = s
        pos += len(s)
    return out


def fuse_candidates(mode: str, vector_scores: Sequence[Sequence[float]], bm25_scores: Sequence[Sequence[float]], *,
                    vector_weight: float = 1.0, bm25_weight: float = 1.0, rrf_k: float = 60.0,
                    missing_terms: Optional[Sequence[Sequence[bool]]] = None,
                    n_entities: Optional[Sequence[Sequence[int]]] = None,
                    n_predicates: Optional[Sequence[Sequence[int]]] = None,
                    multipliers: Optional[Sequence[np.ndarray]] = None, noise_threshold: float = 0.0,
                    device: int = 0) -> List[Tuple[np.ndarray, np.ndarray]]:
    """One entry per query: (scores float64 [n], order int32 [n] — candidate indices best first, ties in candidate
    order).  ``mode``: "linear" / "rrf" (``_hybrid_search``) or "v2" (``_enhanced_hybrid_search_v2``; candidates
    whose score is 0 are still listed — the reference drops them after the loop, :1144).
    ``missing_terms[q][i]``: the candidate does not contain any must-have term; ``n_entities`` / ``n_predicates``:
    how many boost entities / predicates it contains (linear); ``multipliers[q]``: [n, 4] section, lexical, entity,
    predicate factors (v2; 1.0 where the reference does not apply one)."""
    nq = len(vector_scores)
    if nq == 0:
        return []
    lens = [len(v) for v in vector_scores]
    offs = np.zeros(nq + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    total = int(offs[-1])
    a = _flat(vector_scores, np.float64, total)
    b = _flat([list(s)[:n] + [0.0] * (n - len(s)) for s, n in zip(bm25_scores, lens)], np.float64, total)
    fl = _flat(missing_terms, np.int32, total)
    ne = _flat(n_entities, np.int32, total)
    npd = _flat(n_predicates, np.int32, total)
    mult = None
    if mode == "v2":
        mult = (np.concatenate([np.asarray(m, dtype=np.float64).reshape(-1, 4) for m in multipliers])
                if multipliers is not None else np.ones((total, 4)))
        mult = np.ascontiguousarray(mult, dtype=np.float64)
    score = np.empty(total, dtype=np.float64)
    order = np.empty(total, dtype=np.int32)

    def ptr(x):
        return x.ctypes.data if x is not None else None
    _lib.check(_lib.load().anr_fuse_candidates(int(device), MODES[mode], nq, ptr(offs), ptr(a), ptr(b), ptr(fl), ptr(ne),
                                               ptr(npd), ptr(mult), float(vector_weight), float(bm25_weight),
                                               float(rrf_k), float(noise_threshold), ptr(score), ptr(order)),
               "anr_fuse_candidates")
    return [(score[offs[q]:offs[q + 1]], order[offs[q]:offs[q + 1]]) for q in range(nq)]
