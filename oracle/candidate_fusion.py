"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of the candidate-level fusion loops of the reference's QueryProcessor:
  * ``hybrid_scores``   follows ``_hybrid_search`` (query/query_processor.py:3703-3762): linear
    (``vector_weight * v + bm25_weight * b`` after the guardrail multipliers 0.1 / 1.2 / 1.3) or rrf with 0-based
    ranks from stable descending sorts, then the stable descending sort by ``hybrid_score``;
  * ``enhanced_v2_scores`` follows ``_enhanced_hybrid_search_v2`` (:1104-1146): ``1.0 * dense + 0.6 * sparse``, section /
    lexical penalties, noise floor, entity / predicate boosts, drop the zeros, sort.
PINNED by tests/golden/candidate_fusion_cases.json, which tests/golden/make_golden.py produced by compiling those two
method bodies from the reference's source as they stand and running them.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence


def guardrail_inputs(candidates: Sequence[Dict[str, Any]], must_have_terms, boost_entities, boost_predicates):
    """the text matching of :3714-3731 as per-candidate (missing, n_entities, n_predicates)"""
    miss, ne, npd = [], [], []
    for c in candidates:
        content = c.get("content", "").lower()
        miss.append(bool(must_have_terms) and not any(t.lower() in content for t in must_have_terms))
        ne.append(sum(1 for e in (boost_entities or []) if e.lower() in content))
        npd.append(sum(1 for p in (boost_predicates or []) if p.lower() in content))
    return miss, ne, npd


def hybrid_scores(method: str, vs: Sequence[float], bs: Sequence[float], miss, ne, npd, vector_weight: float,
                  bm25_weight: float, rrf_k: float):
    n = len(vs)
    out = []
    if method == "linear":
        for i in range(n):
            v, b = vs[i], bs[i]
            if miss[i]:
                b *= 0.1
            for _ in range(ne[i]):
                v *= 1.2
            for _ in range(npd[i]):
                b *= 1.3
            out.append(vector_weight * v + bm25_weight * b)
    else:
        vr = {i: r for r, i in enumerate(sorted(range(n), key=lambda x: vs[x], reverse=True))}
        br = {i: r for r, i in enumerate(sorted(range(n), key=lambda x: bs[x], reverse=True))}
        for i in range(n):
            s = vector_weight / (rrf_k + vr[i]) + bm25_weight / (rrf_k + br[i])
            if miss[i]:
                s *= 0.1
            out.append(s)
    order = sorted(range(n), key=lambda i: out[i], reverse=True)
    return out, order


def enhanced_v2_scores(dense, sparse, mult, not_ok, noise_threshold: float):
    """mult[i] = (section, lexical, entity, predicate) factors, 1.0 where the reference applies none"""
    out = []
    for i in range(len(dense)):
        f = 1.0 * dense[i] + 0.6 * sparse[i]
        f *= mult[i][0]
        f *= mult[i][1]
        if f < noise_threshold and not_ok[i]:
            f = 0.0
        f *= mult[i][2]
        f *= mult[i][3]
        out.append(f)
    order = [i for i in sorted(range(len(out)), key=lambda i: out[i], reverse=True) if out[i] > 0]
    return out, order
