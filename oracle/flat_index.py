"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of the reference's exact dense search, vector_store/vector_index.py:
  * ``preprocess_vectors``  follows ``VectorIndex._preprocess_vectors`` (:265-282): float32 cast,
    C-contiguous, and for cosine ``v / where(norm == 0, 1, norm)`` with numpy's float32 norm;
  * ``flat_search``         restates what ``self.index.search(q, top_k)`` (:223) returns for
    ``index_type == 'Flat'`` (:77-80): faiss ``IndexFlatIP`` / ``IndexFlatL2`` — a third-party library
    (``faiss-cpu>=1.7.4`` / ``faiss-gpu-cu12>=1.7.4``, requirements.txt:9-10, not vendored, not
    installable here).  Published semantics restated: exact float32 inner product (larger first) or
    squared L2 (smaller first), best-first, ``-1`` ids when fewer than k rows exist;
  * ``shape_results``       follows the result loop of ``VectorIndex.search`` (:226-259).

PARITY UNPINNED at the faiss boundary: the reference holds no golden vector, fixture or test for this
call and faiss cannot be run here.  Ranking is arbitrated in float64 (the value
both faiss's float32 sgemm and this build's exact re-score approximate); ties go to the lower id.
"""
from __future__ import annotations

import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def preprocess_vectors(vectors: np.ndarray, similarity_metric: str = "cosine") -> np.ndarray:
    """vector_index.py:265-282."""
    if vectors.dtype != np.float32:
        vectors = vectors.astype(np.float32)
    if not vectors.flags["C_CONTIGUOUS"]:
        vectors = np.ascontiguousarray(vectors)
    if similarity_metric == "cosine":
        norms = np.linalg.norm(vectors, axis=1, keepdims=True)
        norms = np.where(norms == 0, 1, norms)
        vectors = vectors / norms
    return vectors


def exact_scores(q: np.ndarray, x: np.ndarray, metric: str = "ip", block: int = 65536) -> np.ndarray:
    """float64-accumulated scores of float32 inputs, [nq, n]; 'ip' or 'l2' (squared distance)."""
    q64 = q.astype(np.float64)
    out = np.empty((q.shape[0], x.shape[0]), dtype=np.float64)
    for s in range(0, x.shape[0], block):
        xb = x[s:s + block].astype(np.float64)
        if metric == "ip":
            out[:, s:s + block] = q64 @ xb.T
        else:
            # direct form, no cancellation
            d = q64[:, None, :] - xb[None, :, :] if xb.shape[0] * q.shape[0] * q.shape[1] < 2e8 else None
            if d is not None:
                out[:, s:s + block] = np.einsum("qnd,qnd->qn", d, d)
            else:
                for i in range(q.shape[0]):
                    dd = xb - q64[i]
                    out[i, s:s + block] = np.einsum("nd,nd->n", dd, dd)
    return out


def flat_search(q: np.ndarray, x: np.ndarray, k: int, metric: str = "ip"):
    """Exact top-k of preprocessed float32 rows: returns (D float32 [nq,k], I int64 [nq,k]).

    Order: best first; ties by ascending id; padding id -1 with the neutral score of faiss's heaps
    (-FLT_MAX for IP, +FLT_MAX for L2).
    """
    nq, n = q.shape[0], x.shape[0]
    D = np.full((nq, k), -FLT_MAX if metric == "ip" else FLT_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if n == 0 or nq == 0:
        return D, I
    s = exact_scores(q, x, metric)
    s32 = s.astype(np.float32)  # the value reported (correctly rounded float32)
    kk = min(k, n)
    for i in range(nq):
        key = -s32[i].astype(np.float64) if metric == "ip" else s32[i].astype(np.float64)
        if kk < n:
            part = np.argpartition(key, kk - 1)[:kk]
            kth = key[part].max()
            cand = np.nonzero(key <= kth)[0]  # include every tie of the k-th value
        else:
            cand = np.arange(n)
        order = np.lexsort((cand, key[cand]))[:kk]
        sel = cand[order]
        I[i, :kk] = sel
        D[i, :kk] = s32[i, sel]
    return D, I


def near_tie_equal(I_a: np.ndarray, I_ref: np.ndarray, scores64: np.ndarray, k: int, tol: float) -> bool:
    """True when the id sets agree up to rows whose exact score is within `tol` of the k-th best."""
    for i in range(I_ref.shape[0]):
        a = set(int(v) for v in I_a[i] if v >= 0)
        b = set(int(v) for v in I_ref[i] if v >= 0)
        if a == b:
            continue
        kth = np.sort(scores64[i])[::-1][min(k, scores64.shape[1]) - 1]
        for r in a ^ b:
            if abs(scores64[i, r] - kth) > tol:
                return False
    return True


def shape_results(D: np.ndarray, I: np.ndarray, similarity_metric: str = "cosine"):
    """vector_index.py:226-259: list-of-dicts, -1 ids dropped, flat list for a single query."""
    results = []
    for qi in range(D.shape[0]):
        qr = []
        for rank in range(D.shape[1]):
            idx = I[qi][rank]
            score = D[qi][rank]
            if idx == -1:
                continue
            r = {"index": int(idx), "score": float(score), "rank": rank}
            r["similarity"] = float(score) if similarity_metric == "cosine" else 1.0 / (1.0 + float(score))
            qr.append(r)
        results.append(qr)
    if len(results) == 1:
        return results[0]
    return results
