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


class BlockedTopK:
    """Streaming form of ``flat_search`` for corpora that are generated / loaded block by block (bench.py's recall
    leg at 10 M rows, the large parity tests): feed preprocessed float32 row blocks with ``push``; per block a
    float32 sgemm nominates every row within ``margin`` of the block's k-th best score, the nominees are re-scored
    with float64 accumulation (the arbiter, as in ``flat_search``) and merged into the running top-k
    (score desc / distance asc, ties by ascending id)."""

    def __init__(self, q: np.ndarray, k: int, metric: str = "ip", margin: float = 1e-4):
        self.q = np.ascontiguousarray(q, dtype=np.float32)
        self.q64 = self.q.astype(np.float64)
        self.k, self.metric, self.margin = int(k), metric, float(margin)
        nq = self.q.shape[0]
        self.S = np.zeros((nq, 0), dtype=np.float64)
        self.I = np.zeros((nq, 0), dtype=np.int64)

    def push(self, xb: np.ndarray, row0: int) -> None:
        n = xb.shape[0]
        if n == 0:
            return
        if self.metric == "ip":
            s = self.q @ xb.T
        else:
            s = -((self.q * self.q).sum(1)[:, None] - 2.0 * (self.q @ xb.T) + (xb * xb).sum(1)[None, :])
        kk = min(self.k, n)
        newS, newI = [], []
        for i in range(self.q.shape[0]):
            if kk < n:
                part = np.argpartition(-s[i], kk - 1)[:kk]
                cand = np.nonzero(s[i] >= s[i][part].min() - self.margin)[0]
            else:
                cand = np.arange(n)
            xc = xb[cand].astype(np.float64)
            if self.metric == "ip":
                e = xc @ self.q64[i]
            else:
                d = xc - self.q64[i]
                e = -np.einsum("nd,nd->n", d, d)
            newS.append(e)
            newI.append(cand.astype(np.int64) + int(row0))
        for i in range(self.q.shape[0]):
            S = np.concatenate([self.S[i], newS[i]]) if self.S.shape[1] else newS[i]
            I = np.concatenate([self.I[i], newI[i]]) if self.I.shape[1] else newI[i]
            # ranking on the float32-rounded value reported, ties by id — the order flat_search uses
            order = np.lexsort((I, -S.astype(np.float32).astype(np.float64)))[: self.k]
            newS[i], newI[i] = S[order], I[order]
        width = max(len(v) for v in newS)
        self.S = np.full((len(newS), width), -np.inf)
        self.I = np.full((len(newS), width), -1, dtype=np.int64)
        for i in range(len(newS)):
            self.S[i, : len(newS[i])] = newS[i]
            self.I[i, : len(newI[i])] = newI[i]

    def result(self):
        """(scores float64 [nq, <=k] — inner products, or MINUS squared distances for 'l2' — and ids int64)"""
        return self.S, self.I


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


def scan_image(x32: np.ndarray, bits: int = 12) -> np.ndarray:
    """The image of stored float32 rows that the device's streaming scan reads (test infrastructure: restates
    csrc/index_kernels.hpp f16_to_12 / k_add, nothing in the reference — faiss scans float32; the image only NOMINATES
    candidates, the float32 rows decide).  bits = 16: the rows rounded to float16 (round-to-nearest-even).  bits = 12: that
    float16 rounded again, to nearest-even, to its top 12 bits (sign, 5 exponent bits, 6 mantissa bits); a value that would
    round up to infinity is truncated instead.  Returned as float32."""
    h = np.asarray(x32, dtype=np.float32).astype(np.float16)
    if bits == 16:
        return h.astype(np.float32)
    if bits != 12:
        raise ValueError("bits must be 12 or 16")
    b = h.view(np.uint16).astype(np.uint32)
    r = (b + 7 + ((b >> 4) & 1)) >> 4
    would_overflow = ((r & 0x7C0) == 0x7C0) & ((b & 0x7C00) != 0x7C00)
    r = np.where(would_overflow, b >> 4, r) & 0xFFF
    return (r << 4).astype(np.uint16).view(np.float16).astype(np.float32)


def scan_image_error(x32: np.ndarray, bits: int = 12) -> float:
    """max over the rows of || image row - row || (float64): the quantity the device tracks for its certificate"""
    x = np.asarray(x32, dtype=np.float32)
    d = scan_image(x, bits).astype(np.float64) - x.astype(np.float64)
    return float(np.sqrt((d * d).sum(axis=1)).max()) if len(x) else 0.0
