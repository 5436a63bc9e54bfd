"""MI355X counterpart of the reference's ``utils/bm25_search.py`` scoring path.

Same function names and return conventions as the reference for the part the fusion consumes —
``tokenize_text`` (:237-241), ``build_bm25_corpus(notes, text_fn)`` (:244-283) and
``bm25_scores(corpus, docs, query)`` (:286-340, scores normalised to [0, 1] by the maximum, zeros for an empty
query or on error) — with the per-query scoring loop of ``SimpleBM25.get_scores`` (:43-63) running on the
device (``anr_bm25_*``: CSR postings scatter, float64, additions in the reference's order -> bit-identical).
Tokenisation, vocabulary and the posting weights are host work.  To use it inside the reference tree copy this
file over ``utils/bm25_search.py``; the ``rank_bm25`` variant of the reference (different IDF)
is not reproduced — this is the ``SimpleBM25`` fallback the reference uses when rank_bm25 is absent.
No CPU scoring fallback: without the HIP library / a device, building the corpus raises.
"""
from __future__ import annotations

import ctypes as C
import math
import re
from collections import Counter
from typing import Optional, Any, Callable, Dict, List, Sequence

import numpy as np

from . import _lib
from .compat import logger

RANK_BM25_AVAILABLE = False


def tokenize_text(text: str) -> List[str]:
    return re.findall(r"\b\w+\b", text.lower())


class DeviceBM25:
    """SimpleBM25 (reference utils/bm25_search.py:16-63) with device-resident postings."""

    def __init__(self, corpus: List[List[str]], k1: float = 1.5, b: float = 0.75, device: int = 0):
        self.k1, self.b = k1, b
        self.device = int(device)
        self.doc_len = [len(d) for d in corpus]
        self.doc_count = len(corpus)
        self.avgdl = sum(self.doc_len) / len(self.doc_len) if self.doc_len else 0
        self.vocab: Dict[str, int] = {}
        post_doc: List[List[int]] = []
        post_tf: List[List[int]] = []
        for di, doc in enumerate(corpus):
            for term, tf in Counter(doc).items():
                ti = self.vocab.get(term)
                if ti is None:
                    ti = len(post_doc)
                    self.vocab[term] = ti
                    post_doc.append([])
                    post_tf.append([])
                post_doc[ti].append(di)
                post_tf[ti].append(tf)
        n_terms = len(post_doc)
        self.idf = np.empty(n_terms, dtype=np.float64)
        indptr = np.zeros(n_terms + 1, dtype=np.int64)
        for ti in range(n_terms):
            n = len(post_doc[ti])
            # math.log, exactly as the reference (:41)
            self.idf[ti] = math.log((self.doc_count - n + 0.5) / (n + 0.5) + 1.0)
            indptr[ti + 1] = indptr[ti] + n
        docs = np.fromiter((d for lst in post_doc for d in lst), dtype=np.int32, count=int(indptr[-1]))
        tf = np.fromiter((t for lst in post_tf for t in lst), dtype=np.float64, count=int(indptr[-1]))
        if len(docs):
            dl = np.asarray(self.doc_len, dtype=np.float64)[docs]
            term_of = np.repeat(np.arange(n_terms), np.diff(indptr))
            # the reference's expression, operation by operation in float64 (:56-58)
            numerator = tf * (self.k1 + 1)
            denominator = tf + self.k1 * (1 - self.b + self.b * (dl / self.avgdl))
            weights = self.idf[term_of] * (numerator / denominator)
        else:
            weights = np.zeros(0, dtype=np.float64)
        self._post_len = np.diff(indptr)  # documents per term: sizes the sparse rows (scores_sparse_device)
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.anr_bm25_create(int(device), self.doc_count, n_terms, indptr.ctypes.data,
                                             docs.ctypes.data, weights.ctypes.data,
                                             C.byref(h)), "anr_bm25_create")
        self._h = h

    def _encode_queries(self, queries: Sequence[Sequence[str]]):
        indptr = np.zeros(len(queries) + 1, dtype=np.int64)
        terms: List[int] = []
        for i, q in enumerate(queries):
            for tok in q:
                ti = self.vocab.get(tok)
                if ti is not None:  # a token no document holds adds nothing (:51)
                    terms.append(ti)
            indptr[i + 1] = len(terms)
        return indptr, np.asarray(terms, dtype=np.int32)

    def scores_batch(self, queries: Sequence[Sequence[str]], normalize: bool = False) -> np.ndarray:
        """[nq, n_docs] float64"""
        indptr, terms = self._encode_queries(queries)
        out = np.zeros((len(queries), self.doc_count), dtype=np.float64)
        _lib.check(self._lib.anr_bm25_scores(self._h, len(queries), indptr.ctypes.data,
                                             terms.ctypes.data, int(bool(normalize)),
                                             out.ctypes.data), "anr_bm25_scores")
        return out

    def scores_device(self, queries: Sequence[Sequence[str]], normalize: bool = True):
        """[nq, n_docs] float64 left in device memory (an ``anorag_hip.fusion.DeviceArray``): the array source of
        ``anr_fuse_dense`` — the N-vector never crosses PCIe"""
        from .fusion import DeviceArray
        indptr, terms = self._encode_queries(queries)
        out = DeviceArray(len(queries), self.doc_count, np.float64, self.device)
        out.row_max = DeviceArray(len(queries), 1, np.float64, self.device)  # by-product: each row's maximum
        _lib.check(self._lib.anr_bm25_scores_dev(self._h, len(queries), indptr.ctypes.data,
                                                 terms.ctypes.data, int(bool(normalize)),
                                                 C.c_void_p(out.ptr), C.c_void_p(out.row_max.ptr)), "anr_bm25_scores_dev")
        return out

    SPARSE_CAP = 6144        # documents of a row that one LDS hash table holds (kSpMaxCap, csrc/bm25.hip): rows unordered
    SPARSE_CAP_MAX = 65536   # ... of a row cut into document-range slices (kSpMaxCapBig): rows ascending by id

    def scores_sparse_device(self, queries: Sequence[Sequence[str]], normalize: bool = True, cap: Optional[int] = None,
                             allow_overflow: bool = False):
        """the same scores as ``scores_device`` in SPARSE form (``anorag_hip.fusion.SparseRows``): per query the
        documents its postings touch and their scores, left in device memory for ``fuse_dense`` — the N-vector is never
        formed.  ``cap`` (documents per row, at most ``SPARSE_CAP_MAX``) defaults to what the batch needs: the largest
        summed posting length of its queries (an upper bound of the documents a query touches), ``SPARSE_CAP`` when that
        fits one table.  A query that touches more than `cap` documents cannot be held: with ``allow_overflow`` its row
        is marked (``counts[i] == -1``: score that query with ``scores_device``), otherwise None is returned."""
        from .fusion import SparseRows
        indptr, terms = self._encode_queries(queries)
        nq = len(queries)
        if cap is None:
            need = 0
            if len(terms):
                csum = np.concatenate(([0], np.cumsum(self._post_len[terms])))
                need = int((csum[indptr[1:]] - csum[indptr[:-1]]).max()) if nq else 0
            cap = self.SPARSE_CAP if need <= self.SPARSE_CAP else min(self.SPARSE_CAP_MAX, -(-need // 1024) * 1024)
        out = SparseRows(nq, self.doc_count, cap, self.device)
        cnt = np.empty((nq,), dtype=np.int32)
        _lib.check(self._lib.anr_bm25_sparse_dev(self._h, nq, indptr.ctypes.data,
                                                 terms.ctypes.data, int(bool(normalize)), int(cap),
                                                 C.c_void_p(out.ids_ptr), C.c_void_p(out.scores_ptr),
                                                 C.c_void_p(out.count_ptr), C.c_void_p(out.max_ptr),
                                                 cnt.ctypes.data), "anr_bm25_sparse_dev")
        if nq and int(cnt.min()) < 0 and not allow_overflow:
            out.free()
            return None
        out.counts = cnt
        return out

    def nonzero_batch(self, queries: Sequence[Sequence[str]], normalize: bool = True, cap: int = 4096):
        """per query: (doc ids, scores) of the documents with a non-zero score, best first, at most `cap`"""
        indptr, terms = self._encode_queries(queries)
        nq = len(queries)
        docs = np.empty((nq, cap), dtype=np.int32)
        sc = np.empty((nq, cap), dtype=np.float64)
        cnt = np.empty((nq,), dtype=np.int32)
        _lib.check(self._lib.anr_bm25_nonzero(self._h, nq, indptr.ctypes.data,
                                              terms.ctypes.data, int(bool(normalize)), int(cap),
                                              docs.ctypes.data, sc.ctypes.data,
                                              cnt.ctypes.data), "anr_bm25_nonzero")
        out = []
        for i in range(nq):
            n = min(int(cnt[i]), cap)
            order = np.lexsort((docs[i, :n], -sc[i, :n]))
            out.append((docs[i, :n][order].astype(np.int64), sc[i, :n][order]))
        return out

    def get_scores(self, query: List[str]) -> List[float]:
        """raw (un-normalised) scores of one tokenised query, like SimpleBM25.get_scores"""
        return self.scores_batch([query], normalize=False)[0].tolist()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.anr_bm25_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class DeviceFieldWeightedBM25:
    """FieldWeightedBM25 (reference utils/bm25_search.py:66-146): one BM25 per field (title / entities / content by
    default) with its own document lengths, average length and IDF, combined as sum_f weight_f * score_f.  Each field
    is a ``DeviceBM25`` (same posting-weight expression, same addition order); the fields are added up on the device
    in the order of ``field_weights`` — float64, bit-identical to the reference's ``get_scores``."""

    def __init__(self, corpus: List[Dict[str, List[str]]], field_weights: Dict[str, float] = None, k1: float = 1.5,
                 b: float = 0.75, device: int = 0):
        self.corpus = corpus
        self.field_weights = field_weights or {"title": 2.0, "entities": 1.5, "content": 1.0}
        self.k1, self.b = k1, b
        self.doc_count = len(corpus)
        self.device = int(device)
        self.fields: Dict[str, DeviceBM25] = {}
        for field in self.field_weights:
            self.fields[field] = DeviceBM25([doc.get(field, []) for doc in corpus], k1, b, device)

    def scores_device(self, queries: Sequence[Sequence[str]], normalize: bool = False):
        """[nq, n_docs] float64 in device memory (``DeviceArray``): feeds ``anr_fuse_dense`` as is"""
        from .fusion import DeviceArray
        nq = len(queries)
        parts = [bm.scores_device(queries, normalize=False) for bm in self.fields.values()]
        out = DeviceArray(nq, self.doc_count, np.float64, self.device)
        out.row_max = DeviceArray(nq, 1, np.float64, self.device)
        ptrs = (C.c_void_p * len(parts))(*[p.ptr for p in parts])
        w = np.asarray([float(v) for v in self.field_weights.values()], dtype=np.float64)
        try:
            _lib.check(_lib.load().anr_bm25_combine_fields(self.device, len(parts), ptrs, w.ctypes.data, nq,
                                                           self.doc_count, int(bool(normalize)), C.c_void_p(out.ptr),
                                                           C.c_void_p(out.row_max.ptr)),
                       "anr_bm25_combine_fields")
        finally:
            for p in parts:
                p.free()
        return out

    def scores_batch(self, queries: Sequence[Sequence[str]], normalize: bool = False) -> np.ndarray:
        if self.doc_count == 0 or not queries:
            return np.zeros((len(queries), self.doc_count), dtype=np.float64)
        d = self.scores_device(queries, normalize)
        try:
            return d.numpy()
        finally:
            d.free()

    def get_scores(self, query: List[str]) -> List[float]:
        return self.scores_batch([query], normalize=False)[0].tolist()

    def close(self):
        for bm in self.fields.values():
            bm.close()


def build_field_weighted_bm25_corpus(notes: List[Dict[str, Any]], field_weights: Dict[str, float] = None) -> DeviceFieldWeightedBM25:
    """reference utils/bm25_search.py:149-187: title / entities / content token lists per note"""
    if field_weights is None:
        field_weights = {"title": 2.0, "entities": 1.5, "content": 1.0}
    field_corpus = []
    for note in notes:
        entities = note.get("entities", []) or []
        entities_text = " ".join(entities) if isinstance(entities, list) else str(entities)
        field_corpus.append({"title": tokenize_text(note.get("title", "") or ""),
                             "entities": tokenize_text(entities_text),
                             "content": tokenize_text(note.get("content", "") or "")})
    return DeviceFieldWeightedBM25(field_corpus, field_weights)


def field_weighted_bm25_scores(corpus: DeviceFieldWeightedBM25, docs: List[Dict[str, Any]], query: str) -> List[float]:
    """reference utils/bm25_search.py:190-234"""
    try:
        tokens = tokenize_text(query)
        if not tokens:
            return [0.0] * len(docs)
        scores = corpus.scores_batch([tokens], normalize=False)[0].tolist()
        if len(scores) != len(docs):
            scores = scores + [0.0] * (len(docs) - len(scores)) if len(scores) < len(docs) else scores[:len(docs)]
        if scores:
            m = max(scores)
            if m > 0:
                scores = [s / m for s in scores]
        return scores
    except Exception as e:
        logger.error(f"Error calculating field-weighted BM25 scores: {e}")
        return [0.0] * len(docs)


def build_bm25_corpus(notes: List[Dict[str, Any]], text_fn: Callable[[Dict[str, Any]], str]) -> DeviceBM25:
    tokenized = []
    for note in notes:
        try:
            text = text_fn(note)
            tokenized.append(tokenize_text(text) if text else [])
        except Exception as e:
            logger.warning(f"Error extracting text from note: {e}")
            tokenized.append([])
    return DeviceBM25(tokenized)


def bm25_scores(corpus: Any, docs: List[Dict[str, Any]], query: str) -> List[float]:
    try:
        tokens = tokenize_text(query)
        if not tokens:
            return [0.0] * len(docs)
        if not hasattr(corpus, "get_scores"):
            logger.error("Invalid corpus object")
            return [0.0] * len(docs)
        if isinstance(corpus, DeviceBM25):
            scores = corpus.scores_batch([tokens], normalize=False)[0].tolist()
        else:
            scores = list(corpus.get_scores(tokens))
        if len(scores) != len(docs):
            scores = scores + [0.0] * (len(docs) - len(scores)) if len(scores) < len(docs) else scores[:len(docs)]
        if scores:
            m = max(scores)
            if m > 0:
                scores = [s / m for s in scores]
        return scores
    except Exception as e:
        logger.error(f"Error calculating BM25 scores: {e}")
        return [0.0] * len(docs)
