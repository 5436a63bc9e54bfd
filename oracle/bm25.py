"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of the lexical scores that feed the fusion, reference utils/bm25_search.py:
  * ``tokenize_text``   (:237-241)  ``re.findall(r'\\b\\w+\\b', text.lower())``;
  * ``SimpleBM25``      (:16-63)    k1 1.5, b 0.75, ``idf = ln((N - n + 0.5)/(n + 0.5) + 1)``,
                                    ``score = sum_t idf * tf*(k1+1) / (tf + k1*(1 - b + b*dl/avgdl))`` where the
                                    sum runs over the query tokens WITH repetition (a repeated query token
                                    counts twice);
  * ``bm25_scores``     (:286-340)  empty query -> zeros; scores / max when max > 0.
The rank_bm25.BM25Okapi variant (different IDF) is not installable here, so only the fallback
``SimpleBM25`` the reference uses when rank_bm25 is missing (:271-283) is restated.
  * ``FieldWeightedBM25`` / ``build_field_weighted_bm25_corpus`` / ``field_weighted_bm25_scores`` (:66-234).
Pinned by tests/golden/bm25_cases.json and tests/golden/bm25_field_cases.json, produced by running the reference
file itself.
"""
from __future__ import annotations

import math
import re
from collections import Counter
from typing import Callable, Dict, List


def tokenize_text(text: str) -> List[str]:
    return re.findall(r"\b\w+\b", text.lower())


class SimpleBM25:
    def __init__(self, corpus: List[List[str]], k1: float = 1.5, b: float = 0.75):
        self.k1, self.b = k1, b
        self.doc_len = [len(d) for d in corpus]
        self.avgdl = sum(self.doc_len) / len(self.doc_len) if self.doc_len else 0
        self.doc_freqs = [Counter(d) for d in corpus]
        self.doc_count = len(corpus)
        df: Dict[str, int] = {}
        for f in self.doc_freqs:
            for t in f:
                df[t] = df.get(t, 0) + 1
        self.idf = {t: math.log((self.doc_count - n + 0.5) / (n + 0.5) + 1.0) for t, n in df.items()}

    def get_scores(self, query: List[str]) -> List[float]:
        out = []
        for i, f in enumerate(self.doc_freqs):
            s = 0.0
            dl = self.doc_len[i]
            for t in query:
                if t in f:
                    tf = f[t]
                    s += self.idf.get(t, 0) * (tf * (self.k1 + 1) / (tf + self.k1 * (1 - self.b + self.b * (dl / self.avgdl))))
            out.append(s)
        return out


class FieldWeightedBM25:
    """utils/bm25_search.py:66-146: per field its own lengths / average / IDF; total = sum_f weight_f * score_f"""

    def __init__(self, corpus: List[Dict[str, List[str]]], field_weights: Dict[str, float] = None, k1: float = 1.5,
                 b: float = 0.75):
        self.field_weights = field_weights or {"title": 2.0, "entities": 1.5, "content": 1.0}
        self.k1, self.b = k1, b
        self.doc_count = len(corpus)
        self.stats = {}
        for field in self.field_weights:
            freqs = [Counter(d.get(field, [])) for d in corpus]
            lens = [len(d.get(field, [])) for d in corpus]
            df: Dict[str, int] = {}
            for f in freqs:
                for t in f:
                    df[t] = df.get(t, 0) + 1
            idf = {t: math.log((self.doc_count - n + 0.5) / (n + 0.5) + 1.0) for t, n in df.items()}
            self.stats[field] = (freqs, lens, sum(lens) / len(lens) if lens else 0, idf)

    def get_scores(self, query: List[str]) -> List[float]:
        out = []
        for i in range(self.doc_count):
            total = 0.0
            for field, weight in self.field_weights.items():
                freqs, lens, avgdl, idf = self.stats[field]
                fs = 0.0
                for t in query:
                    if t in freqs[i]:
                        tf = freqs[i][t]
                        den = tf + self.k1 * (1 - self.b + self.b * (lens[i] / avgdl)) if avgdl > 0 else tf + self.k1
                        fs += idf.get(t, 0) * (tf * (self.k1 + 1) / den)
                total += weight * fs
            out.append(total)
        return out


def build_field_weighted_bm25_corpus(notes, field_weights=None) -> FieldWeightedBM25:
    """utils/bm25_search.py:149-187"""
    corpus = []
    for n in notes:
        ents = n.get("entities", []) or []
        corpus.append({"title": tokenize_text(n.get("title", "") or ""),
                       "entities": tokenize_text(" ".join(ents) if isinstance(ents, list) else str(ents)),
                       "content": tokenize_text(n.get("content", "") or "")})
    return FieldWeightedBM25(corpus, field_weights or {"title": 2.0, "entities": 1.5, "content": 1.0})


def field_weighted_bm25_scores(corpus: FieldWeightedBM25, docs, query: str) -> List[float]:
    """utils/bm25_search.py:190-234"""
    q = tokenize_text(query)
    if not q:
        return [0.0] * len(docs)
    scores = corpus.get_scores(q)
    scores = scores + [0.0] * (len(docs) - len(scores)) if len(scores) < len(docs) else scores[:len(docs)]
    if scores:
        m = max(scores)
        if m > 0:
            scores = [s / m for s in scores]
    return scores


def build_bm25_corpus(notes, text_fn: Callable) -> SimpleBM25:
    toks = []
    for n in notes:
        try:
            t = text_fn(n)
            toks.append(tokenize_text(t) if t else [])
        except Exception:
            toks.append([])
    return SimpleBM25(toks)


def bm25_scores(corpus: SimpleBM25, docs, query: str) -> List[float]:
    try:
        q = tokenize_text(query)
        if not q:
            return [0.0] * len(docs)
        scores = corpus.get_scores(q)
        if len(scores) < len(docs):
            scores.extend([0.0] * (len(docs) - len(scores)))
        else:
            scores = scores[:len(docs)]
        if scores:
            m = max(scores)
            if m > 0:
                scores = [s / m for s in scores]
        return scores
    except Exception:
        return [0.0] * len(docs)
