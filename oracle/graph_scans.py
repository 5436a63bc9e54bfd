"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of the graph layer's two dense scans:
  * ``similarity_matrix``                follows ``RelationExtractor._compute_similarity_matrix``
                                         (graph/relation_extractor.py:769-782);
  * ``similarity_rank``                  follows ``_get_similarity_rank`` (:784-791);
  * ``semantic_similarity_relations``    follows ``extract_semantic_similarity_relations`` (:591-629);
  * ``find_embedding_candidates``        follows ``GraphRetriever._find_embedding_candidates``
                                         (graph/graph_retriever.py:153-170) up to the index → note-id mapping.

PINNED: tests/test_oracle_golden.py checks the first three against tests/golden/similarity_relation_cases.json and
``find_embedding_candidates`` against tests/golden/embedding_candidates_cases.json, both produced by
tests/golden/make_golden.py running the reference's own relation_extractor.py / graph_retriever.py (their
package-level imports satisfied by stand-in modules; networkx is installed).
"""
from __future__ import annotations

from typing import Any, Dict, List

import numpy as np


def similarity_matrix(embeddings: np.ndarray) -> np.ndarray:
    """relation_extractor.py:769-782: row-normalise (zero norms -> 1), X X^T, zero diagonal."""
    norms = np.linalg.norm(embeddings, axis=1, keepdims=True)
    norms = np.where(norms == 0, 1, norms)
    normalized = embeddings / norms
    sim = np.dot(normalized, normalized.T)
    np.fill_diagonal(sim, 0)
    return sim


def similarity_rank(source_index: int, target_index: int, sim: np.ndarray) -> int:
    """relation_extractor.py:784-791: 1-based position of the target in the source row sorted descending."""
    order = np.argsort(sim[source_index])[::-1]
    return int(np.where(order == target_index)[0][0] + 1)


def semantic_similarity_relations(atomic_notes: List[Dict[str, Any]], embeddings: np.ndarray, threshold: float = 0.7,
                                  weight: float = 0.5) -> List[Dict[str, Any]]:
    """relation_extractor.py:591-629 (threshold = graph.similarity_threshold, weight = graph.weights.semantic_similarity)."""
    relations: List[Dict[str, Any]] = []
    if embeddings.shape[0] != len(atomic_notes):
        return relations
    sim = similarity_matrix(embeddings)
    n = len(atomic_notes)
    for i in range(n):
        for j in range(i + 1, n):
            s = sim[i, j]
            if s >= threshold:
                relations.append({
                    "source_id": atomic_notes[i].get("note_id"),
                    "target_id": atomic_notes[j].get("note_id"),
                    "relation_type": "semantic_similarity",
                    "weight": weight * s,
                    "metadata": {"cosine_similarity": float(s), "similarity_rank": similarity_rank(i, j, sim)},
                })
    return relations


def find_embedding_candidates(embeddings: np.ndarray, query_embedding: np.ndarray, top_k: int = 15) -> np.ndarray:
    """graph_retriever.py:158-162: un-normalised inner products, indices of the top_k largest, best first."""
    similarities = np.dot(embeddings, query_embedding)
    return np.argsort(similarities)[-top_k:][::-1]
