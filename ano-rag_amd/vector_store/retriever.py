"""MI355X drop-in for the reference's ``vector_store/retriever.py`` (class ``VectorRetriever``).

Owns the encoder (``EmbeddingManager``), the index (``VectorIndex``) and the notes, exactly like the
reference (retriever.py:29-1034): same public methods, argument meaning, return shapes, attribute names and
error conventions (public methods never raise; they log and return ``False`` / ``[]`` / ``[[] ...]``).
All arithmetic is behind the two owned objects, i.e. on the device; this file is result shaping only.

Reference quirks kept on purpose: ``threshold or default`` treats an explicit ``0.0`` as
unset (retriever.py:200, :374); no ``id_to_index`` attribute exists; ``get_retrieval_stats`` calls the
non-existent ``embedding_manager.get_embedding_stats`` when embeddings are held (:763) and raises there.
"""
from __future__ import annotations

import os
import re
from typing import Any, Callable, Dict, List, Optional

import numpy as np

from anorag_hip.compat import FileUtils, config, default_tmp, logger

from .embedding_manager import EmbeddingManager
from .vector_index import VectorIndex

try:  # same optional imports as the reference (retriever.py:15-27)
    from retrieval.hybrid_search import HybridSearcher, create_hybrid_searcher  # noqa: F401
    HYBRID_SEARCH_AVAILABLE = True
except ImportError:  # pragma: no cover
    HYBRID_SEARCH_AVAILABLE = False

try:
    from retrieval.retrieval_guardrail import create_retrieval_guardrail  # type: ignore
    GUARDRAIL_AVAILABLE = True
except ImportError:
    GUARDRAIL_AVAILABLE = False


class _Batcher:
    """stand-in for the reference's unused ``utils.BatchProcessor`` attribute (retriever.py:71-74)"""

    def __init__(self, batch_size, use_gpu=True):
        self.batch_size = batch_size
        self.use_gpu = use_gpu


def _text_of(content) -> str:
    if isinstance(content, dict):
        return content.get("text", "") or content.get("content", "") or str(content)
    return content if isinstance(content, str) else str(content)


class VectorRetriever:
    """Vector retriever: embedding manager + vector index + the notes (reference retriever.py:29)."""

    def __init__(self):
        self.embedding_manager = EmbeddingManager()
        self.vector_index = VectorIndex(self.embedding_manager.embedding_dim)

        self.top_k = config.get("vector_store.top_k", 20)
        self.similarity_threshold = config.get("vector_store.similarity_threshold", 0.5)
        self.batch_size = config.get("vector_store.batch_size", 32)

        enh = config.get("vector_store.retriever_enhancement", {}) or {}
        self.default_topk_multiplier = enh.get("topk_multiplier", 3.0)
        self.must_have_terms_penalty = enh.get("must_have_terms_penalty", 0.6)
        self.entity_boost_factor = enh.get("entity_boost_factor", 1.2)
        self.predicate_boost_factor = enh.get("predicate_boost_factor", 1.15)
        self.enable_filter_logging = enh.get("enable_filter_logging", True)
        self.default_must_have_terms = enh.get("default_must_have_terms", [])
        self.default_boost_entities = enh.get("default_boost_entities", [])
        self.default_boost_predicates = enh.get("default_boost_predicates", [])

        self.atomic_notes: List[Dict[str, Any]] = []
        self.note_embeddings: Optional[np.ndarray] = None
        self.note_id_to_index: Dict[Any, int] = {}
        self.index_to_note_id: Dict[int, Any] = {}

        self.data_dir = config.get("storage.vector_store_path") or default_tmp("anorag_vector_store")
        FileUtils.ensure_dir(self.data_dir)
        self.batch_processor = _Batcher(self.batch_size, config.get("performance.use_gpu", True))

        # TF-IDF "BM25 fallback" (retriever.py:76-82, :924-1002)
        self.bm25_enabled = config.get("vector_store.bm25_fallback.enabled", True)
        self.bm25_k1 = config.get("vector_store.bm25_fallback.k1", 1.2)
        self.bm25_b = config.get("vector_store.bm25_fallback.b", 0.75)
        self.tfidf_vectorizer = None
        self.tfidf_matrix = None
        self.processed_texts: List[str] = []

        self.hybrid_searcher = None
        self.hybrid_config = config.get("vector_store.hybrid_search", {}) or {}
        self.enable_hybrid_search = bool(self.hybrid_config.get("enable_hybrid_search", False)) and HYBRID_SEARCH_AVAILABLE
        if self.enable_hybrid_search:
            try:
                self.hybrid_searcher = create_hybrid_searcher(config)
            except Exception as e:
                logger.warning(f"Failed to initialize hybrid searcher: {e}")
                self.enable_hybrid_search = False

        self.retrieval_guardrail = None
        self.guardrail_config = config.get("retrieval_guardrail", {}) or {}
        self.enable_guardrail = bool(self.guardrail_config.get("enabled", True)) and GUARDRAIL_AVAILABLE
        if self.enable_guardrail:
            try:
                self.retrieval_guardrail = create_retrieval_guardrail(config)
            except Exception as e:
                logger.warning(f"Failed to initialize retrieval guardrail: {e}")
                self.enable_guardrail = False

        self._validate_embedding_consistency()
        logger.info("VectorRetriever initialized (HIP dense path)")

    # -- building --------------------------------------------------------------------------------
    def build_index(self, atomic_notes: List[Dict[str, Any]], force_rebuild: bool = False,
                    save_index: bool = True) -> bool:
        if not atomic_notes:
            logger.warning("No atomic notes provided for indexing")
            return False
        try:
            if not force_rebuild and self._can_load_existing_index(atomic_notes):
                return True
            self.atomic_notes = atomic_notes
            self._build_id_mappings()
            if not self._build_device_resident(atomic_notes):
                self.note_embeddings = self.embedding_manager.encode_atomic_notes(atomic_notes, include_metadata=True)
                if self.note_embeddings.size == 0:
                    logger.error("Failed to generate embeddings")
                    return False
                if not self.vector_index.create_index():
                    logger.error("Failed to create vector index")
                    return False
                ids = np.arange(len(atomic_notes), dtype=np.int64)
                if not self.vector_index.add_vectors(self.note_embeddings, ids):
                    logger.error("Failed to add vectors to index")
                    return False
            if self.bm25_enabled:
                self._build_bm25_index(atomic_notes)
            if self.enable_hybrid_search and self.hybrid_searcher:
                try:  # the reference calls a method HybridSearcher does not have (retriever.py:170)
                    self.hybrid_searcher.build_index(atomic_notes)
                except Exception as e:
                    logger.warning(f"Failed to build hybrid search index: {e}")
                    self.enable_hybrid_search = False
            if save_index:
                self._save_index_data()
            logger.info(f"Vector index built successfully with {len(atomic_notes)} notes")
            return True
        except Exception as e:
            logger.error(f"Failed to build vector index: {e}")
            return False

    def _build_device_resident(self, atomic_notes: List[Dict[str, Any]]) -> bool:
        """encode -> index without the host round trip (reference flow retriever.py:140-157): the encoder writes the
        embeddings into device memory, the index adds them from there, and ``note_embeddings`` (which callers index
        into, query_processor.py:293) is filled by ONE device-to-host copy.  False = not applicable, nothing changed."""
        em = self.embedding_manager
        if not hasattr(em, "encode_texts_device") or not hasattr(getattr(em, "model", None), "encode_device"):
            return False
        dev = None
        try:
            dev = em.encode_texts_device(em._assemble_note_texts(atomic_notes))
            if not self.vector_index.create_index():
                return False
            if not self.vector_index.add_vectors_device(dev, np.arange(len(atomic_notes), dtype=np.int64)):
                return False
            self.note_embeddings = dev.numpy()
            return True
        except Exception as e:
            logger.warning(f"Device-resident index build not used: {e}")
            return False
        finally:
            if dev is not None:
                dev.free()

    # -- searching -------------------------------------------------------------------------------
    def search(self, queries: List[str], top_k: Optional[int] = None, similarity_threshold: Optional[float] = None,
               include_metadata: bool = True) -> List[List[Dict[str, Any]]]:
        if not queries:
            return []
        if not self.atomic_notes or self.vector_index.total_vectors == 0:
            logger.warning("Vector index is empty")
            return [[] for _ in queries]
        top_k = top_k or self.top_k
        similarity_threshold = similarity_threshold or self.similarity_threshold
        try:
            raw = self._search_device_resident(queries, top_k)
            if raw is None:
                q_emb = self.embedding_manager.encode_queries(queries)
                if q_emb.size == 0:
                    logger.error("Failed to generate query embeddings")
                    return [[] for _ in queries]
                raw = self.vector_index.search(q_emb, top_k=top_k)
            if len(queries) == 1 and isinstance(raw, list) and raw and isinstance(raw[0], dict):
                raw = [raw]  # single query comes back flat (vector_index.py:255-257)
            out: List[List[Dict[str, Any]]] = []
            for qi, query in enumerate(queries):
                hits: List[Dict[str, Any]] = []
                for r in (raw[qi] if qi < len(raw) else []):
                    if r.get("similarity", 0) < similarity_threshold:
                        continue
                    ni = r["index"]
                    if ni >= len(self.atomic_notes):
                        continue
                    note = self.atomic_notes[ni].copy()
                    info = {"similarity": r["similarity"], "score": r["score"], "rank": r["rank"],
                            "query": query, "retrieval_method": "vector_search"}
                    if include_metadata:
                        note["retrieval_info"] = info
                    else:
                        note = {"note_id": note.get("note_id"), "content": note.get("content"),
                                "paragraph_idxs": note.get("paragraph_idxs", []), "retrieval_info": info}
                    hits.append(note)
                out.append(hits)
            logger.info(f"Search completed: {sum(len(h) for h in out)} results for {len(queries)} queries")
            return out
        except Exception as e:
            logger.error(f"Failed to search: {e}")
            return [[] for _ in queries]

    def _search_device_resident(self, queries: List[str], top_k: int):
        """query embeddings go from the encoder to the scan in device memory (reference flow retriever.py:206-216);
        None = not applicable"""
        em = self.embedding_manager
        if not hasattr(em, "encode_queries_device") or not hasattr(getattr(em, "model", None), "encode_device"):
            return None
        dev = None
        try:
            dev = em.encode_queries_device(queries)
            return self.vector_index.search_device(dev, top_k=top_k)
        except Exception:
            return None
        finally:
            if dev is not None:
                dev.free()

    def search_single(self, query: str, top_k: Optional[int] = None, similarity_threshold: Optional[float] = None,
                      include_metadata: bool = True) -> List[Dict[str, Any]]:
        res = self.search([query], top_k, similarity_threshold, include_metadata)
        return res[0] if res else []

    def hybrid_search(self, queries: List[str], top_k: Optional[int] = None,
                      similarity_threshold: Optional[float] = None, include_metadata: bool = True,
                      **kwargs) -> List[List[Dict[str, Any]]]:
        if not self.enable_hybrid_search or not self.hybrid_searcher:
            logger.warning("Hybrid search not available, falling back to vector search")
            return self.search(queries, top_k, similarity_threshold, include_metadata)
        try:  # reference retriever.py:288 — HybridSearcher has no ``search``; the except branch is what runs
            results = self.hybrid_searcher.search(queries=queries, documents=self.atomic_notes,
                                                  top_k=top_k or self.top_k,
                                                  similarity_threshold=similarity_threshold or self.similarity_threshold,
                                                  **kwargs)
            shaped = []
            for hits in results:
                if include_metadata:
                    shaped.append(list(hits))
                else:
                    shaped.append([{"note_id": h.get("note_id"), "content": h.get("content"),
                                    "paragraph_idxs": h.get("paragraph_idxs", []),
                                    "retrieval_info": h.get("retrieval_info")} for h in hits])
            return shaped
        except Exception as e:
            logger.error(f"Hybrid search failed: {e}")
            return self.search(queries, top_k, similarity_threshold, include_metadata)

    def hybrid_search_single(self, query: str, top_k: Optional[int] = None,
                             similarity_threshold: Optional[float] = None, include_metadata: bool = True,
                             **kwargs) -> List[Dict[str, Any]]:
        res = self.hybrid_search([query], top_k, similarity_threshold, include_metadata, **kwargs)
        return res[0] if res else []

    def retrieve(self, query: str, top_k: Optional[int] = None, similarity_threshold: Optional[float] = None,
                 filter_fn: Optional[Callable] = None, must_have_terms: Optional[List[str]] = None,
                 boost_entities: Optional[List[str]] = None, boost_predicates: Optional[List[str]] = None,
                 topk_multiplier: float = 3.0, include_metadata: bool = True,
                 enable_guardrail: Optional[bool] = None) -> List[Dict[str, Any]]:
        """Over-fetch, optional filter, term penalty / entity + predicate boosts, threshold, sort, cut
        (reference retriever.py:339-512)."""
        if not query:
            return []
        if not self.atomic_notes or self.vector_index.total_vectors == 0:
            logger.warning("Vector index is empty")
            return []
        top_k = top_k or self.top_k
        similarity_threshold = similarity_threshold or self.similarity_threshold
        try:
            mult = topk_multiplier if topk_multiplier is not None else self.default_topk_multiplier
            cands = self.search_single(query=query, top_k=int(top_k * mult), similarity_threshold=0.0,
                                       include_metadata=include_metadata)
            if not cands:
                return []
            if filter_fn:
                kept = []
                for c in cands:
                    try:
                        if filter_fn(c):
                            kept.append(c)
                    except Exception as e:
                        logger.warning(f"Filter function failed for candidate: {e}")
                        kept.append(c)  # a failing filter keeps the candidate
                cands = kept
            adjusted = []
            for c in cands:
                low = _text_of(c.get("content", "")).lower()
                base = c.get("retrieval_info", {}).get("similarity", 0.0)
                sim = base
                notes = []
                if must_have_terms and not any(t.lower() in low for t in must_have_terms):
                    sim *= self.must_have_terms_penalty
                    notes.append("downweighted_missing_terms")
                if boost_entities:
                    hit = [e for e in boost_entities if e.lower() in low]
                    if hit:
                        sim *= self.entity_boost_factor
                        notes.append(f"boosted_entities_{len(hit)}")
                if boost_predicates:
                    hit = [p for p in boost_predicates if p.lower() in low]
                    if hit:
                        sim *= self.predicate_boost_factor
                        notes.append(f"boosted_predicates_{len(hit)}")
                cc = c.copy()
                if "retrieval_info" in cc:
                    info = cc["retrieval_info"].copy()
                    info["similarity"] = sim
                    info["original_similarity"] = base
                    info["adjustments"] = notes
                    cc["retrieval_info"] = info
                adjusted.append(cc)
            final = [c for c in adjusted if c.get("retrieval_info", {}).get("similarity", 0.0) >= similarity_threshold]
            final.sort(key=lambda x: x.get("retrieval_info", {}).get("similarity", 0.0), reverse=True)
            return final[:top_k]
        except Exception as e:
            logger.error(f"Enhanced retrieval failed: {e}")
            return self.search_single(query, top_k, similarity_threshold, include_metadata)

    def score_candidates(self, queries: List[str], candidates: List[List[Any]]) -> List[List[float]]:
        """EXTENSION (not in the reference; opt-in): cosine similarity of query i with each of its candidate notes
        (note ids) from the embeddings already on the device — what query_processor.py:3492-3589 obtains by
        re-encoding every candidate's text.  Not wired into any reference call path: using stored embeddings
        instead of re-encoded text changes the scores.  Unknown ids score 0.0; negative
        similarities are clamped to 0.0 as there."""
        if not queries:
            return []
        per = max((len(c) for c in candidates), default=0)
        if per == 0 or self.vector_index.index is None:
            return [[0.0] * len(c) for c in candidates]
        q_emb = self.embedding_manager.encode_queries(queries)
        ids = np.full((len(queries), per), -1, dtype=np.int64)
        for i, cand in enumerate(candidates):
            for j, nid in enumerate(cand):
                ids[i, j] = self.note_id_to_index.get(nid, -1)
        sims = self.vector_index.index.score_rows(q_emb, ids)
        return [[0.0 if (ids[i, j] < 0 or not np.isfinite(sims[i, j])) else max(0.0, float(sims[i, j]))
                 for j in range(len(cand))] for i, cand in enumerate(candidates)]

    # -- incremental maintenance --------------------------------------------------------------------
    def add_notes(self, new_notes: List[Dict[str, Any]], rebuild_index: bool = False) -> bool:
        if not new_notes:
            return True
        try:
            if rebuild_index or not self.atomic_notes:
                return self.build_index(self.atomic_notes + new_notes, force_rebuild=True)
            start = len(self.atomic_notes)
            self.atomic_notes.extend(new_notes)
            self._build_id_mappings()
            emb = self.embedding_manager.encode_atomic_notes(new_notes, include_metadata=True)
            if emb.size == 0:
                logger.error("Failed to generate embeddings for new notes")
                return False
            if not self.vector_index.add_vectors(emb, np.arange(start, start + len(new_notes), dtype=np.int64)):
                logger.error("Failed to add new vectors to index")
                return False
            self.note_embeddings = emb if self.note_embeddings is None else np.vstack([self.note_embeddings, emb])
            return True
        except Exception as e:
            logger.error(f"Failed to add notes: {e}")
            return False

    def remove_notes(self, note_ids: List[str]) -> bool:
        if not note_ids:
            return True
        try:
            rows = sorted({self.note_id_to_index[n] for n in note_ids if n in self.note_id_to_index}, reverse=True)
            if not rows:
                logger.warning("No matching notes found to remove")
                return True
            for r in rows:
                if r < len(self.atomic_notes):
                    del self.atomic_notes[r]
            return self.build_index(self.atomic_notes, force_rebuild=True)  # full rebuild (retriever.py:582-592)
        except Exception as e:
            logger.error(f"Failed to remove notes: {e}")
            return False

    def update_note(self, note_id: str, updated_note: Dict[str, Any]) -> bool:
        try:
            if note_id not in self.note_id_to_index:
                logger.warning(f"Note {note_id} not found")
                return False
            row = self.note_id_to_index[note_id]
            self.atomic_notes[row] = updated_note
            emb = self.embedding_manager.encode_atomic_notes([updated_note], include_metadata=True)
            if emb.size == 0:
                logger.error("Failed to generate embedding for updated note")
                return False
            if self.note_embeddings is not None:
                self.note_embeddings[row] = emb[0]
            return self.build_index(self.atomic_notes, force_rebuild=True)
        except Exception as e:
            logger.error(f"Failed to update note {note_id}: {e}")
            return False

    # -- lookups -------------------------------------------------------------------------------------
    def get_note_by_id(self, note_id: str) -> Optional[Dict[str, Any]]:
        row = self.note_id_to_index.get(note_id)
        if row is not None and row < len(self.atomic_notes):
            return self.atomic_notes[row]
        return None

    def get_notes_by_ids(self, note_ids: List[str]) -> List[Dict[str, Any]]:
        return [n for n in (self.get_note_by_id(i) for i in note_ids) if n]

    def get_similar_notes(self, note_id: str, top_k: int = 10, exclude_self: bool = True) -> List[Dict[str, Any]]:
        note = self.get_note_by_id(note_id)
        if not note or not note.get("content", ""):
            return []
        res = self.search_single(note.get("content", ""), top_k=top_k + (1 if exclude_self else 0))
        if exclude_self:
            res = [r for r in res if r.get("note_id") != note_id]
        return res[:top_k]

    def _build_id_mappings(self):
        self.note_id_to_index = {}
        self.index_to_note_id = {}
        for row, note in enumerate(self.atomic_notes):
            nid = note.get("note_id")
            if nid:
                self.note_id_to_index[nid] = row
                self.index_to_note_id[row] = nid

    # -- persistence (reference retriever.py:680-749) ------------------------------------------------
    def _can_load_existing_index(self, atomic_notes: List[Dict[str, Any]]) -> bool:
        files = [f for f in os.listdir(self.data_dir) if f.endswith(".faiss")]
        notes_file = os.path.join(self.data_dir, "atomic_notes.json")
        if not files or not os.path.exists(notes_file):
            return False
        try:
            old = FileUtils.read_json(notes_file)
            if len(old) != len(atomic_notes):
                return False
            if old and atomic_notes and old[0].get("note_id") != atomic_notes[0].get("note_id"):
                return False
            if self.vector_index.load_index(files[0]):
                self.atomic_notes = old
                self._build_id_mappings()
                emb_file = os.path.join(self.data_dir, "note_embeddings.npz")
                if os.path.exists(emb_file):
                    self.note_embeddings = np.load(emb_file)["embeddings"]
                return True
        except Exception as e:
            logger.warning(f"Failed to load existing index: {e}")
        return False

    def _save_index_data(self):
        try:
            self.vector_index.save_index()
            FileUtils.write_json(self.atomic_notes, os.path.join(self.data_dir, "atomic_notes.json"))
            if self.note_embeddings is not None:
                np.savez_compressed(os.path.join(self.data_dir, "note_embeddings.npz"), embeddings=self.note_embeddings)
            FileUtils.write_json({"note_id_to_index": self.note_id_to_index, "index_to_note_id": self.index_to_note_id},
                                 os.path.join(self.data_dir, "id_mappings.json"))
        except Exception as e:
            logger.error(f"Failed to save index data: {e}")

    # -- stats / tuning ------------------------------------------------------------------------------
    def get_retrieval_stats(self) -> Dict[str, Any]:
        stats = {"total_notes": len(self.atomic_notes), "embedding_dim": self.embedding_manager.embedding_dim,
                 "model_name": self.embedding_manager.model_name, "index_stats": self.vector_index.get_index_stats(),
                 "top_k": self.top_k, "similarity_threshold": self.similarity_threshold}
        if self.note_embeddings is not None:
            stats["embedding_stats"] = self.embedding_manager.get_embedding_stats(self.note_embeddings)
        return stats

    def optimize_retrieval(self, test_queries: List[str], ground_truth: List[List[str]],
                           target_recall: float = 0.9) -> Dict[str, Any]:
        if not test_queries or not ground_truth:
            logger.warning("No test data provided for optimization")
            return {}
        gt_rows = [[self.note_id_to_index[n] for n in ids if n in self.note_id_to_index] for ids in ground_truth]
        q_emb = self.embedding_manager.encode_queries(test_queries)
        index_opt = self.vector_index.optimize_search_params(q_emb, np.array(gt_rows, dtype=object), target_recall)
        best_thr, best_f1 = self.similarity_threshold, 0.0
        for thr in (0.1, 0.3, 0.5, 0.7, 0.9):
            f1 = self._calculate_f1_score(self.search(test_queries, similarity_threshold=thr), ground_truth)
            if f1 > best_f1:
                best_f1, best_thr = f1, thr
        self.similarity_threshold = best_thr
        return {"index_optimization": index_opt, "best_similarity_threshold": best_thr, "best_f1_score": best_f1,
                "target_recall": target_recall}

    def _calculate_f1_score(self, search_results: List[List[Dict[str, Any]]], ground_truth: List[List[str]]) -> float:
        if not search_results or not ground_truth:
            return 0.0
        total, n = 0.0, 0
        for hits, gt in zip(search_results, ground_truth):
            if not gt:
                continue
            got, want = {h.get("note_id") for h in hits}, set(gt)
            if not got or not want:
                f1 = 1.0 if (not got and not want) else 0.0
            else:
                p, r = len(got & want) / len(got), len(got & want) / len(want)
                f1 = 0.0 if p + r == 0 else 2 * p * r / (p + r)
            total += f1
            n += 1
        return total / n if n else 0.0

    def clear_index(self):
        self.atomic_notes = []
        self.note_embeddings = None
        self.note_id_to_index = {}
        self.index_to_note_id = {}
        if self.vector_index:
            self.vector_index.reset_index()

    def cleanup(self):
        if self.embedding_manager:
            self.embedding_manager.cleanup()
        if self.vector_index:
            self.vector_index.cleanup()
        if self.hybrid_searcher:
            try:
                self.hybrid_searcher.cleanup()
            except Exception as e:
                logger.warning(f"Failed to cleanup hybrid searcher: {e}")
        if self.retrieval_guardrail:
            try:
                self.retrieval_guardrail.stats.clear()
            except Exception as e:
                logger.warning(f"Failed to cleanup retrieval guardrail: {e}")

    def _validate_embedding_consistency(self):
        strict = (config.get("model_consistency", {}) or {}).get("violation_handling", {}).get("strict_mode", False)
        try:
            ok, details = self.embedding_manager.validate_model_consistency()
            if not ok:
                logger.warning(f"Embedding model consistency check failed: {details}")
                if strict:
                    raise RuntimeError(f"Model consistency violation: {details}")
        except Exception as e:
            logger.error(f"Error during embedding consistency validation: {e}")
            if strict:
                raise

    def get_embedding_model_info(self) -> Dict[str, Any]:
        try:
            return self.embedding_manager.get_model_info()
        except Exception as e:
            return {"error": str(e)}

    # -- TF-IDF fallback, named BM25 in the reference (retriever.py:924-1002) --------------------------
    def _build_bm25_index(self, atomic_notes: List[Dict[str, Any]]) -> None:
        try:
            from sklearn.feature_extraction.text import TfidfVectorizer
            texts = [self._preprocess_text(n.get("content", "")) if n.get("content", "") else "" for n in atomic_notes]
            self.processed_texts = texts
            self.tfidf_vectorizer = TfidfVectorizer(lowercase=True, stop_words="english", max_features=10000,
                                                    ngram_range=(1, 2))
            if texts:
                self.tfidf_matrix = self.tfidf_vectorizer.fit_transform(texts)
        except Exception as e:
            logger.error(f"Failed to build BM25 index: {e}")
            self.bm25_enabled = False

    def _preprocess_text(self, text: str) -> str:
        text = re.sub(r"[^a-zA-Z0-9\s]", " ", text)
        return re.sub(r"\s+", " ", text).strip().lower()

    def _bm25_search(self, query: str, top_k: int = 20) -> List[Dict[str, Any]]:
        if not self.bm25_enabled or not self.tfidf_vectorizer or self.tfidf_matrix is None:
            return []
        try:
            from sklearn.metrics.pairwise import cosine_similarity
            qv = self.tfidf_vectorizer.transform([self._preprocess_text(query)])
            sims = cosine_similarity(qv, self.tfidf_matrix).flatten()
            out = []
            for row in np.argsort(sims)[::-1][:top_k]:
                if row < len(self.atomic_notes) and sims[row] > 0:
                    note = self.atomic_notes[row].copy()
                    note["retrieval_info"] = {"similarity": float(sims[row]), "score": float(sims[row]),
                                              "rank": len(out) + 1, "query": query,
                                              "retrieval_method": "bm25_fallback"}
                    out.append(note)
            return out
        except Exception as e:
            logger.error(f"BM25 search failed: {e}")
            return []

    def search_with_namespace_fallback(self, queries: List[str], dataset: str, qid: str, top_k: Optional[int] = None,
                                       similarity_threshold: Optional[float] = None,
                                       include_metadata: bool = True) -> List[List[Dict[str, Any]]]:
        from utils.dataset_guard import filter_notes_by_namespace  # the reference's own helper (retriever.py:1009)
        out = []
        for query, hits in zip(queries, self.search(queries, top_k, similarity_threshold, include_metadata)):
            kept = filter_notes_by_namespace(hits, dataset, qid)
            if not kept and self.bm25_enabled:
                kept = filter_notes_by_namespace(self._bm25_search(query, top_k or self.top_k), dataset, qid)
            out.append(kept)
        return out
