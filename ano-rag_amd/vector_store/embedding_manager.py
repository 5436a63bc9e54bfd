"""MI355X drop-in for the reference's ``vector_store/embedding_manager.py`` (class ``EmbeddingManager``).

Process-wide singleton with the reference's attributes and methods (embedding_manager.py:57-779).  The object
in ``self.model`` is an ``anorag_hip.encoder.SentenceEncoder`` — the HIP forward pass — where the reference
holds a ``sentence_transformers.SentenceTransformer``; text assembly, truncation, query prefixing and the
never-raise conventions are the reference's.  Similarity helpers (``compute_similarity`` /
``find_most_similar``) run through the device index.

Model discovery looks in the same places as the reference (``<root>/models/embedding/...``,
embedding_manager.py:250-339) plus ``embedding.model_path`` / ``$ANORAG_MODEL_DIR``; there is no download
path (no network) and no CPU forward: if no local model directory is found, construction raises
``RuntimeError`` exactly like the reference does when every load attempt fails (:158-160, :248).
"""
from __future__ import annotations

import os
import re
import threading
import unicodedata
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from anorag_hip.compat import FileUtils, config, default_tmp, hip_available, logger

_QUERY_PREFIX = "Represent this sentence for searching relevant passages: "


class _Batcher:
    """placeholder for the reference's never-used BatchProcessor attribute (embedding_manager.py:94-97)"""

    def __init__(self, batch_size, use_gpu=True):
        self.batch_size, self.use_gpu = batch_size, use_gpu


class EmbeddingManager:
    """Embedding manager (local models only)."""

    _instance = None
    _model_loaded = False
    _lock = None

    def __new__(cls):
        if cls._instance is None:
            cls._instance = super(EmbeddingManager, cls).__new__(cls)
            cls._lock = threading.Lock()
        return cls._instance

    def __init__(self):
        if self._model_loaded:
            return
        with self._lock:
            if self._model_loaded:
                return
            self.model_name = config.get("embedding.model_name", "BAAI/bge-m3")
            self.batch_size = config.get("embedding.batch_size", 32)
            # the reference reports 'cuda' when torch sees a GPU (utils/gpu_utils.py:46-55); on ROCm that
            # string is what torch calls the HIP device too
            self.device = "cuda" if hip_available() else "cpu"
            self.hip_device = int(config.get("anorag_hip.device", 0) or 0)
            self.max_length = config.get("embedding.max_length", 512)
            self.normalize_embeddings = config.get("embedding.normalize", True)
            self.model = None
            self.embedding_dim = None
            self._load_local_model()
            self.batch_processor = _Batcher(self.batch_size, config.get("performance.use_gpu", True))
            self.cache_dir = config.get("storage.embedding_cache_path")
            if not self.cache_dir:
                work_dir = config.get("storage.work_dir")
                self.cache_dir = os.path.join(work_dir, "embeddings") if work_dir else default_tmp("anorag_embeddings")
            FileUtils.ensure_dir(self.cache_dir)
            self.consistency_checker = None  # utils.model_consistency is outside the hot path
            logger.info(f"EmbeddingManager initialized with model: {self.model_name}, device: {self.device}")
            EmbeddingManager._model_loaded = True

    @classmethod
    def _reset_singleton(cls):
        """tests only: drop the process-wide instance"""
        inst = cls._instance
        if inst is not None and getattr(inst, "model", None) is not None:
            try:
                inst.model.close()
            except Exception:
                pass
        cls._instance = None
        cls._model_loaded = False

    # -- model discovery ---------------------------------------------------------------------------
    def _get_local_model_paths(self, local_models_dir: str) -> List[str]:
        name = self.model_name
        paths = []
        for extra in (config.get("embedding.model_path"), os.environ.get("ANORAG_MODEL_DIR")):
            if extra:
                paths.append(extra)
        if os.path.isdir(name):
            paths.append(name)
        paths += [os.path.join(local_models_dir, name), os.path.join(local_models_dir, name.replace("/", "_")),
                  os.path.join(local_models_dir, f"models--{name.replace('/', '--')}")]
        for cache in (os.path.join(local_models_dir, name.replace("/", "_")),
                      os.path.join(local_models_dir, f"models--{name.replace('/', '--')}")):
            snaps = os.path.join(cache, "snapshots")
            if os.path.isdir(snaps):
                for d in sorted(os.listdir(snaps), reverse=True):
                    sp = os.path.join(snaps, d)
                    if os.path.isdir(sp) and (os.path.exists(os.path.join(sp, "modules.json"))
                                              or os.path.exists(os.path.join(sp, "config.json"))):
                        paths.insert(0, sp)
                        break
        if "sentence-transformers" in name:
            paths.append(os.path.join(local_models_dir, name.split("/")[-1]))
        return paths

    def _load_local_model(self):
        try:
            root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
            candidates = self._get_local_model_paths(os.path.join(root, "models/embedding"))
            for path in candidates:
                if self._try_load_model_from_path(path):
                    return
            if self._try_fallback_model(os.path.join(root, "models/embedding")):
                return
            raise FileNotFoundError(f"no local model directory among {candidates} (nor for the fallback models); "
                                    "this build cannot download")
        except Exception as e:
            logger.error(f"Failed to load the embedding model: {e}")
            raise RuntimeError(f"Unable to load the embedding model: {e}")

    _FALLBACK_MODELS = ("sentence-transformers/all-MiniLM-L6-v2", "sentence-transformers/all-mpnet-base-v2",
                        "sentence-transformers/paraphrase-multilingual-MiniLM-L12-v2")

    def _try_fallback_model(self, local_models_dir: str) -> bool:
        """The reference's three fallback models (embedding_manager.py:215-248), resolved LOCALLY: the same directories
        as the configured model plus the Hugging Face caches SentenceTransformer(name) would read offline
        ($SENTENCE_TRANSFORMERS_HOME, $HF_HOME/hub, $TRANSFORMERS_CACHE, ~/.cache/huggingface/hub).  No download.  As in
        the reference, model_name becomes the fallback's NAME (:238) and max_seq_length is left as the model has it."""
        configured = self.model_name
        caches = [os.environ.get("SENTENCE_TRANSFORMERS_HOME"), os.environ.get("TRANSFORMERS_CACHE"),
                  os.path.join(os.environ["HF_HOME"], "hub") if os.environ.get("HF_HOME") else None,
                  os.path.join(os.path.expanduser("~"), ".cache", "huggingface", "hub")]
        try:
            for name in self._FALLBACK_MODELS:
                self.model_name = name
                paths = self._get_local_model_paths(local_models_dir)
                for cache in caches:
                    if not cache:
                        continue
                    paths.append(os.path.join(cache, name.replace("/", "_")))
                    snaps = os.path.join(cache, f"models--{name.replace('/', '--')}", "snapshots")
                    if os.path.isdir(snaps):
                        paths += [os.path.join(snaps, d) for d in sorted(os.listdir(snaps), reverse=True)]
                for path in paths:
                    if path in (config.get("embedding.model_path"), os.environ.get("ANORAG_MODEL_DIR")):
                        continue  # already tried for the configured model
                    if self._try_load_model_from_path(path, keep_seq_length=True):
                        self.model_name = name
                        self.max_seq_length = getattr(self.model, "max_seq_length", 512)  # (:237)
                        logger.info(f"Fallback model loaded: {name} from {path}")
                        return True
        except Exception as e:
            logger.warning(f"Fallback model lookup failed: {e}")
        self.model_name = configured
        return False

    def _try_load_model_from_path(self, model_path: str, keep_seq_length: bool = False) -> bool:
        if not os.path.isdir(model_path):
            return False
        if not any(os.path.exists(os.path.join(model_path, f))
                   for f in ("modules.json", "config_sentence_transformers.json", "config.json")):
            return False
        try:
            from anorag_hip.encoder import SentenceEncoder
            self.model = SentenceEncoder(model_path, device=self.hip_device, trust_remote_code=True)
            self.embedding_dim = self.model.get_sentence_embedding_dimension()
            if not keep_seq_length:
                self.model.max_seq_length = self.max_length        # embedding_manager.py:361-362
            self.model_name = model_path                           # :365 — the name becomes the path
            logger.info(f"Model loaded from {model_path}, embedding dim {self.embedding_dim}")
            return True
        except Exception as e:
            logger.warning(f"Loading the model from {model_path} failed: {e}")
            return False

    # -- encoding ------------------------------------------------------------------------------------
    def encode_texts(self, texts: List[str], batch_size: Optional[int] = None, show_progress: bool = True,
                     normalize: Optional[bool] = None) -> np.ndarray:
        if not texts:
            return np.array([])
        batch_size = batch_size or self.batch_size
        normalize = normalize if normalize is not None else self.normalize_embeddings
        try:
            return self.model.encode(self._preprocess_texts(texts), batch_size=batch_size,
                                     show_progress_bar=show_progress, convert_to_numpy=True,
                                     normalize_embeddings=normalize, device=self.device)
        except Exception as e:
            logger.error(f"Text encoding failed: {e}")
            return np.zeros((len(texts), self.embedding_dim))  # float64 zeros, as in the reference (:407)

    def encode_texts_device(self, texts: List[str], batch_size: Optional[int] = None, normalize: Optional[bool] = None):
        """EXTENSION: ``encode_texts`` with the embeddings left in device memory (``anorag_hip.fusion.DeviceArray``
        [n, D] float32) for ``VectorRetriever`` to hand straight to the index (reference flow
        vector_store/retriever.py:140-157, :206-216).  Raises on failure instead of returning zeros — the caller falls
        back to the host-array path."""
        batch_size = batch_size or self.batch_size
        normalize = normalize if normalize is not None else self.normalize_embeddings
        return self.model.encode_device(self._preprocess_texts(texts), batch_size=batch_size,
                                        normalize_embeddings=normalize)

    def encode_atomic_notes(self, atomic_notes: List[Dict[str, Any]], content_field: str = "content",
                            include_metadata: bool = True) -> np.ndarray:
        if not atomic_notes:
            return np.array([])
        return self.encode_texts(self._assemble_note_texts(atomic_notes, content_field))

    def _assemble_note_texts(self, atomic_notes: List[Dict[str, Any]], content_field: str = "content") -> List[str]:
        strat = (config.get("embedding_strategy", {}) or {}).get("atomic_note_embedding", {})
        text_strategy = strat.get("text_strategy", "title_raw_span")
        priority = strat.get("field_priority", ["title", "raw_span", "original_text", "content"])
        combo = strat.get("text_combination", {})
        prep = strat.get("preprocessing", {})
        qc = strat.get("quality_control", {})
        texts = []
        for note in atomic_notes:
            try:
                if text_strategy == "content_only":
                    text = note.get(content_field, "")
                elif text_strategy == "title_content":
                    text = self._extract_title_content_text(note, content_field, combo)
                else:
                    text = self._extract_title_raw_span_text(note, priority, combo)
                text = self._preprocess_embedding_text(text, prep)
                if self._should_skip_note(text, qc):
                    text = "Empty note"
                texts.append(text)
            except Exception as e:
                logger.warning(f"Note text assembly failed: {e}")
                texts.append("Empty note" if qc.get("skip_invalid_encoding", True)
                             else note.get(content_field, "Empty note"))
        return texts

    def _extract_title_raw_span_text(self, note, field_priority, text_combination) -> str:
        title = note.get("title", "").strip()
        content = note.get("content", "").strip() or note.get("raw_span", "").strip()
        entities = note.get("entities", [])
        ent = ""
        if entities:
            ent = ", ".join(str(e) for e in entities if e) if isinstance(entities, list) else str(entities)
        text = f"{title} || {content} || ENTITIES: {ent}"
        limit = text_combination.get("max_combined_length", 512)
        if len(text) > limit:
            how = text_combination.get("truncate_strategy", "tail")
            if how == "head":
                text = text[:limit]
            elif how == "tail":
                text = text[-limit:]
            elif how == "middle":
                text = text[:limit // 2] + text[-(limit // 2):]
        return text or "Empty note"

    def _extract_title_content_text(self, note, content_field, text_combination) -> str:
        parts = [p for p in (note.get("title", "").strip(), note.get(content_field, "").strip()) if p]
        return text_combination.get("separator", " ").join(parts) or "Empty note"

    def _preprocess_embedding_text(self, text: str, preprocessing: Dict[str, Any]) -> str:
        if not text:
            return text
        if preprocessing.get("remove_extra_whitespace", True):
            text = re.sub(r"\s+", " ", text).strip()
        if preprocessing.get("normalize_unicode", True):
            text = unicodedata.normalize("NFKC", text)
        if preprocessing.get("remove_control_chars", True):
            text = re.sub(r"[\x00-\x1f\x7f-\x9f]", "", text)
        return text

    def _should_skip_note(self, text: str, quality_control: Dict[str, Any]) -> bool:
        if quality_control.get("skip_empty_notes", True) and not text.strip():
            return True
        return len(text.strip()) < quality_control.get("min_text_length", 3)

    def encode_queries(self, queries: List[str], query_prefix: str = _QUERY_PREFIX) -> np.ndarray:
        if not queries:
            return np.array([])
        if "bge" in self.model_name.lower() and query_prefix:  # tested against the *path* once loaded (:365, :558)
            queries = [query_prefix + q for q in queries]
        return self.encode_texts(queries)

    def encode_queries_device(self, queries: List[str], query_prefix: str = _QUERY_PREFIX):
        """EXTENSION: ``encode_queries`` with the embeddings left in device memory (see ``encode_texts_device``)"""
        if "bge" in self.model_name.lower() and query_prefix:
            queries = [query_prefix + q for q in queries]
        return self.encode_texts_device(queries)

    def _preprocess_texts(self, texts: List[str]) -> List[str]:
        out = []
        for t in texts:
            t = t.strip()
            if len(t) > self.max_length * 4:
                t = t[:self.max_length * 4]
            out.append(t or "Empty content")
        return out

    # -- similarity helpers (embedding_manager.py:586-660) through the device index ---------------------
    def compute_similarity(self, embeddings1: np.ndarray, embeddings2: np.ndarray, metric: str = "cosine") -> np.ndarray:
        """reference embedding_manager.py:586-629 (pinned by tests/golden/similarity_cases.json, a reference-run fixture): cosine
        x / (||x|| + 1e-8) with 1-D inputs promoted to one row; euclidean 1 / (1 + cdist) — float64 whatever the inputs, and
        cdist REFUSES 1-D inputs (the reference then returns its empty-array sentinel); dot = np.dot(a, b.T), whose result
        loses the axis of a 1-D operand; an unknown metric or mismatched widths -> the sentinel"""
        if embeddings1.size == 0 or embeddings2.size == 0:
            return np.array([])
        try:
            if metric not in ("cosine", "dot", "euclidean"):
                raise ValueError(f"unsupported similarity metric: {metric}")
            if metric == "euclidean" and (embeddings1.ndim != 2 or embeddings2.ndim != 2):
                raise ValueError("XA and XB must be 2-dimensional arrays")  # scipy.spatial.distance.cdist's own check
            a = embeddings1.reshape(1, -1) if embeddings1.ndim == 1 else embeddings1
            b = embeddings2.reshape(1, -1) if embeddings2.ndim == 1 else embeddings2
            import ctypes as C
            from anorag_hip import _lib
            # one tiled kernel (anr_similarity_matrix): the cosine normalisation x / (||x|| + 1e-8), the products
            # (float64 accumulation) and the euclidean 1 / (1 + distance) all run on the device
            a32 = np.ascontiguousarray(a, dtype=np.float32)
            b32 = np.ascontiguousarray(b, dtype=np.float32)
            if a32.shape[1] != b32.shape[1]:
                raise ValueError(f"shapes {a.shape} and {b.shape} not aligned")
            sim = np.empty((a32.shape[0], b32.shape[0]), dtype=np.float64)
            code = {"cosine": 0, "dot": 1, "euclidean": 2}[metric]
            _lib.check(_lib.load().anr_similarity_matrix(self.hip_device, a32.ctypes.data, a32.shape[0],
                                                         b32.ctypes.data, b32.shape[0], a32.shape[1], code,
                                                         sim.ctypes.data), "anr_similarity_matrix")
            if metric == "euclidean":
                return sim  # scipy's cdist returns float64 whatever the inputs are (:613-616)
            sim = sim.astype(np.result_type(embeddings1.dtype, embeddings2.dtype), copy=False)
            if metric == "dot":  # np.dot(a, b.T): a 1-D operand's axis is not in the result
                if embeddings1.ndim == 1 and embeddings2.ndim == 1:
                    return sim.reshape(())[()]
                if embeddings1.ndim == 1 or embeddings2.ndim == 1:
                    return sim.reshape(-1)
            return sim
        except Exception as e:
            logger.error(f"Similarity computation failed: {e}")
            return np.array([])

    def find_most_similar(self, query_embedding: np.ndarray, candidate_embeddings: np.ndarray, top_k: int = 10,
                          metric: str = "cosine") -> List[Dict[str, Any]]:
        """reference embedding_manager.py:631-660: the similarities of the one query (the device kernel above), then the
        reference's OWN ordering call — np.argsort(similarities)[::-1][:top_k] — so that equal similarities come out in the
        order numpy gives the reference (the exact-scan index orders ties by ascending id, numpy's reversed ascending sort
        the other way round)"""
        if query_embedding.size == 0 or candidate_embeddings.size == 0:
            return []
        similarities = self.compute_similarity(query_embedding.reshape(1, -1), candidate_embeddings, metric=metric)
        if similarities.size == 0:
            return []
        similarities = similarities.flatten()
        top = np.argsort(similarities)[::-1][:top_k]
        return [{"index": int(i), "similarity": float(similarities[i])} for i in top]

    # -- metadata --------------------------------------------------------------------------------------
    def get_model_info(self) -> Dict[str, Any]:
        return {"model_name": self.model_name, "embedding_dim": self.embedding_dim, "device": str(self.device),
                "max_length": self.max_length, "batch_size": self.batch_size,
                "normalize_embeddings": self.normalize_embeddings}

    def cleanup(self):
        logger.info("EmbeddingManager cleanup completed")

    def register_model_signature(self) -> None:
        return None

    def get_model_signature(self) -> Optional[Dict[str, Any]]:
        if not self.model:
            return None
        return {"model_name": self.model_name, "model_type": "sentence_transformer", "dimension": self.embedding_dim,
                "max_length": self.max_length, "normalize": self.normalize_embeddings,
                "metadata": {"device": str(self.device), "batch_size": self.batch_size}}

    def validate_model_consistency(self, other_signature: Optional[Dict[str, Any]] = None) -> Tuple[bool, Optional[str]]:
        if not self.model:
            return True, None
        try:
            cur = self.get_model_signature()
            if not cur:
                return False, "Failed to create model signature"
            if other_signature:
                for attr in ("model_name", "model_type", "dimension", "normalize"):
                    if cur.get(attr) != other_signature.get(attr):
                        return False, f"Inconsistent {attr}: {cur.get(attr)} vs {other_signature.get(attr)}"
                return True, "Model signatures are consistent"
            return True, "Model consistency validated"
        except Exception as e:
            return False, str(e)
