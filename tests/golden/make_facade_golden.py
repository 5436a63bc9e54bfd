#!/usr/bin/env python3
"""Golden vectors for the Python-level logic of the three drop-in classes, produced by RUNNING THE REFERENCE's own
files (build container only; the GPU box never runs this script):

  python tests/golden/make_facade_golden.py [/root/reference]

  vector_store/vector_index.py      VectorIndex._preprocess_vectors (:265-282) and the result shaping of
                                    VectorIndex.search (:226-259: dict keys, -1 dropped, flat list for one query,
                                    similarity = score | 1 / (1 + score), 1-D input -> [])
  vector_store/retriever.py         VectorRetriever.search (:186-272: `thr or default`, threshold filter, note copy,
                                    retrieval_info, include_metadata=False slim dict) and .retrieve (:339-512:
                                    over-fetch, filter_fn, x0.6 / x1.2 / x1.15 multipliers, adjustments, re-threshold,
                                    sort, cut), plus the constructor's defaults
  vector_store/embedding_manager.py encode_atomic_notes text assembly (:409-549), _preprocess_texts (:566-584),
                                    encode_queries prefix rule (:551-564), encode_texts sentinels (:374-407)

The reference files import third-party packages that are not installed here (faiss, sentence-transformers, loguru) and
its own `utils` / `config` packages (which pull in many more).  None of the logic pinned here lives in them, so they are
satisfied by STAND-INS in sys.modules, clearly NOT the real libraries:
  * `faiss`: IndexFlatIP / IndexFlatL2 as a numpy exact search (float32 matmul, stable best-first order, -1 padding).
    It pins nothing about faiss itself (SURVEY.md 8a row b7 stays "parity unpinned"): every case stores the
    (scores, indices) the stand-in returned, and the fixture pins only what the reference's Python does with them.
  * `sentence_transformers.SentenceTransformer`: records the texts it is asked to encode and returns the vectors the
    case assigns to them.  It pins nothing about the encoder (row a5 stays unpinned).
  * `loguru`, `utils` (GPUUtils / FileUtils / BatchProcessor), `config` (get(key, default) -> default or the case's
    override): no arithmetic.
Only data is written (inputs + the reference's outputs); no reference source is copied.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
CONFIG_OVERRIDES = {}


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install_stand_ins():
    lg = types.ModuleType("loguru")

    class _L:
        def __getattr__(self, _):
            return lambda *a, **k: None

    lg.logger = _L()
    sys.modules["loguru"] = lg

    # ---- faiss stand-in: numpy exact flat index -------------------------------------------------------------------
    fa = types.ModuleType("faiss")
    fa.METRIC_INNER_PRODUCT, fa.METRIC_L2 = 0, 1
    fa.get_num_gpus = lambda: 0
    fa.last_search = None

    class _Flat:
        def __init__(self, d, ip):
            self.d, self.ip, self.x, self.is_trained, self.ntotal = d, ip, np.zeros((0, d), np.float32), True, 0

        def add(self, v):
            self.x = np.vstack([self.x, np.asarray(v, np.float32)])
            self.ntotal = self.x.shape[0]

        def search(self, q, k):
            q = np.asarray(q, np.float32)
            if self.ip:
                s = q @ self.x.T
                order = np.argsort(-s, axis=1, kind="stable")
            else:
                s = ((q[:, None, :] - self.x[None, :, :]) ** 2).sum(-1).astype(np.float32)
                order = np.argsort(s, axis=1, kind="stable")
            kk = min(k, self.ntotal)
            I = np.full((q.shape[0], k), -1, np.int64)
            D = np.full((q.shape[0], k), -3.4028235e38 if self.ip else 3.4028235e38, np.float32)
            I[:, :kk] = order[:, :kk]
            D[:, :kk] = np.take_along_axis(s, order[:, :kk], axis=1)
            fa.last_search = (D.copy(), I.copy())
            return D, I

        def reset(self):
            self.x = np.zeros((0, self.d), np.float32)
            self.ntotal = 0

    fa.IndexFlatIP = lambda d: _Flat(d, True)
    fa.IndexFlatL2 = lambda d: _Flat(d, False)
    sys.modules["faiss"] = fa

    # ---- sentence_transformers stand-in -----------------------------------------------------------------------------
    stm = types.ModuleType("sentence_transformers")

    class SentenceTransformer:
        def __init__(self, *a, **k):
            self.calls = []          # every list of texts passed to encode, with the keyword arguments
            self.table = {}          # text -> vector
            self.seen = {}           # text -> vector the stand-in made up for a text outside the table
            self.dim = 8
            self.fail = False
            self.max_seq_length = 512

        def get_sentence_embedding_dimension(self):
            return self.dim

        def encode(self, texts, **kw):
            self.calls.append({"texts": list(texts), "kwargs": {k: (v if isinstance(v, (int, float, bool, str, type(None))) else str(v))
                                                               for k, v in kw.items()}})
            if self.fail:
                raise RuntimeError("stand-in encoder asked to fail")
            out = np.zeros((len(texts), self.dim), np.float32)
            for i, t in enumerate(texts):
                if t in self.table:
                    out[i] = self.table[t]
                else:
                    import hashlib
                    seed = int.from_bytes(hashlib.md5(t.encode("utf-8")).digest()[:4], "little")
                    v = np.random.default_rng(seed).standard_normal(self.dim).astype(np.float32)
                    out[i] = v / np.linalg.norm(v)
                    self.seen[t] = out[i].copy()
            return out

    stm.SentenceTransformer = SentenceTransformer
    sys.modules["sentence_transformers"] = stm

    # ---- utils / config stand-ins -------------------------------------------------------------------------------------
    u = types.ModuleType("utils")

    class GPUUtils:
        @staticmethod
        def is_cuda_available():
            return False

        @staticmethod
        def get_optimal_device():
            return "cpu"

    class FileUtils:
        @staticmethod
        def ensure_dir(p):
            os.makedirs(p, exist_ok=True)

    class BatchProcessor:
        def __init__(self, *a, **k):
            pass

    u.GPUUtils, u.FileUtils, u.BatchProcessor = GPUUtils, FileUtils, BatchProcessor
    sys.modules["utils"] = u
    c = types.ModuleType("config")

    class _Cfg:
        def get(self, key, default=None):
            return CONFIG_OVERRIDES.get(key, default)

    c.config = _Cfg()
    sys.modules["config"] = c
    return fa, stm


def jf(a):
    return [[float(v) for v in row] for row in np.asarray(a)]


def vector_index_cases(vi_mod, fa):
    rng = np.random.default_rng(20261101)
    cases = {"preprocess": [], "search": []}

    def new_index(metric, dim, itype="Flat"):
        CONFIG_OVERRIDES.clear()
        CONFIG_OVERRIDES.update({"vector_store.index_type": itype, "vector_store.similarity_metric": metric,
                                 "vector_store.dimension": dim})
        v = vi_mod.VectorIndex(dim)
        assert v.create_index()
        return v

    # _preprocess_vectors: dtype, layout, zero rows, metric
    for name, arr, metric in [
        ("f64_rows", rng.standard_normal((5, 6)), "cosine"),
        ("f32_zero_row", np.vstack([rng.standard_normal((2, 4)).astype(np.float32), np.zeros((1, 4), np.float32)]), "cosine"),
        ("non_contiguous", np.asfortranarray(rng.standard_normal((4, 5)).astype(np.float32)), "cosine"),
        ("strided_view", rng.standard_normal((6, 8)).astype(np.float32)[::2, ::2], "cosine"),
        ("l2_untouched", rng.standard_normal((3, 4)) * 5.0, "l2"),
        ("dot_product_means_l2", rng.standard_normal((3, 4)) * 5.0, "dot_product"),
        ("int_input", rng.integers(-3, 4, (4, 4)), "cosine"),
        ("tiny_norms", rng.standard_normal((3, 4)).astype(np.float32) * 1e-20, "cosine"),
    ]:
        v = new_index(metric, arr.shape[1])
        out = v._preprocess_vectors(arr)
        cases["preprocess"].append({"name": name, "metric": metric, "input": jf(arr), "input_dtype": str(arr.dtype),
                                    "input_c_contiguous": bool(arr.flags["C_CONTIGUOUS"]),
                                    "expected": jf(out), "expected_dtype": str(out.dtype),
                                    "expected_c_contiguous": bool(out.flags["C_CONTIGUOUS"])})

    # search shaping (the stand-in index supplies scores / indices; they are stored with the case)
    def run(name, metric, n, d, nq, k, note=None, q_override=None, itype="Flat", add_zero=False):
        v = new_index(metric, d, itype)
        x = rng.standard_normal((n, d)).astype(np.float32) * (1.0 if metric == "cosine" else 0.7)
        if add_zero and n:
            x[0] = 0.0
        if n:
            assert v.add_vectors(x)
        q = q_override if q_override is not None else rng.standard_normal((nq, d)).astype(np.float32)
        fa.last_search = None
        out = v.search(q, top_k=k)
        raw = fa.last_search
        cases["search"].append({"name": name, "metric": metric, "index_type": itype, "vectors": jf(x) if n else [],
                                "queries": (jf(q) if np.asarray(q).ndim == 2 else [float(t) for t in q]),
                                "queries_ndim": int(np.asarray(q).ndim), "top_k": k,
                                "raw_scores": jf(raw[0]) if raw else None, "raw_indices": [[int(t) for t in r] for r in raw[1]] if raw else None,
                                "total_vectors": int(v.total_vectors), "expected": out, "note": note})

    run("cosine_batch", "cosine", 40, 8, 5, 6)
    run("cosine_single_query_flat_list", "cosine", 40, 8, 1, 6, note="one query -> flat list (:255-257)")
    run("cosine_k_exceeds_n", "cosine", 4, 8, 3, 7, note="-1 padded entries are dropped (:233-235)")
    run("cosine_k_exceeds_n_single", "cosine", 3, 8, 1, 5)
    run("l2_batch", "l2", 30, 6, 4, 5, note="similarity = 1 / (1 + score) (:247-249)")
    run("dot_product_is_l2", "dot_product", 30, 6, 2, 4, note="any metric other than cosine means L2 (:69-74)")
    run("empty_index", "cosine", 0, 8, 2, 5, note="total_vectors == 0 -> []")
    run("one_dim_query", "cosine", 10, 8, 1, 3, q_override=rng.standard_normal(8).astype(np.float32),
        note="1-D input: norm(axis=1) raises -> [] (:261-263)")
    run("zero_row_in_corpus", "cosine", 12, 6, 2, 12, add_zero=True)
    run("float64_queries", "cosine", 20, 8, 2, 4, q_override=rng.standard_normal((2, 8)))
    return cases


class FakeNotes:
    @staticmethod
    def make(rng, n):
        vocab = ["alpha", "beta", "gamma", "delta", "paris", "london", "founded", "born", "located", "river", "city", "year"]
        notes = []
        for i in range(n):
            words = [str(w) for w in rng.choice(vocab, size=int(rng.integers(3, 9)))]
            note = {"note_id": f"note_{i:04d}", "title": f"Title {i}", "content": " ".join(words),
                    "paragraph_idxs": [int(i % 7), int((i * 3) % 5)], "entities": [str(w) for w in rng.choice(vocab, 2)]}
            if i % 5 == 0:
                note["content"] = {"text": " ".join(words)}   # dict content (retriever.py:415-417)
            if i % 7 == 3:
                note["content"] = 12345                          # non-string content (:418-419)
            notes.append(note)
        return notes


def retriever_cases(rt_mod, em_mod, stm):
    rng = np.random.default_rng(20261102)
    cases = []
    d = 8
    # the singleton the reference's VectorRetriever() picks up: built without __init__ (the real one would try to
    # download a model and save it under the reference tree); its methods are the reference's own
    em = object.__new__(em_mod.EmbeddingManager)
    em.model = stm.SentenceTransformer()
    em.model.dim = d
    em.model_name, em.batch_size, em.device, em.max_length = "BAAI/bge-m3", 32, "cpu", 512
    em.normalize_embeddings, em.embedding_dim, em.consistency_checker = True, d, None
    em_mod.EmbeddingManager._instance = em
    em_mod.EmbeddingManager._model_loaded = True

    def unit(v):
        v = np.asarray(v, np.float32)
        return v / np.linalg.norm(v, axis=-1, keepdims=True)

    def fresh(n, metric="cosine"):
        CONFIG_OVERRIDES.clear()
        CONFIG_OVERRIDES.update({"vector_store.index_type": "Flat", "vector_store.similarity_metric": metric})
        r = rt_mod.VectorRetriever()
        notes = FakeNotes.make(rng, n)
        emb = unit(rng.standard_normal((n, d)))
        r.atomic_notes = notes
        r.note_embeddings = emb
        r._build_id_mappings()
        assert r.vector_index.create_index()
        assert r.vector_index.add_vectors(emb, np.arange(n, dtype=np.int64))
        return r, notes, emb

    defaults = None

    def query_vecs(emb, queries, near):
        """query i close to note near[i] (similarity ~0.9 with a spread of lower ones)"""
        prefix = "Represent this sentence for searching relevant passages: "
        table = {}
        for qtext, j in zip(queries, near):
            v = emb[j] + 0.35 * rng.standard_normal(d).astype(np.float32)
            table[prefix + qtext] = unit(v)
        return table

    def add_search(name, n, queries, near, **kw):
        nonlocal defaults
        r, notes, emb = fresh(n)
        if defaults is None:
            defaults = {k: getattr(r, k) for k in ("top_k", "similarity_threshold", "batch_size", "default_topk_multiplier",
                                                   "must_have_terms_penalty", "entity_boost_factor", "predicate_boost_factor",
                                                   "default_must_have_terms", "default_boost_entities", "default_boost_predicates",
                                                   "bm25_enabled", "enable_hybrid_search")}
        em.model.table = query_vecs(emb, queries, near)
        em.model.calls = []
        out = r.search(queries, **kw)
        cases.append({"kind": "search", "name": name, "notes": notes, "note_embeddings": jf(emb), "queries": queries,
                      "encoder_texts": em.model.calls[-1]["texts"] if em.model.calls else None,
                      "query_vectors": {k: [float(t) for t in v] for k, v in em.model.table.items()},
                      "kwargs": kw, "expected": out})

    def add_retrieve(name, n, query, near, filter_spec=None, **kw):
        r, notes, emb = fresh(n)
        em.model.table = query_vecs(emb, [query], [near])
        em.model.calls = []
        call_kw = dict(kw)
        if filter_spec is not None:
            mod = filter_spec["paragraph_mod"]
            call_kw["filter_fn"] = (lambda c: 1 // (c["paragraph_idxs"][0] - mod["raise_on"]) and c["paragraph_idxs"][0] != mod["reject"])
        out = r.retrieve(query, **call_kw)
        cases.append({"kind": "retrieve", "name": name, "notes": notes, "note_embeddings": jf(emb), "query": query,
                      "query_vectors": {k: [float(t) for t in v] for k, v in em.model.table.items()},
                      "kwargs": kw, "filter_spec": filter_spec, "expected": out})

    add_search("defaults_two_queries", 60, ["who founded paris", "river city"], [3, 17])
    add_search("explicit_top_k_and_threshold", 60, ["alpha beta"], [5], top_k=8, similarity_threshold=0.2)
    add_search("zero_threshold_falls_back_to_default", 60, ["gamma"], [9], top_k=10, similarity_threshold=0.0)
    add_search("slim_dicts", 40, ["delta london", "born year"], [1, 2], top_k=5, similarity_threshold=0.1, include_metadata=False)
    add_search("negative_threshold_keeps_all", 30, ["x y"], [4], top_k=30, similarity_threshold=-1.0)
    add_search("top_k_beyond_corpus", 6, ["short corpus"], [2], top_k=10, similarity_threshold=-1.0)
    add_search("empty_queries", 10, [], [])
    add_retrieve("plain", 80, "who founded paris", 11)
    add_retrieve("must_have_terms", 80, "paris river", 12, must_have_terms=["paris", "RIVER"], top_k=10, similarity_threshold=0.1)
    add_retrieve("boost_entities_and_predicates", 80, "london born", 13, boost_entities=["London", "alpha"],
                 boost_predicates=["born", "founded"], top_k=12, similarity_threshold=0.05)
    add_retrieve("all_adjustments", 80, "city year", 14, must_have_terms=["gamma"], boost_entities=["city"],
                 boost_predicates=["located"], top_k=15, similarity_threshold=0.05, topk_multiplier=2.0)
    add_retrieve("zero_threshold_quirk", 80, "beta", 15, top_k=6, similarity_threshold=0.0, must_have_terms=["zzz"])
    add_retrieve("multiplier_none_uses_default", 80, "delta", 16, top_k=4, topk_multiplier=None, similarity_threshold=0.05)
    add_retrieve("filter_fn_with_exception", 80, "alpha city", 18, top_k=10, similarity_threshold=0.05,
                 filter_spec={"paragraph_mod": {"reject": 2, "raise_on": 4}})
    add_retrieve("slim", 50, "gamma river", 19, top_k=5, similarity_threshold=0.05, include_metadata=False,
                 must_have_terms=["river"])
    add_retrieve("empty_query", 20, "", 0)
    return {"defaults": defaults, "cases": cases}


def lifecycle_cases(rt_mod, em_mod, stm):
    """build_index / add_notes / remove_notes / update_note / get_similar_notes and the TF-IDF "BM25" fallback with the
    namespace filter (retriever.py:118-184, 514-659, 924-1034; sklearn and the reference's utils/dataset_guard.py are the
    real ones), every step's observable state recorded"""
    dg = load_by_path("utils.dataset_guard", os.path.join(REF, "utils", "dataset_guard.py"))
    sys.modules["utils"].dataset_guard = dg
    rng = np.random.default_rng(20261103)
    d = 8
    em = object.__new__(em_mod.EmbeddingManager)
    em.model = stm.SentenceTransformer()
    em.model.dim = d
    em.model_name, em.batch_size, em.device, em.max_length = "BAAI/bge-m3", 32, "cpu", 512
    em.normalize_embeddings, em.embedding_dim, em.consistency_checker = True, d, None
    em_mod.EmbeddingManager._instance = em
    em_mod.EmbeddingManager._model_loaded = True
    CONFIG_OVERRIDES.clear()
    CONFIG_OVERRIDES.update({"vector_store.index_type": "Flat", "vector_store.similarity_metric": "cosine"})
    import tempfile
    CONFIG_OVERRIDES["storage.vector_store_path"] = tempfile.mkdtemp(prefix="anr_golden_vs_")
    CONFIG_OVERRIDES["storage.vector_index_path"] = tempfile.mkdtemp(prefix="anr_golden_vi_")
    r = rt_mod.VectorRetriever()
    notes = FakeNotes.make(rng, 30)
    for i, n in enumerate(notes):
        if not isinstance(n["content"], str):
            n["content"] = f"plain content {i} paris river"      # the lifecycle methods call .get('content') as text
        n["source_info"] = {"file_path": f"/data/{'musique' if i % 3 else 'hotpot'}/q{i % 4}/doc_{i}.json"}
    steps = []

    def state():
        return {"n_notes": len(r.atomic_notes), "note_ids": [n.get("note_id") for n in r.atomic_notes],
                "total_vectors": int(r.vector_index.total_vectors),
                "embeddings_shape": list(r.note_embeddings.shape) if r.note_embeddings is not None else None,
                "note_id_to_index": dict(r.note_id_to_index), "index_to_note_id": {str(k): v for k, v in r.index_to_note_id.items()},
                "tfidf_rows": int(r.tfidf_matrix.shape[0]) if r.tfidf_matrix is not None else None}

    em.model.calls = []
    ok = r.build_index([dict(n) for n in notes], force_rebuild=True, save_index=False)
    steps.append({"op": "build_index", "returned": ok, "encoder_texts": em.model.calls[-1]["texts"], "state": state()})
    steps.append({"op": "build_index_empty", "returned": r.build_index([], force_rebuild=True, save_index=False)})
    q = ["paris river founded", "gamma delta city", "zzzz qqqq"]
    steps.append({"op": "_bm25_search", "queries": q, "top_k": 5, "returned": [r._bm25_search(x, top_k=5) for x in q]})
    em.model.calls = []
    res = r.search_with_namespace_fallback(q, "musique", "q1", top_k=6, similarity_threshold=-1.0)
    steps.append({"op": "search_with_namespace_fallback", "queries": q, "dataset": "musique", "qid": "q1", "top_k": 6,
                  "similarity_threshold": -1.0, "returned": res})
    res = r.search_with_namespace_fallback(q[:2], "nosuch", "q9", top_k=4, similarity_threshold=-1.0)
    steps.append({"op": "search_with_namespace_fallback", "queries": q[:2], "dataset": "nosuch", "qid": "q9", "top_k": 4,
                  "similarity_threshold": -1.0, "returned": res})
    nid = notes[3]["note_id"]
    steps.append({"op": "get_similar_notes", "note_id": nid, "top_k": 4, "returned": r.get_similar_notes(nid, top_k=4)})
    steps.append({"op": "get_similar_notes", "note_id": "missing", "top_k": 4, "returned": r.get_similar_notes("missing", top_k=4)})
    steps.append({"op": "get_notes_by_ids", "ids": [notes[1]["note_id"], "missing", notes[7]["note_id"]],
                  "returned": r.get_notes_by_ids([notes[1]["note_id"], "missing", notes[7]["note_id"]])})
    new = [{"note_id": "new_a", "title": "New A", "content": "alpha beta new content", "entities": ["alpha"], "paragraph_idxs": [1]},
           {"note_id": "new_b", "title": "New B", "content": "river city founded year", "entities": [], "paragraph_idxs": [2]}]
    em.model.calls = []
    ok = r.add_notes([dict(n) for n in new])
    steps.append({"op": "add_notes", "notes": new, "returned": ok, "encoder_texts": em.model.calls[-1]["texts"], "state": state()})
    steps.append({"op": "add_notes_empty", "returned": r.add_notes([])})
    steps.append({"op": "search", "queries": ["river city"], "kwargs": {"top_k": 5, "similarity_threshold": -1.0},
                  "returned": r.search(["river city"], top_k=5, similarity_threshold=-1.0)})
    em.model.calls = []
    ok = r.remove_notes([notes[2]["note_id"], "not_there", "new_a"])
    steps.append({"op": "remove_notes", "ids": [notes[2]["note_id"], "not_there", "new_a"], "returned": ok,
                  "encoder_texts": em.model.calls[-1]["texts"] if em.model.calls else None, "state": state()})
    steps.append({"op": "remove_notes_unknown", "returned": r.remove_notes(["nobody"]), "state": state()})
    upd = {"note_id": notes[5]["note_id"], "title": "Updated", "content": "completely new text about london", "entities": ["london"],
           "paragraph_idxs": [9]}
    em.model.calls = []
    ok = r.update_note(notes[5]["note_id"], dict(upd))
    steps.append({"op": "update_note", "note_id": notes[5]["note_id"], "note": upd, "returned": ok,
                  "encoder_texts": [c["texts"] for c in em.model.calls], "state": state()})
    steps.append({"op": "update_note_unknown", "returned": r.update_note("nobody", dict(upd))})
    steps.append({"op": "search", "queries": ["london text"], "kwargs": {"top_k": 3, "similarity_threshold": -1.0},
                  "returned": r.search(["london text"], top_k=3, similarity_threshold=-1.0)})
    vectors = {k: [float(t) for t in v] for k, v in em.model.seen.items()}
    return {"notes": notes, "dim": d, "vectors": vectors, "steps": steps}


def embedding_manager_cases(em_mod, stm):
    CONFIG_OVERRIDES.clear()
    em = object.__new__(em_mod.EmbeddingManager)
    em.model = stm.SentenceTransformer()
    em.model.dim = 4
    em.model_name, em.batch_size, em.device, em.max_length = "BAAI/bge-m3", 32, "cpu", 512
    em.normalize_embeddings, em.embedding_dim, em.consistency_checker = True, 4, None
    out = {}
    long_text = "word " * 700
    notes = [
        {"title": "Plain", "content": "some content here", "entities": ["A", "B"]},
        {"title": "  padded title  ", "content": "  padded content ", "raw_span": "ignored span", "entities": ["x", "", None, "y"]},
        {"title": "Raw span only", "content": "", "raw_span": "the raw span text", "entities": []},
        {"title": "String entities", "content": "c", "entities": "just a string"},
        {"title": "No entities key", "content": "c2"},
        {"title": "", "content": "", "raw_span": ""},
        {"title": "Long", "content": long_text, "entities": ["tail", "kept"]},
        {"title": "White\tspace\n\nrun", "content": "a  b c   d", "entities": ["ＡＢ"]},
        {"title": "Ctl", "content": "x\x00y\x1fz\x7fw\x85v", "entities": []},
        {"content": "no title"},
        {"title": "ﬁ ligature ① ½", "content": "ｆｕｌｌ width", "entities": ["é", "é"]},
        {"title": "T", "content": 0, "entities": []},                                      # falsy non-string -> raw_span path raises
    ]
    em.model.calls = []
    em.encode_atomic_notes(notes)
    out["encode_atomic_notes"] = {"notes": notes, "texts_given_to_encoder": em.model.calls[-1]["texts"],
                                  "encode_kwargs": em.model.calls[-1]["kwargs"]}
    texts = ["  strip me  ", "", "   ", "x" * 2047, "y" * 2048, "z" * 2049, "z" * 5000 + "  ", "\n\ttabbed\n"]
    out["preprocess_texts"] = {"max_length": em.max_length, "input": texts, "expected": em._preprocess_texts(texts)}
    em.max_length = 16
    out["preprocess_texts_short_limit"] = {"max_length": 16, "input": texts, "expected": em._preprocess_texts(texts)}
    em.max_length = 512
    q = ["what is x", "  spaced  ", ""]
    res = {}
    for model_name in ("BAAI/bge-m3", "/models/embedding/BAAI_bge-base-en", "sentence-transformers/all-MiniLM-L6-v2",
                       "/data/BGE-large"):
        em.model_name = model_name
        em.model.calls = []
        em.encode_queries(q)
        res[model_name] = em.model.calls[-1]["texts"]
        em.model.calls = []
        em.encode_queries(q, query_prefix="")
        res[model_name + "|empty_prefix"] = em.model.calls[-1]["texts"]
    out["encode_queries"] = {"queries": q, "texts_given_to_encoder": res}
    em.model_name = "BAAI/bge-m3"
    r = em.encode_texts([])
    out["encode_texts_empty"] = {"shape": list(r.shape), "dtype": str(r.dtype)}
    em.model.fail = True
    r = em.encode_texts(["a", "b", "c"])
    out["encode_texts_failure"] = {"shape": list(r.shape), "dtype": str(r.dtype), "all_zero": bool((r == 0).all())}
    em.model.fail = False
    em.model.calls = []
    em.encode_texts(["t"], batch_size=7, show_progress=False, normalize=False)
    out["encode_texts_kwargs"] = em.model.calls[-1]["kwargs"]
    r = em.encode_atomic_notes([])
    out["encode_atomic_notes_empty"] = {"shape": list(r.shape), "dtype": str(r.dtype)}
    return out


def main():
    fa, stm = install_stand_ins()
    pkg = types.ModuleType("vector_store")
    pkg.__path__ = [os.path.join(REF, "vector_store")]
    sys.modules["vector_store"] = pkg
    em_mod = load_by_path("vector_store.embedding_manager", os.path.join(REF, "vector_store", "embedding_manager.py"))
    vi_mod = load_by_path("vector_store.vector_index", os.path.join(REF, "vector_store", "vector_index.py"))
    rt_mod = load_by_path("vector_store.retriever", os.path.join(REF, "vector_store", "retriever.py"))
    note = ("faiss / sentence-transformers / loguru / utils / config are stand-ins (see make_facade_golden.py): these fixtures "
            "pin the reference's own Python around them, not those libraries")
    with open(os.path.join(HERE, "vector_index_facade_cases.json"), "w") as f:
        json.dump({"source": "reference vector_store/vector_index.py VectorIndex._preprocess_vectors / .search", "note": note,
                   **vector_index_cases(vi_mod, fa)}, f)
    with open(os.path.join(HERE, "retriever_facade_cases.json"), "w") as f:
        json.dump({"source": "reference vector_store/retriever.py VectorRetriever.search / .retrieve", "note": note,
                   **retriever_cases(rt_mod, em_mod, stm)}, f)
    with open(os.path.join(HERE, "retriever_lifecycle_cases.json"), "w") as f:
        json.dump({"source": "reference vector_store/retriever.py build_index / add_notes / remove_notes / update_note / "
                             "get_similar_notes / _bm25_search / search_with_namespace_fallback (sklearn and the reference's "
                             "utils/dataset_guard.py are the real ones)", "note": note, **lifecycle_cases(rt_mod, em_mod, stm)}, f)
    with open(os.path.join(HERE, "embedding_manager_facade_cases.json"), "w") as f:
        json.dump({"source": "reference vector_store/embedding_manager.py text assembly / preprocessing / prefix / sentinels",
                   "note": note, **embedding_manager_cases(em_mod, stm)}, f)
    print("wrote vector_index_facade_cases.json, retriever_facade_cases.json, retriever_lifecycle_cases.json, embedding_manager_facade_cases.json")


if __name__ == "__main__":
    main()
