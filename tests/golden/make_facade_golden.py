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

    class _IVF(_Flat):
        """IndexIVFFlat / IndexIVFPQ stand-in: an EXACT flat search that carries nlist / nprobe and wants training — it pins
        the reference's Python around an IVF index (training, add_with_ids, the nprobe sweep), nothing about IVF recall"""
        def __init__(self, quantizer, d, nlist, *rest):
            metric = rest[-1]
            super().__init__(d, metric == fa.METRIC_INNER_PRODUCT)
            self.nlist, self.nprobe, self.is_trained, self.ids = nlist, 1, False, None
            self.nprobe_seen = []

        def train(self, v):
            self.is_trained = True

        def add_with_ids(self, v, ids):
            self.add(v)
            self.ids = np.asarray(ids, np.int64) if self.ids is None else np.concatenate([self.ids, np.asarray(ids, np.int64)])

        def search(self, q, k):
            self.nprobe_seen.append(int(self.nprobe))
            D, I = super().search(q, k)
            if self.ids is not None:
                I = np.where(I >= 0, self.ids[np.clip(I, 0, None)], -1)
                fa.last_search = (D.copy(), I.copy())
            return D, I

    fa.IndexIVFFlat = _IVF
    fa.IndexIVFPQ = _IVF

    def write_index(index, path):     # what faiss.write_index / read_index mean to the reference: the object comes back
        import pickle
        with open(path, "wb") as f:
            pickle.dump({"cls": type(index).__name__, "d": index.d, "ip": index.ip, "x": index.x,
                         "nlist": getattr(index, "nlist", None), "nprobe": getattr(index, "nprobe", None),
                         "ids": getattr(index, "ids", None)}, f)

    def read_index(path):
        import pickle
        with open(path, "rb") as f:
            st = pickle.load(f)
        if st["cls"] == "_IVF":
            ix = _IVF(None, st["d"], st["nlist"], fa.METRIC_INNER_PRODUCT if st["ip"] else fa.METRIC_L2)
            ix.nprobe, ix.ids, ix.is_trained = st["nprobe"], st["ids"], True
        else:
            ix = _Flat(st["d"], st["ip"])
        ix.x, ix.ntotal = st["x"], st["x"].shape[0]
        return ix

    fa.write_index, fa.read_index = write_index, read_index
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

        # (the reference's utils/file_utils.py cannot be imported here — it pulls in jsonlines and python-docx; what its
        # write_json / read_json do for these callers: numpy scalars become Python numbers, then plain json)
        @staticmethod
        def write_json(data, path):
            def conv(o):
                if isinstance(o, np.integer):
                    return int(o)
                if isinstance(o, np.floating):
                    return float(o)
                if isinstance(o, np.ndarray):
                    return o.tolist()
                if isinstance(o, dict):
                    return {(int(k) if isinstance(k, np.integer) else k): conv(v) for k, v in o.items()}
                if isinstance(o, (list, tuple)):
                    return [conv(v) for v in o]
                return o
            with open(path, "w", encoding="utf-8") as f:
                json.dump(conv(data), f, ensure_ascii=False, indent=2)

        @staticmethod
        def read_json(path):
            with open(path, "r", encoding="utf-8") as f:
                return json.load(f)

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



def arr_out(a):
    a = np.asarray(a)
    return {"shape": list(a.shape), "dtype": str(a.dtype), "values": a.astype(np.float64).reshape(-1).tolist()}


def similarity_cases(em_mod):
    """EmbeddingManager.compute_similarity / find_most_similar (embedding_manager.py:586-660): the +1e-8 denominators, 1-D
    inputs, 1 / (1 + cdist), argsort[::-1] tie order, the empty / failure sentinels"""
    rng = np.random.default_rng(20261104)
    em = object.__new__(em_mod.EmbeddingManager)
    out = {"compute_similarity": [], "find_most_similar": []}
    a = rng.standard_normal((4, 6)).astype(np.float32)
    b = rng.standard_normal((5, 6)).astype(np.float32)
    bz = b.copy()
    bz[2] = 0.0
    tie = np.vstack([b[0], b[1], b[0], b[3], b[1], b[0]]).astype(np.float32)   # repeated rows: equal similarities
    inputs = [("f32_2d", a, b), ("f64_2d", a.astype(np.float64), b.astype(np.float64)), ("zero_row", a, bz),
              ("one_dim_first", a[0], b), ("one_dim_both", a[1], b[2]), ("one_dim_second", a, b[1]),
              ("empty_first", np.zeros((0, 6), np.float32), b), ("empty_second", a, np.zeros((0, 6), np.float32)),
              ("mismatched_dims", a, rng.standard_normal((3, 5)).astype(np.float32)), ("single_rows", a[:1], b[:1]),
              ("tiny_norms", a * 1e-12, b), ("mixed_dtypes", a, b.astype(np.float64))]
    for name, x, y in inputs:
        for metric in ("cosine", "euclidean", "dot", "manhattan"):
            r = em.compute_similarity(np.array(x), np.array(y), metric=metric)
            out["compute_similarity"].append({"name": name, "metric": metric, "a": arr_out(x), "b": arr_out(y), "expected": arr_out(r)})
    for name, q, c, k in [("plain", a[0], b, 3), ("top_k_beyond", a[1], b, 10), ("ties", b[0], tie, 6), ("ties_top2", b[1], tie, 2),
                          ("empty_query", np.zeros((0,), np.float32), b, 3), ("empty_candidates", a[0], np.zeros((0, 6), np.float32), 3),
                          ("mismatched", a[0], rng.standard_normal((3, 5)).astype(np.float32), 2), ("f64", a[2].astype(np.float64), b.astype(np.float64), 4),
                          ("zero_candidates", a[0], np.zeros((3, 6), np.float32), 3), ("top_k_zero", a[0], b, 0)]:
        for metric in ("cosine", "euclidean", "dot"):
            r = em.find_most_similar(np.array(q), np.array(c), top_k=k, metric=metric)
            out["find_most_similar"].append({"name": name, "metric": metric, "query": arr_out(q), "candidates": arr_out(c), "top_k": k,
                                             "expected": r})
    return out


def index_persistence_cases(vi_mod, fa):
    """VectorIndex.save_index / load_index (vector_index.py:284-364: file names, the sidecar's schema, the returned path,
    what a load restores), get_index_stats, optimize_search_params / _calculate_recall (:428-491)"""
    import tempfile
    rng = np.random.default_rng(20261105)
    out = {"save_load": [], "optimize": [], "calculate_recall": []}

    def new_index(metric, dim, itype, index_dir, nlist=None):
        CONFIG_OVERRIDES.clear()
        CONFIG_OVERRIDES.update({"vector_store.index_type": itype, "vector_store.similarity_metric": metric,
                                 "vector_store.dimension": dim, "storage.vector_index_path": index_dir})
        if nlist is not None:
            CONFIG_OVERRIDES["vector_store.nlist"] = nlist
        v = vi_mod.VectorIndex(dim)
        v.index_dir = index_dir
        return v

    def attrs(v):
        return {k: getattr(v, k) for k in ("index_type", "embedding_dim", "similarity_metric", "total_vectors", "is_trained", "nlist", "nprobe")}

    for name, metric, itype, dim, n, fname in [("flat_cosine_default_name", "cosine", "Flat", 6, 12, None),
                                               ("flat_l2_custom_name", "l2", "Flat", 5, 9, "my_index.faiss"),
                                               ("ivfflat_cosine", "cosine", "IVFFlat", 6, 40, None),
                                               ("name_without_faiss_suffix", "cosine", "Flat", 4, 7, "plain.bin")]:
        d0 = tempfile.mkdtemp(prefix="anr_golden_ix_")
        v = new_index(metric, dim, itype, d0, nlist=4)
        assert v.create_index()
        x = rng.standard_normal((n, dim)).astype(np.float32)
        ids = np.arange(100, 100 + n, dtype=np.int64)
        assert v.add_vectors(x, ids)
        q = rng.standard_normal((3, dim)).astype(np.float32)
        before = v.search(q, top_k=4)
        saved_attrs = attrs(v)
        path = v.save_index(fname) if fname else v.save_index()
        files = sorted(os.listdir(d0))
        meta_name = os.path.basename(path).replace(".faiss", "_metadata.json")
        meta = json.load(open(os.path.join(d0, meta_name))) if os.path.exists(os.path.join(d0, meta_name)) else None
        v2 = new_index("l2" if metric == "cosine" else "cosine", 3, "Flat", d0)   # other settings: the sidecar must win
        loaded = v2.load_index(os.path.basename(path))
        after = v2.search(q, top_k=4)
        out["save_load"].append({"name": name, "metric": metric, "index_type": itype, "dim": dim, "vectors": jf(x), "ids": ids.tolist(),
                                 "queries": jf(q), "filename_arg": fname, "returned_basename": os.path.basename(path),
                                 "returned_dir_is_index_dir": os.path.dirname(path) == d0, "files": files,
                                 "metadata_file": meta_name, "metadata": meta, "attrs_at_save": saved_attrs, "load_returned": loaded,
                                 "attrs_after_load": attrs(v2), "search_before": before, "search_after_load": after,
                                 "stats_after_load": v2.get_index_stats(), "load_missing": v2.load_index("nope.faiss")})
    d0 = tempfile.mkdtemp(prefix="anr_golden_ix_")
    v = new_index("cosine", 4, "Flat", d0)
    out["save_without_index"] = {"returned": v.save_index(), "stats": v.get_index_stats()}
    # a file WITHOUT a sidecar: the attributes keep the loader's own values, total_vectors stays what it was
    v = new_index("cosine", 4, "Flat", d0)
    assert v.create_index() and v.add_vectors(rng.standard_normal((5, 4)).astype(np.float32))
    p = v.save_index("bare.faiss")
    os.remove(p.replace(".faiss", "_metadata.json"))
    v3 = new_index("cosine", 4, "Flat", d0)
    out["load_without_sidecar"] = {"load_returned": v3.load_index("bare.faiss"), "attrs_after_load": attrs(v3)}

    # optimize_search_params: only IVF types; the sweep over nprobe <= nlist; best / early stop; the nprobe left behind
    for name, itype, nlist, n, gt_kind, target in [("flat_is_refused", "Flat", 4, 30, "exact", 0.9),
                                                   ("ivf_exact_truth_stops_at_first", "IVFFlat", 64, 200, "exact", 0.9),
                                                   ("ivf_half_wrong_truth_sweeps", "IVFFlat", 64, 200, "half", 0.9),
                                                   ("ivf_small_nlist_breaks", "IVFFlat", 8, 60, "half", 0.9),
                                                   ("ivfpq_low_target", "IVFPQ", 64, 200, "half", 0.4),
                                                   ("ivf_single_query", "IVFFlat", 64, 200, "exact1", 0.9)]:
        d0 = tempfile.mkdtemp(prefix="anr_golden_ix_")
        v = new_index("cosine", 8, itype, d0, nlist=nlist)
        assert v.create_index()
        x = rng.standard_normal((n, 8)).astype(np.float32)
        assert v.add_vectors(x)
        nq = 1 if gt_kind == "exact1" else 5
        q = rng.standard_normal((nq, 8)).astype(np.float32)
        xn = x / np.linalg.norm(x, axis=1, keepdims=True)
        qn = q / np.linalg.norm(q, axis=1, keepdims=True)
        truth = np.argsort(-(qn @ xn.T), axis=1, kind="stable")[:, :6]
        if gt_kind == "half":
            truth = truth.copy()
            truth[:, 3:] = n + 1000 + np.arange(3)        # rows that do not exist: recall can reach 0.5 at most
        nprobe_before = v.nprobe
        try:   # (one query: search returns a FLAT list and _calculate_recall then indexes a str — the reference raises)
            res = v.optimize_search_params(q, truth, target_recall=target)
            raised = None
        except Exception as e:
            res, raised = None, type(e).__name__
        out["optimize"].append({"name": name, "index_type": itype, "nlist": nlist, "vectors": jf(x), "queries": jf(q),
                                "ground_truth": truth.tolist(), "target_recall": target, "nprobe_before": nprobe_before,
                                "expected": res, "raises": raised, "nprobe_after": v.nprobe, "nlist_after": v.nlist,
                                "index_nprobe_after": getattr(v.index, "nprobe", None)})
    v = new_index("cosine", 4, "Flat", tempfile.mkdtemp(prefix="anr_golden_ix_"))
    hits = lambda *rows: [[{"index": i} for i in r] for r in rows]
    for name, res, gt in [("plain", hits([1, 2, 3], [4, 5, 6]), [[1, 2, 9], [7, 8, 9]]),
                          ("empty_results", [], [[1]]), ("empty_truth", hits([1]), []),
                          ("more_results_than_truth", hits([1, 2], [3, 4], [5, 6]), [[1, 2], [9, 9]]),
                          ("more_truth_than_results", hits([1, 2]), [[1, 5], [3], [4]]),
                          ("empty_truth_row", hits([1, 2], [3]), [[], [3]]),
                          ("duplicates_in_truth", hits([1, 2, 3]), [[1, 1, 2, 7]])]:
        out["calculate_recall"].append({"name": name, "search_results": res, "ground_truth": gt,
                                        "expected": v._calculate_recall(res, np.array(gt, dtype=object) if gt and len({len(g) for g in gt}) > 1 else np.array(gt))})
    return out


def retriever_persistence_cases(rt_mod, em_mod, stm):
    """VectorRetriever._save_index_data / _can_load_existing_index (retriever.py:680-749: the file set, npz keys, the
    id_mappings.json schema, when an existing index is taken), optimize_retrieval / _calculate_f1_score / get_retrieval_stats
    (:751-860)"""
    import tempfile
    rng = np.random.default_rng(20261106)
    d = 8
    em = object.__new__(em_mod.EmbeddingManager)
    em.model = stm.SentenceTransformer()
    em.model.dim = d
    em.model_name, em.batch_size, em.device, em.max_length = "BAAI/bge-m3", 32, "cpu", 512
    em.normalize_embeddings, em.embedding_dim, em.consistency_checker = True, d, None
    em_mod.EmbeddingManager._instance = em
    em_mod.EmbeddingManager._model_loaded = True
    out = {}

    def fresh(data_dir, index_dir, itype="Flat"):
        CONFIG_OVERRIDES.clear()
        CONFIG_OVERRIDES.update({"vector_store.index_type": itype, "vector_store.similarity_metric": "cosine",
                                 "storage.vector_store_path": data_dir, "storage.vector_index_path": index_dir})
        r = rt_mod.VectorRetriever()
        r.data_dir = data_dir
        r.vector_index.index_dir = index_dir
        return r

    notes = FakeNotes.make(rng, 14)
    for i, n in enumerate(notes):
        if not isinstance(n["content"], str):
            n["content"] = f"plain content {i} paris river"
    data_dir = tempfile.mkdtemp(prefix="anr_golden_rd_")
    r = fresh(data_dir, data_dir)            # the reference loads `index_files[0]` from data_dir through vector_index.load_index:
    em.model.calls = []                      # it only finds the file when both directories are the same one
    ok = r.build_index([dict(n) for n in notes], force_rebuild=True, save_index=True)
    files = sorted(os.listdir(data_dir))
    npz = np.load(os.path.join(data_dir, "note_embeddings.npz"))
    out["save"] = {"notes": notes, "build_returned": ok, "files": files, "npz_keys": sorted(npz.files),
                   "npz_embeddings_shape": list(npz["embeddings"].shape), "npz_embeddings_dtype": str(npz["embeddings"].dtype),
                   "atomic_notes_json": json.load(open(os.path.join(data_dir, "atomic_notes.json"))),
                   "id_mappings_json": json.load(open(os.path.join(data_dir, "id_mappings.json"))),
                   "vectors": {k: [float(t) for t in v] for k, v in em.model.seen.items()}}
    can = []
    for name, cand in [("same_notes", [dict(n) for n in notes]), ("other_count", [dict(n) for n in notes[:-1]]),
                       ("other_first_id", [dict(notes[1])] + [dict(n) for n in notes[1:]]),
                       ("same_count_same_first_other_rest", [dict(notes[0])] + [dict(n, note_id="zzz") for n in notes[1:]])]:
        r2 = fresh(data_dir, data_dir)
        got = r2._can_load_existing_index(cand)
        can.append({"name": name, "candidate_ids": [n.get("note_id") for n in cand], "returned": got,
                    "n_notes_after": len(r2.atomic_notes), "total_vectors_after": int(r2.vector_index.total_vectors),
                    "embeddings_shape_after": list(r2.note_embeddings.shape) if r2.note_embeddings is not None else None,
                    "note_id_to_index_after": dict(r2.note_id_to_index)})
    empty_dir = tempfile.mkdtemp(prefix="anr_golden_rd_")
    r3 = fresh(empty_dir, empty_dir)
    can.append({"name": "empty_directory", "candidate_ids": [n["note_id"] for n in notes], "returned": r3._can_load_existing_index(notes)})
    only_index = tempfile.mkdtemp(prefix="anr_golden_rd_")
    open(os.path.join(only_index, "x.faiss"), "wb").write(b"junk")
    r4 = fresh(only_index, only_index)
    can.append({"name": "index_file_without_notes_file", "candidate_ids": [n["note_id"] for n in notes],
                "returned": r4._can_load_existing_index(notes)})
    out["can_load"] = can
    # build_index without force_rebuild takes the existing index: no encoder call
    r5 = fresh(data_dir, data_dir)
    em.model.calls = []
    ok = r5.build_index([dict(n) for n in notes], force_rebuild=False, save_index=False)
    out["build_reuses_existing"] = {"returned": ok, "encoder_calls": len(em.model.calls), "n_notes": len(r5.atomic_notes)}

    # get_retrieval_stats: with embeddings present the reference calls a method its EmbeddingManager does not have
    def stats_of(rr):
        try:
            return {"returned": rr.get_retrieval_stats()}
        except Exception as e:
            return {"raises": type(e).__name__}
    r6 = fresh(tempfile.mkdtemp(prefix="anr_golden_rd_"), tempfile.mkdtemp(prefix="anr_golden_ri_"))
    out["stats_empty"] = stats_of(r6)
    out["stats_built"] = stats_of(r)

    # _calculate_f1_score
    hits = lambda *rows: [[{"note_id": i} for i in rr] for rr in rows]
    f1 = []
    for name, res, gt in [("plain", hits(["a", "b", "c"], ["d"]), [["a", "x"], ["d"]]), ("no_results", [], [["a"]]),
                          ("no_truth", hits(["a"]), []), ("empty_truth_rows_skipped", hits(["a"], ["b"]), [[], ["b"]]),
                          ("nothing_retrieved", hits([], ["b"]), [["a"], ["b", "c"]]), ("disjoint", hits(["a"]), [["z"]]),
                          ("all_truth_rows_empty", hits(["a"]), [[]]), ("missing_note_id_key", [[{"x": 1}, {"note_id": "a"}]], [["a"]]),
                          ("more_results_than_truth", hits(["a"], ["b"], ["c"]), [["a"]])]:
        f1.append({"name": name, "search_results": res, "ground_truth": gt, "expected": r._calculate_f1_score(res, gt)})
    out["f1"] = f1

    # optimize_retrieval on the built retriever (Flat: index_optimization == {}), and on an IVFFlat one
    prefix = "Represent this sentence for searching relevant passages: "
    opt = []
    for itype in ("Flat", "IVFFlat"):
        dd = tempfile.mkdtemp(prefix="anr_golden_rd_")
        rr = fresh(dd, dd, itype)
        em.model.calls = []
        assert rr.build_index([dict(n) for n in notes], force_rebuild=True, save_index=False)
        emb = rr.note_embeddings
        queries = ["first probe", "second probe", "third probe"]
        near = [2, 5, 9]
        table = {}
        for qt, j in zip(queries, near):
            v = emb[j] + 0.25 * rng.standard_normal(d).astype(np.float32)
            table[prefix + qt] = (v / np.linalg.norm(v)).astype(np.float32)
        em.model.table = table
        gt = [[notes[2]["note_id"], notes[3]["note_id"]], [notes[5]["note_id"]], [notes[9]["note_id"], "unknown_id"]]
        thr_before = rr.similarity_threshold
        res = rr.optimize_retrieval(queries, gt, target_recall=0.8)
        opt.append({"index_type": itype, "queries": queries, "ground_truth": gt, "target_recall": 0.8,
                    "query_vectors": {k: [float(t) for t in v] for k, v in table.items()},
                    "note_vectors": {k: [float(t) for t in v] for k, v in em.model.seen.items()},
                    "threshold_before": thr_before, "expected": res, "threshold_after": rr.similarity_threshold})
        em.model.table = {}
    out["optimize_retrieval"] = opt
    out["optimize_retrieval_no_data"] = {"no_queries": r.optimize_retrieval([], [["a"]]), "no_truth": r.optimize_retrieval(["q"], [])}
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
    with open(os.path.join(HERE, "similarity_cases.json"), "w") as f:
        json.dump({"source": "reference vector_store/embedding_manager.py compute_similarity / find_most_similar (scipy's cdist is "
                             "the real one)", "note": note, **similarity_cases(em_mod)}, f)
    with open(os.path.join(HERE, "index_persistence_cases.json"), "w") as f:
        json.dump({"source": "reference vector_store/vector_index.py save_index / load_index / get_index_stats / "
                             "optimize_search_params / _calculate_recall", "note": note + "; the IVF stand-in is an exact search "
                             "that carries nlist / nprobe, faiss.write_index / read_index bring the stand-in object back",
                   **index_persistence_cases(vi_mod, fa)}, f, default=lambda o: o.item() if hasattr(o, "item") else str(o))
    with open(os.path.join(HERE, "retriever_persistence_cases.json"), "w") as f:
        json.dump({"source": "reference vector_store/retriever.py _save_index_data / _can_load_existing_index / optimize_retrieval "
                             "/ _calculate_f1_score / get_retrieval_stats", "note": note,
                   **retriever_persistence_cases(rt_mod, em_mod, stm)}, f, default=lambda o: o.item() if hasattr(o, "item") else str(o))
    print("wrote similarity_cases.json, index_persistence_cases.json, retriever_persistence_cases.json")
    print("wrote vector_index_facade_cases.json, retriever_facade_cases.json, retriever_lifecycle_cases.json, embedding_manager_facade_cases.json")


if __name__ == "__main__":
    main()
