#!/usr/bin/env python3
"""Generate the golden fusion / BM25 vectors from the REFERENCE itself (run in the build container only).

  PYTHONHASHSEED=0 python tests/golden/make_golden.py [/root/reference]

Imports, by file path, the two reference files of the hot path that are importable here
(SURVEY.md §8c): retrieval/hybrid_search.py (stdlib only) and utils/bm25_search.py (needs a `loguru`
stub, the same sys.modules technique the reference's own tests use for GPUtil).  Writes
  tests/golden/fusion_cases.json   inputs + the reference's HybridSearcher.fuse output
  tests/golden/bm25_cases.json     notes + queries + the reference's bm25_scores output (SimpleBM25 variant)
  tests/golden/similarity_relation_cases.json   embeddings + the reference's semantic-similarity relations
      (graph/relation_extractor.py:591-629, 769-791).  That file's module-level imports (`utils`, `config`) pull
      in packages that are absent here; the three methods used are pure numpy, so the two names are satisfied by
      empty stand-in modules whose `config.get(key, default)` returns the default — the arithmetic that runs is
      the reference's own.
Only data is written; no reference source is copied.  The GPU box never runs this script.
"""
import importlib.util
import json
import os
import random
import sys
import types

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def stub_loguru():
    m = types.ModuleType("loguru")

    class _L:
        def __getattr__(self, _):
            return lambda *a, **k: None

    m.logger = _L()
    sys.modules["loguru"] = m


def rand_list(rng, ids, n, lo=0.0, hi=1.0, ties=False, dup=False):
    pick = rng.sample(ids, min(n, len(ids)))
    out = []
    for i in pick:
        s = rng.uniform(lo, hi)
        if ties:
            s = round(s, 1)
        out.append([i, s])
    if dup and out:
        out.append([out[0][0], rng.uniform(lo, hi)])  # later duplicate overwrites (hybrid_search.py:54-59)
    return out


def fusion_cases(hs):
    rng = random.Random(20251031)
    ids = [f"note_{i:05d}" for i in range(400)]
    weights = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
    cases = []

    def add(name, method, dense, bm25, graph, path, pool=50, w=weights, rrf_k=60, enabled=True):
        cfg = {"retrieval": {"candidate_pool": pool,
                             "hybrid": {"enabled": enabled, "fusion_method": method, "weights": w, "rrf_k": rrf_k}}}
        searcher = hs.HybridSearcher(cfg)
        out = searcher.fuse(dense=[tuple(x) for x in dense] if dense is not None else None,
                            bm25=[tuple(x) for x in bm25] if bm25 is not None else None,
                            graph=[tuple(x) for x in graph] if graph is not None else None,
                            path=[tuple(x) for x in path] if path is not None else None)
        cases.append({"name": name, "config": cfg, "dense": dense, "bm25": bm25, "graph": graph, "path": path,
                      "expected": out})

    # the survey's hand example
    ex = dict(dense=[["a", .9], ["b", .8], ["c", .1]], bm25=[["b", 3], ["d", 2]], graph=None, path=[["d", 1]])
    add("survey_example_rrf", "rrf", **ex)
    add("survey_example_linear", "linear", **ex)
    for method in ("linear", "rrf"):
        for t in range(6):
            add(f"{method}_random_{t}", method,
                rand_list(rng, ids, rng.randint(1, 120)), rand_list(rng, ids, rng.randint(1, 200), 0, 9.0),
                rand_list(rng, ids, rng.randint(0, 40)), rand_list(rng, ids, rng.randint(0, 20)),
                pool=rng.choice([10, 50, 80]), rrf_k=rng.choice([1, 60]))
        add(f"{method}_ties", method, rand_list(rng, ids, 60, ties=True), rand_list(rng, ids, 60, ties=True),
            rand_list(rng, ids, 30, ties=True), rand_list(rng, ids, 10, ties=True), pool=500)
        add(f"{method}_duplicates", method, rand_list(rng, ids, 30, dup=True), rand_list(rng, ids, 30, dup=True),
            rand_list(rng, ids, 5, dup=True), rand_list(rng, ids, 5, dup=True), pool=500)
        add(f"{method}_empty_all", method, [], [], [], [])
        add(f"{method}_none_all", method, None, None, None, None)
        add(f"{method}_dense_only", method, rand_list(rng, ids, 25), None, None, None)
        add(f"{method}_bm25_only", method, None, rand_list(rng, ids, 25, 0, 12.0), None, None)
        add(f"{method}_path_only_ids", method, rand_list(rng, ids[:50], 20), [], [],
            rand_list(rng, ids[300:], 10), pool=500)
        add(f"{method}_zero_max", method, [[i, 0.0] for i in ids[:10]], [[i, 0.0] for i in ids[5:15]],
            [], [], pool=500)
        add(f"{method}_negative_scores", method, rand_list(rng, ids, 20, -1.0, -0.1), rand_list(rng, ids, 20, -2, 2),
            [], [], pool=500)
        add(f"{method}_missing_weights", method, rand_list(rng, ids, 20), rand_list(rng, ids, 20),
            rand_list(rng, ids, 20), rand_list(rng, ids, 5), w={"dense": 0.7}, pool=500)
        add(f"{method}_disabled", method, rand_list(rng, ids, 5), None, None, None, enabled=False)
        add(f"{method}_pool_1", method, rand_list(rng, ids, 50), rand_list(rng, ids, 50), [], [], pool=1)
    return cases


def bm25_cases(bm):
    rng = random.Random(7)
    vocab = [f"w{i}" for i in range(300)] + ["natural", "language", "processing", "machine", "learning", "AI"]
    cases = []
    # the module's own sample (utils/bm25_search.py:346-351)
    notes = [
        {"title": "Machine Learning", "content": "Machine learning is a subset of artificial intelligence"},
        {"title": "Deep Learning", "content": "Deep learning uses neural networks with multiple layers"},
        {"title": "Natural Language Processing", "content": "NLP deals with text and language understanding"},
        {"title": "AI", "content": "natural language and more natural language"},
    ]

    def text_fn(n):
        return f"{n.get('title', '')} {n.get('content', '')}"

    def add(name, notes, queries):
        corpus = bm.build_bm25_corpus(notes, text_fn)
        assert type(corpus).__name__ == "SimpleBM25", "golden vectors are for the SimpleBM25 variant"
        cases.append({"name": name, "notes": notes, "queries": queries,
                      "expected": [bm.bm25_scores(corpus, notes, q) for q in queries],
                      "tokens": [bm.tokenize_text(q) for q in queries]})

    add("module_sample", notes, ["natural language", "machine learning", "", "zzz unknown", "Deep, deep LEARNING!"])
    for t in range(4):
        nn = []
        for i in range(rng.randint(5, 60)):
            nn.append({"title": " ".join(rng.choices(vocab, k=rng.randint(0, 4))),
                       "content": " ".join(rng.choices(vocab, k=rng.randint(0, 40)))})
        qs = [" ".join(rng.choices(vocab, k=rng.randint(1, 6))) for _ in range(5)]
        add(f"random_{t}", nn, qs)
    add("empty_docs", [{"title": "", "content": ""}, {"title": "a b", "content": "c"}, {"title": "", "content": ""}],
        ["a", "c b", "q"])
    return cases


def stub_utils_and_config():
    u = types.ModuleType("utils")

    class _Any:
        def __init__(self, *a, **k):
            pass

    u.TextUtils = u.GPUUtils = u.BatchProcessor = _Any
    sys.modules["utils"] = u
    c = types.ModuleType("config")

    class _Cfg:
        def get(self, key, default=None):
            return default

    c.config = _Cfg()
    sys.modules["config"] = c


def similarity_cases(rx):
    import numpy as np
    ext = rx.RelationExtractor()
    cases = []

    def add(name, emb):
        notes = [{"note_id": f"n{i:04d}"} for i in range(emb.shape[0])]
        rel = ext.extract_semantic_similarity_relations(notes, emb)
        cases.append({"name": name, "threshold": ext.similarity_threshold,
                      "weight": ext.relation_weights["semantic_similarity"],
                      "embeddings": [[float(v) for v in row] for row in emb],
                      "expected": [{"source_id": r["source_id"], "target_id": r["target_id"], "weight": float(r["weight"]),
                                    "cosine_similarity": r["metadata"]["cosine_similarity"],
                                    "similarity_rank": int(r["metadata"]["similarity_rank"])} for r in rel]})

    rng = np.random.default_rng(20251031)
    for t, (n, d, k, sigma) in enumerate([(24, 16, 4, 0.35), (60, 32, 6, 0.5), (40, 8, 3, 0.6)]):
        cent = rng.standard_normal((k, d))
        emb = (cent[rng.integers(0, k, n)] + sigma * rng.standard_normal((n, d))).astype(np.float32)
        add(f"clusters_{t}", emb)
    emb = rng.standard_normal((20, 12)).astype(np.float32)
    emb[3] = 0.0                       # zero row: norm replaced by 1, similarity 0 (relation_extractor.py:772-773)
    emb[7] = 2.5 * emb[5]              # scaled copy: cosine 1
    emb[11] = emb[5] + 0.05 * rng.standard_normal(12).astype(np.float32)
    add("zero_row_and_copies", emb)
    add("nothing_above_threshold", np.eye(10, dtype=np.float32))
    return cases


def main():
    stub_loguru()
    hs = load_by_path("ref_hybrid_search", os.path.join(REF, "retrieval", "hybrid_search.py"))
    bm = load_by_path("ref_bm25_search", os.path.join(REF, "utils", "bm25_search.py"))
    with open(os.path.join(HERE, "fusion_cases.json"), "w") as f:
        json.dump({"source": "reference retrieval/hybrid_search.py HybridSearcher.fuse", "cases": fusion_cases(hs)}, f)
    with open(os.path.join(HERE, "bm25_cases.json"), "w") as f:
        json.dump({"source": "reference utils/bm25_search.py build_bm25_corpus + bm25_scores (SimpleBM25)",
                   "cases": bm25_cases(bm)}, f)
    stub_utils_and_config()
    rx = load_by_path("ref_relation_extractor", os.path.join(REF, "graph", "relation_extractor.py"))
    with open(os.path.join(HERE, "similarity_relation_cases.json"), "w") as f:
        json.dump({"source": "reference graph/relation_extractor.py extract_semantic_similarity_relations",
                   "cases": similarity_cases(rx)}, f)
    print("wrote fusion_cases.json, bm25_cases.json, similarity_relation_cases.json")


if __name__ == "__main__":
    main()
