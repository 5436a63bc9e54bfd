#!/usr/bin/env python3
"""Generate the golden fusion / BM25 vectors from the REFERENCE itself (run in the build container only).

  PYTHONHASHSEED=0 python tests/golden/make_golden.py [/root/reference]

Imports, by file path, the two reference files of the hot path that are importable here
(SURVEY.md §8c): retrieval/hybrid_search.py (stdlib only) and utils/bm25_search.py (needs a `loguru`
stub, the same sys.modules technique the reference's own tests use for GPUtil).  Writes
  tests/golden/fusion_cases.json   inputs + the reference's HybridSearcher.fuse output
  tests/golden/bm25_cases.json     notes + queries + the reference's bm25_scores output (SimpleBM25 variant)
  tests/golden/fusion_rrf_multi_long_cases.json   rrf with two or three FULL-corpus lists
  tests/golden/fusion_long_cases.json   HybridSearcher.fuse where bm25 (and, in some cases, dense) is a FULL-corpus
      list — one (id, score) entry per note, what zipping the note ids with bm25_scores() gives — with integer note
      ids; the long lists are stored sparsely (non-zero entries + the value every other entry has)
  tests/golden/embedding_candidates_cases.json   GraphRetriever._find_embedding_candidates
      (graph/graph_retriever.py:153-170), with `graph.graph_index` / `config` satisfied by stand-in modules
  tests/golden/candidate_fusion_cases.json   QueryProcessor._hybrid_search / _enhanced_hybrid_search_v2
      (query/query_processor.py:3680-3768, :1088-1143): query_processor.py cannot be imported here (LLM clients,
      rerankers, ...), so the two METHODS are taken from its source with `ast`, compiled as they stand and run on a
      stand-in `self` whose collaborators (vector / bm25 similarities, penalties, boosts) return the case's inputs
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


def fusion_long_cases(hs):
    """full-corpus bm25 vectors (N entries, mostly zero) fused with a dense top-100, small graph / path lists"""
    import numpy as np
    rng = np.random.default_rng(20261004)
    weights = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
    cases = []

    def sparse_vec(n, frac, ties=False, fill=0.0, normalise=True):
        v = np.full(n, fill, dtype=np.float64)
        nz = rng.choice(n, size=max(1, int(n * frac)), replace=False)
        vals = np.abs(rng.standard_normal(len(nz)))
        if ties:
            vals = np.round(vals, 1)
        if normalise and vals.max() > 0:
            vals = vals / vals.max()          # bm25_scores divides by the maximum (bm25_search.py:329-333)
        v[nz] = vals
        return v

    def pairs(n_total, m, lo=0.0, hi=1.0, ties=False):
        ids = rng.choice(n_total, size=m, replace=False)
        sc = rng.uniform(lo, hi, m)
        if ties:
            sc = np.round(sc, 1)
        order = np.argsort(-sc, kind="stable")
        return [[int(i), float(s)] for i, s in zip(ids[order], sc[order])]

    def add(name, method, n, bm25_vec, dense, graph, path, dense_vec=None, pool=80, rrf_k=60, w=weights):
        cfg = {"retrieval": {"candidate_pool": pool,
                             "hybrid": {"enabled": True, "fusion_method": method, "weights": w, "rrf_k": rrf_k}}}
        searcher = hs.HybridSearcher(cfg)
        bm25 = [(i, float(bm25_vec[i])) for i in range(n)] if bm25_vec is not None else None
        d = [(i, float(dense_vec[i])) for i in range(n)] if dense_vec is not None else [tuple(x) for x in dense]
        out = searcher.fuse(dense=d, bm25=bm25, graph=[tuple(x) for x in graph], path=[tuple(x) for x in path])

        def pack(vec):
            if vec is None:
                return None
            vals, counts = np.unique(vec, return_counts=True)
            fill = float(vals[np.argmax(counts)])
            nz = np.nonzero(vec != fill)[0]
            return {"n": int(n), "fill": fill, "idx": [int(i) for i in nz], "val": [float(vec[i]) for i in nz]}

        cases.append({"name": name, "config": cfg, "n": int(n), "bm25_vec": pack(bm25_vec), "dense_vec": pack(dense_vec),
                      "dense": dense if dense_vec is None else None, "graph": graph, "path": path, "expected": out})

    for method in ("linear", "rrf"):
        for t, (n, frac) in enumerate([(20_000, 0.004), (50_000, 0.001), (9_000, 0.02)]):
            add(f"{method}_bm25_full_{t}", method, n, sparse_vec(n, frac), pairs(n, 100, 0.2, 0.9),
                pairs(n, int(rng.integers(0, 30))), pairs(n, int(rng.integers(0, 10))),
                pool=int(rng.choice([10, 50, 80])), rrf_k=int(rng.choice([1, 60])))
        n = 12_000
        add(f"{method}_bm25_ties", method, n, sparse_vec(n, 0.02, ties=True, normalise=False),
            pairs(n, 100, 0.0, 1.0, ties=True), pairs(n, 20, ties=True), pairs(n, 8, ties=True), pool=120)
        add(f"{method}_bm25_all_zero", method, n, np.zeros(n), pairs(n, 50, 0.1, 0.8), [], [], pool=60)
        add(f"{method}_bm25_dense_overlap", method, n, sparse_vec(n, 0.01),
            # dense hits placed on bm25 hits and on the first ids (which win the zero ties)
            [[int(i), float(s)] for i, s in zip(list(range(0, 40)) + [int(x) for x in rng.choice(n, 60, replace=False)],
                                                sorted(rng.uniform(0.3, 0.95, 100), reverse=True))],
            pairs(n, 25), pairs(n, 6), pool=80)
        add(f"{method}_bm25_negative_fill", method, 8_000, sparse_vec(8_000, 0.01, fill=-0.25, normalise=False),
            pairs(8_000, 100, -0.5, 0.9), [], pairs(8_000, 5), pool=50)
    # linear only: two full-length sources (a dense score for EVERY note as well)
    n = 6_000
    add("linear_two_full_vectors", "linear", n, sparse_vec(n, 0.01), None, pairs(n, 12), pairs(n, 4),
        dense_vec=rng.uniform(-0.2, 0.9, n), pool=80)
    return cases


def fusion_rrf_multi_long_cases(hs):
    """rrf where TWO or THREE sources are full-corpus lists (a dense score and a bm25 score for EVERY note): the
    reference ranks each list by a stable descending sort, so every note has a rank in each of them"""
    import numpy as np
    rng = np.random.default_rng(20261007)
    cases = []

    def run(name, n, vecs, short, weights, pool, rrf_k=60):
        cfg = {"retrieval": {"candidate_pool": pool,
                             "hybrid": {"enabled": True, "fusion_method": "rrf", "weights": weights, "rrf_k": rrf_k}}}
        searcher = hs.HybridSearcher(cfg)
        lists = {}
        for k in ("dense", "bm25", "graph", "path"):
            if k in vecs:
                lists[k] = [(i, float(vecs[k][i])) for i in range(n)]
            else:
                lists[k] = [tuple(x) for x in short.get(k, [])]
        out = searcher.fuse(dense=lists["dense"], bm25=lists["bm25"], graph=lists["graph"], path=lists["path"])
        cases.append({"name": name, "config": cfg, "n": int(n),
                      "vectors": {k: [float(x) for x in v] for k, v in vecs.items()},
                      "short": {k: [[int(i), float(sc)] for i, sc in v] for k, v in short.items()}, "expected": out})

    def pairs(n_total, m):
        ids = rng.choice(n_total, size=m, replace=False)
        sc = np.sort(rng.uniform(0, 1, m))[::-1]
        return [[int(i), float(x)] for i, x in zip(ids, sc)]

    w = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
    n = 6_000
    bm = np.zeros(n); nz = rng.choice(n, 60, replace=False); bm[nz] = np.abs(rng.standard_normal(60))
    run("rrf_two_full_vectors", n, {"dense": rng.uniform(-0.2, 0.9, n), "bm25": bm}, {"graph": pairs(n, 12), "path": pairs(n, 4)}, w, 80)
    n = 5_000
    run("rrf_two_full_ties", n, {"dense": np.round(rng.uniform(0, 1, n), 1), "bm25": np.round(np.abs(rng.standard_normal(n)), 1)},
        {"graph": pairs(n, 20), "path": pairs(n, 6)}, w, 100, rrf_k=1)
    # mirrored rankings with equal weights: notes i and n-1-i get exactly equal finals -> the tie rule decides
    lin = np.linspace(1.0, 0.0, n)
    run("rrf_two_full_mirrored", n, {"dense": lin, "bm25": lin[::-1].copy()}, {}, {"dense": 1.0, "bm25": 1.0, "graph": 0.5, "path": 0.1}, 40)
    n = 4_500
    g = np.zeros(n); g[rng.choice(n, 200, replace=False)] = rng.uniform(0.1, 1.0, 200)
    run("rrf_three_full_vectors", n, {"dense": rng.standard_normal(n), "bm25": np.abs(rng.standard_normal(n)) * (rng.random(n) < 0.05), "graph": g},
        {"path": pairs(n, 9)}, w, 64)
    return cases


def embedding_candidate_cases(gr):
    import numpy as np
    rng = np.random.default_rng(20261005)
    cases = []
    for t, (n, d, k) in enumerate([(300, 16, 15), (40, 8, 15), (9, 8, 15), (120, 12, 5)]):
        emb = (rng.standard_normal((n, d)) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)  # un-normalised rows
        q = rng.standard_normal(d).astype(np.float32)
        idx = types.SimpleNamespace(graph=None, embeddings=emb,
                                    note_id_to_index={f"note_{i:04d}": i for i in range(n)})
        r = gr.GraphRetriever(idx)
        cases.append({"name": f"case_{t}", "top_k": k, "embeddings": [[float(v) for v in row] for row in emb],
                      "query": [float(v) for v in q], "expected": r._find_embedding_candidates(q, top_k=k)})
    return cases


def load_graph_retriever():
    """graph/graph_retriever.py imports `.graph_index` relatively and `config` / loguru / networkx at module level;
    networkx is installed, the other three are stand-ins (no arithmetic lives in them)"""
    pkg = types.ModuleType("graph")
    pkg.__path__ = [os.path.join(REF, "graph")]
    sys.modules["graph"] = pkg
    gi = types.ModuleType("graph.graph_index")
    gi.GraphIndex = type("GraphIndex", (), {})
    sys.modules["graph.graph_index"] = gi
    return load_by_path("graph.graph_retriever", os.path.join(REF, "graph", "graph_retriever.py"))


def candidate_fusion_cases():
    """runs the reference's own _hybrid_search / _enhanced_hybrid_search_v2 bodies on synthetic candidates"""
    import ast
    import numpy as np
    path = os.path.join(REF, "query", "query_processor.py")
    src = open(path, encoding="utf-8").read()
    tree = ast.parse(src)
    wanted = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "QueryProcessor":
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name in ("_hybrid_search", "_enhanced_hybrid_search_v2"):
                    fn.decorator_list = []
                    wanted[fn.name] = fn

    class _Log:
        def __getattr__(self, _):
            return lambda *a, **k: None

    from typing import Any, Dict, List
    ns = {"List": List, "Dict": Dict, "Any": Any, "logger": _Log(), "config": types.SimpleNamespace(get=lambda k, d=None: d)}
    mod = ast.Module(body=list(wanted.values()), type_ignores=[])
    exec(compile(ast.fix_missing_locations(mod), path, "exec"), ns)
    bm_stub = types.ModuleType("utils.bm25_search")
    sys.modules.setdefault("utils", types.ModuleType("utils"))
    sys.modules["utils.bm25_search"] = bm_stub
    rng = np.random.default_rng(20261006)
    vocab = [f"w{i}" for i in range(40)]
    cases = []

    def cands(n, ties=False):
        out = []
        for i in range(n):
            out.append({"note_id": f"n{i}", "content": " ".join(rng.choice(vocab, size=int(rng.integers(3, 12)))),
                        "title": f"t{i % 5}"})
        vs = rng.uniform(0.05, 0.95, n)
        bs = np.abs(rng.standard_normal(n))
        bs = bs / bs.max() if n else bs
        bs[rng.random(n) < 0.4] = 0.0
        if ties:
            vs, bs = np.round(vs, 1), np.round(bs, 1)
        return out, [float(v) for v in vs], [float(v) for v in bs]

    for method in ("linear", "rrf"):
        for t, (n, ties) in enumerate([(30, False), (120, False), (60, True), (1, False), (7, True)]):
            c, vs, bs = cands(n, ties)
            must = [str(w) for w in rng.choice(vocab, 2)] if t != 3 else None
            ents = [str(w) for w in rng.choice(vocab, 3)] if t % 2 == 0 else None
            preds = [str(w) for w in rng.choice(vocab, 3)] if t % 2 == 1 or t == 0 else None
            self = types.SimpleNamespace(structured_logger=_Log(), hybrid_search_enabled=True, bm25_corpus=True,
                                         fusion_method=method, vector_weight=0.7, bm25_weight=0.3, rrf_k=60,
                                         _calculate_vector_similarities=lambda q, cc, vs=vs: list(vs),
                                         _fallback_vector_search=lambda q, cc: (_ for _ in ()).throw(RuntimeError("fallback")))
            bm_stub.bm25_scores = lambda corpus, docs, query, bs=bs: list(bs)
            inp = [dict(x) for x in c]
            out = ns["_hybrid_search"](self, "query", [dict(x) for x in c], must, ents, preds)
            cases.append({"name": f"hybrid_{method}_{t}", "kind": method, "vector_weight": 0.7, "bm25_weight": 0.3, "rrf_k": 60,
                          "candidates": inp, "vector_scores": vs, "bm25_scores": bs, "must_have_terms": must,
                          "boost_entities": ents, "boost_predicates": preds,
                          "expected": [{"note_id": x["note_id"], "hybrid_score": x["hybrid_score"]} for x in out]})
    for t, n in enumerate([25, 90, 40, 3]):
        c, vs, bs = cands(n, ties=(t == 2))
        sec = [float(rng.choice([1.0, 0.5, 0.8])) for _ in range(n)]
        lex = [float(rng.choice([1.0, 0.3])) for _ in range(n)]
        ok = [bool(rng.random() < 0.5) for _ in range(n)]
        eb = [float(rng.choice([1.0, 1.2, 1.44])) for _ in range(n)]
        pb = [float(rng.choice([1.0, 1.15])) for _ in range(n)]
        for x, a1, a2, a3, a4, a5 in zip(c, sec, lex, ok, eb, pb):
            x.update(_sec=a1, _lex=a2, _ok=a3, _eb=a4, _pb=a5)
        must = ["x"] if t != 3 else None
        self = types.SimpleNamespace(_calculate_vector_similarities=lambda q, cc, vs=vs: list(vs),
                                     _calculate_bm25_similarities=lambda q, cc, bs=bs: list(bs),
                                     section_filtering_enabled=(t != 1), lexical_fallback_enabled=True,
                                     noise_threshold=0.35, listt5=None,
                                     _apply_section_filtering=lambda cand, q: cand["_sec"],
                                     _apply_lexical_fallback=lambda cand, terms: cand["_lex"],
                                     _satisfies_must_have_terms=lambda cand, terms: cand["_ok"],
                                     _calculate_entity_boost=lambda cand, e: cand["_eb"],
                                     _calculate_predicate_boost=lambda cand, pr: cand["_pb"])
        ents = ["e"] if t % 2 == 0 else None
        preds = ["p"] if t < 3 else None
        out = ns["_enhanced_hybrid_search_v2"](self, "query", [dict(x) for x in c], must, ents, preds)
        cases.append({"name": f"v2_{t}", "kind": "v2", "noise_threshold": 0.35, "section_filtering_enabled": t != 1,
                      "candidates": c, "vector_scores": vs, "bm25_scores": bs, "must_have_terms": must,
                      "boost_entities": ents, "boost_predicates": preds,
                      "expected": [{"note_id": x["note_id"], "final_base_score": x["final_base_score"]} for x in out]})
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


def bm25_field_cases(bm):
    """FieldWeightedBM25 (utils/bm25_search.py:66-234): title / entities / content with weights"""
    rng = random.Random(11)
    vocab = [f"w{i}" for i in range(120)]
    cases = []

    def add(name, notes, queries, weights=None):
        corpus = bm.build_field_weighted_bm25_corpus(notes, weights)
        cases.append({"name": name, "notes": notes, "queries": queries, "field_weights": weights,
                      "raw": [corpus.get_scores(bm.tokenize_text(q)) for q in queries if bm.tokenize_text(q)],
                      "expected": [bm.field_weighted_bm25_scores(corpus, notes, q) for q in queries]})

    for t in range(4):
        nn = []
        for i in range(rng.randint(5, 50)):
            nn.append({"title": " ".join(rng.choices(vocab, k=rng.randint(0, 4))),
                       "entities": rng.choices(vocab, k=rng.randint(0, 3)) if rng.random() < 0.8 else "w1 w2",
                       "content": " ".join(rng.choices(vocab, k=rng.randint(0, 30)))})
        qs = [" ".join(rng.choices(vocab, k=rng.randint(1, 6))) for _ in range(5)] + ["", "w3 w3 w4"]
        add(f"random_{t}", nn, qs, None if t % 2 == 0 else {"content": 1.0, "title": 3.0})
    add("no_entities_anywhere", [{"title": "a b", "content": "c d a"}, {"title": "", "content": "a"}], ["a", "d q"])
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
    with open(os.path.join(HERE, "bm25_field_cases.json"), "w") as f:
        json.dump({"source": "reference utils/bm25_search.py FieldWeightedBM25 / field_weighted_bm25_scores",
                   "cases": bm25_field_cases(bm)}, f)
    with open(os.path.join(HERE, "fusion_long_cases.json"), "w") as f:
        json.dump({"source": "reference retrieval/hybrid_search.py HybridSearcher.fuse, full-corpus bm25 lists",
                   "cases": fusion_long_cases(hs)}, f)
    with open(os.path.join(HERE, "fusion_rrf_multi_long_cases.json"), "w") as f:
        json.dump({"source": "reference retrieval/hybrid_search.py HybridSearcher.fuse (rrf), two or three full-corpus lists",
                   "cases": fusion_rrf_multi_long_cases(hs)}, f)
    with open(os.path.join(HERE, "candidate_fusion_cases.json"), "w") as f:
        json.dump({"source": "reference query/query_processor.py QueryProcessor._hybrid_search / _enhanced_hybrid_search_v2 "
                             "(method bodies compiled from the reference source as they stand)",
                   "cases": candidate_fusion_cases()}, f)
    sys.modules.pop("utils.bm25_search", None)
    sys.modules.pop("utils", None)
    stub_utils_and_config()
    gr = load_graph_retriever()
    with open(os.path.join(HERE, "embedding_candidates_cases.json"), "w") as f:
        json.dump({"source": "reference graph/graph_retriever.py GraphRetriever._find_embedding_candidates",
                   "cases": embedding_candidate_cases(gr)}, f)
    rx = load_by_path("ref_relation_extractor", os.path.join(REF, "graph", "relation_extractor.py"))
    with open(os.path.join(HERE, "similarity_relation_cases.json"), "w") as f:
        json.dump({"source": "reference graph/relation_extractor.py extract_semantic_similarity_relations",
                   "cases": similarity_cases(rx)}, f)
    print("wrote candidate_fusion_cases.json, fusion_cases.json, fusion_long_cases.json, bm25_cases.json, embedding_candidates_cases.json, "
          "similarity_relation_cases.json")


if __name__ == "__main__":
    main()
