"""CPU tests: the oracle restatements against the golden vectors produced by the reference's own files
(tests/golden/make_golden.py), plus self-consistency of the flat-index oracle."""
import json
import os

import numpy as np
import pytest

from oracle import bm25 as obm
from oracle import flat_index as orc
from oracle import fusion as ofu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)["cases"]


def _tup(lst):
    return None if lst is None else [tuple(x) for x in lst]


def _groups(results):
    """final-score -> set of ids, so hash-order among exactly equal finals does not matter"""
    g = {}
    for r in results:
        g.setdefault(r["final_similarity"], set()).add(r["note_id"])
    return g


@pytest.mark.parametrize("case", _load("fusion_cases.json"), ids=lambda c: c["name"])
def test_fusion_oracle_matches_reference(case):
    h = case["config"]["retrieval"]["hybrid"]
    got = ofu.fuse(_tup(case["dense"]), _tup(case["bm25"]), _tup(case["graph"]), _tup(case["path"]),
                   candidate_pool=case["config"]["retrieval"]["candidate_pool"], enabled=h["enabled"],
                   fusion_method=h["fusion_method"], weights=h["weights"], rrf_k=h["rrf_k"])
    exp = case["expected"]
    assert len(got) == len(exp)
    # bit-exact finals (same float64 arithmetic in the same order), position by position
    assert [r["final_similarity"] for r in got] == [r["final_similarity"] for r in exp]
    gg, ge = _groups(got), _groups(exp)
    last = exp[-1]["final_similarity"] if exp else None
    for score, ids in ge.items():
        if score == last and len(exp) == case["config"]["retrieval"]["candidate_pool"]:
            continue  # a tie group cut by the pool: membership is hash-order dependent in the reference
        assert gg[score] == ids
    by_id = {r["note_id"]: r for r in got}
    for r in exp:
        if r["note_id"] in by_id:
            assert by_id[r["note_id"]]["scores"] == r["scores"]
            assert by_id[r["note_id"]]["tags"] == r["tags"]


def test_fusion_array_form_matches_dict_form():
    rng = np.random.default_rng(5)
    n = 500
    for method in ("linear", "rrf"):
        for _ in range(5):
            lists = []
            for m in (80, 200, 30, 10):
                ids = rng.choice(n, size=m, replace=False).astype(np.int64)
                sc = rng.random(m)
                lists.append((ids, sc))
            w = [1.0, 0.5, 0.35, 0.1]
            sel, fin = ofu.fuse_arrays(n, lists, w, method, 60, 80)
            d = ofu.fuse(*[[(int(i), float(s)) for i, s in zip(*l)] for l in lists], candidate_pool=80,
                         fusion_method=method, weights=dict(zip(("dense", "bm25", "graph", "path"), w)), rrf_k=60)
            assert np.array_equal(np.array([r["final_similarity"] for r in d]), fin)
            if method == "rrf":   # the reference's tie order is deterministic there
                assert [r["note_id"] for r in d] == sel.tolist()
            else:
                assert {r["note_id"] for r in d} == set(sel.tolist())


@pytest.mark.parametrize("case", _load("bm25_cases.json"), ids=lambda c: c["name"])
def test_bm25_oracle_matches_reference(case):
    corpus = obm.build_bm25_corpus(case["notes"], lambda n: f"{n.get('title', '')} {n.get('content', '')}")
    for q, exp, toks in zip(case["queries"], case["expected"], case["tokens"]):
        assert obm.tokenize_text(q) == toks
        assert obm.bm25_scores(corpus, case["notes"], q) == exp


def test_flat_oracle_basics():
    x = np.random.default_rng(0).standard_normal((300, 24), dtype=np.float32)
    x[7] = 0
    xn = orc.preprocess_vectors(x)
    assert xn.dtype == np.float32 and np.all(xn[7] == 0)
    assert np.allclose(np.linalg.norm(np.delete(xn, 7, 0), axis=1), 1, atol=1e-6)
    q = orc.preprocess_vectors(np.random.default_rng(1).standard_normal((3, 24), dtype=np.float32))
    D, I = orc.flat_search(q, xn, 5, "ip")
    brute = np.argsort(-(q.astype(np.float64) @ xn.astype(np.float64).T), axis=1, kind="stable")[:, :5]
    assert np.array_equal(I, brute)
    D2, I2 = orc.flat_search(q, xn[:3], 5, "ip")
    assert np.all(I2[:, 3:] == -1) and np.all(D2[:, 3:] == -orc.FLT_MAX)
    Dl, Il = orc.flat_search(q, xn, 4, "l2")
    d2 = ((q[:, None, :].astype(np.float64) - xn[None].astype(np.float64)) ** 2).sum(-1)
    assert np.array_equal(Il, np.argsort(d2, axis=1, kind="stable")[:, :4])
    # result shaping (vector_index.py:226-259): single query -> flat list, -1 dropped, L2 similarity
    r = orc.shape_results(D2[:1], I2[:1], "cosine")
    assert isinstance(r[0], dict) and len(r) == 3 and set(r[0]) == {"index", "score", "rank", "similarity"}
    r = orc.shape_results(Dl, Il, "l2")
    assert len(r) == 3 and abs(r[0][0]["similarity"] - 1.0 / (1.0 + r[0][0]["score"])) < 1e-12


def test_similarity_relations_oracle_matches_reference_golden():
    """oracle/graph_scans.py vs the relations the reference's own relation_extractor.py produced
    (tests/golden/similarity_relation_cases.json): same pairs in the same order, same similarity, weight, rank."""
    import json, os
    import numpy as np
    from oracle import graph_scans as og
    data = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "similarity_relation_cases.json")))
    assert len(data["cases"]) >= 5
    for case in data["cases"]:
        emb = np.asarray(case["embeddings"], dtype=np.float32)
        notes = [{"note_id": f"n{i:04d}"} for i in range(emb.shape[0])]
        got = og.semantic_similarity_relations(notes, emb, case["threshold"], case["weight"])
        exp = case["expected"]
        assert [(r["source_id"], r["target_id"]) for r in got] == [(r["source_id"], r["target_id"]) for r in exp], case["name"]
        for g, e in zip(got, exp):
            assert g["metadata"]["cosine_similarity"] == e["cosine_similarity"], case["name"]
            assert float(g["weight"]) == e["weight"], case["name"]
            assert g["metadata"]["similarity_rank"] == e["similarity_rank"], case["name"]


def _unpack_vec(v):
    a = np.full(v["n"], v["fill"], dtype=np.float64)
    a[v["idx"]] = v["val"]
    return a


@pytest.mark.parametrize("case", _load("fusion_long_cases.json"), ids=lambda c: c["name"])
def test_fusion_oracle_matches_reference_on_full_corpus_lists(case):
    """the N-entry bm25 lists (one entry per note): both oracle forms against the reference's own output"""
    h = case["config"]["retrieval"]["hybrid"]
    pool, n = case["config"]["retrieval"]["candidate_pool"], case["n"]
    bm = _unpack_vec(case["bm25_vec"]) if case["bm25_vec"] else None
    dv = _unpack_vec(case["dense_vec"]) if case["dense_vec"] else None
    dense = [(i, float(dv[i])) for i in range(n)] if dv is not None else _tup(case["dense"])
    bm25 = [(i, float(bm[i])) for i in range(n)] if bm is not None else None
    got = ofu.fuse(dense, bm25, _tup(case["graph"]), _tup(case["path"]), candidate_pool=pool,
                   fusion_method=h["fusion_method"], weights=h["weights"], rrf_k=h["rrf_k"])
    exp = case["expected"]
    assert [r["final_similarity"] for r in got] == [r["final_similarity"] for r in exp]
    if h["fusion_method"] == "rrf":
        assert [r["note_id"] for r in got] == [r["note_id"] for r in exp]

    def arr(lst):
        return (np.array([p[0] for p in lst], dtype=np.int64), np.array([p[1] for p in lst], dtype=np.float64))
    ids, fin = ofu.fuse_arrays(n, (arr(dense), arr(bm25) if bm25 else None, arr(case["graph"]), arr(case["path"])),
                               [h["weights"].get(k, 0.0) for k in ("dense", "bm25", "graph", "path")],
                               h["fusion_method"], h["rrf_k"], pool)
    assert fin.tolist() == [r["final_similarity"] for r in exp]
    if h["fusion_method"] == "rrf":
        assert ids.tolist() == [r["note_id"] for r in exp]


@pytest.mark.parametrize("case", _load("embedding_candidates_cases.json"), ids=lambda c: c["name"])
def test_embedding_candidates_oracle_matches_reference_golden(case):
    from oracle import graph_scans as og
    emb = np.asarray(case["embeddings"], dtype=np.float32)
    q = np.asarray(case["query"], dtype=np.float32)
    got = og.find_embedding_candidates(emb, q, case["top_k"])
    assert [f"note_{int(i):04d}" for i in got] == case["expected"]


@pytest.mark.parametrize("case", _load("bm25_field_cases.json"), ids=lambda c: c["name"])
def test_field_weighted_bm25_oracle_matches_reference(case):
    corpus = obm.build_field_weighted_bm25_corpus(case["notes"], case["field_weights"])
    raw = iter(case["raw"])
    for q, exp in zip(case["queries"], case["expected"]):
        assert obm.field_weighted_bm25_scores(corpus, case["notes"], q) == exp
        if obm.tokenize_text(q):
            assert corpus.get_scores(obm.tokenize_text(q)) == next(raw)
