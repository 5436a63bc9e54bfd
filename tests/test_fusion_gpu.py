"""GPU parity: the device fusion (HybridSearcher.fuse drop-in, anr_fuse_lists) against the golden vectors
produced by the reference file itself and against the oracle on larger random inputs.  Bar: bit-exact
final_similarity (float64), identical ids and order (modulo the reference's hash-order ties in `linear`)."""
import json
import os

import numpy as np
import pytest

from oracle import fusion as ofu

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases():
    with open(os.path.join(GOLD, "fusion_cases.json")) as f:
        return json.load(f)["cases"]


def _tup(lst):
    return None if lst is None else [tuple(x) for x in lst]


def _compare(got, exp, pool, method):
    assert len(got) == len(exp)
    assert [r["final_similarity"] for r in got] == [r["final_similarity"] for r in exp]
    if method == "rrf":
        assert [r["note_id"] for r in got] == [r["note_id"] for r in exp]
    else:
        groups = {}
        for r in exp:
            groups.setdefault(r["final_similarity"], set()).add(r["note_id"])
        ggot = {}
        for r in got:
            ggot.setdefault(r["final_similarity"], set()).add(r["note_id"])
        last = exp[-1]["final_similarity"] if exp else None
        for sc, ids in groups.items():
            if sc == last and len(exp) == pool:
                continue
            assert ggot[sc] == ids
    by_id = {r["note_id"]: r for r in exp}
    for r in got:
        if r["note_id"] in by_id:
            assert r["scores"] == by_id[r["note_id"]]["scores"]
            assert r["tags"] == by_id[r["note_id"]]["tags"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_fuse_matches_reference_golden(case):
    from retrieval.hybrid_search import HybridSearcher, create_hybrid_searcher
    hs = create_hybrid_searcher(case["config"])
    assert isinstance(hs, HybridSearcher)
    got = hs.fuse(_tup(case["dense"]), _tup(case["bm25"]), _tup(case["graph"]), _tup(case["path"]))
    _compare(got, case["expected"], hs.candidate_pool, hs.fusion_method)


@pytest.mark.parametrize("method", ["linear", "rrf"])
def test_fuse_batch_random_vs_oracle(method):
    from retrieval.hybrid_search import HybridSearcher
    rng = np.random.default_rng(99)
    cfg = {"retrieval": {"candidate_pool": 80, "hybrid": {"enabled": True, "fusion_method": method, "rrf_k": 60,
                                                         "weights": {"dense": 1.0, "bm25": 0.5, "graph": 0.5,
                                                                     "path": 0.1}}}}
    hs = HybridSearcher(cfg)
    queries = []
    for _ in range(32):
        lists = []
        for m in (100, 1500, 60, 12):
            ids = rng.choice(1_000_000, size=m, replace=False)
            sc = np.abs(rng.standard_normal(m))
            lists.append([(f"n{int(i)}", float(s)) for i, s in zip(ids, sc)])
        queries.append(tuple(lists))
    got = hs.fuse_batch(queries)
    for lists, g in zip(queries, got):
        exp = ofu.fuse(*lists, candidate_pool=80, fusion_method=method, weights=hs.weights, rrf_k=60)
        _compare(g, exp, 80, method)


def test_fuse_rejects_oversized_lists():
    from anorag_hip import AnoragError
    from retrieval.hybrid_search import HybridSearcher
    hs = HybridSearcher({"retrieval": {"hybrid": {"weights": {"dense": 1.0}}}})
    big = [(i, 1.0) for i in range(5000)]
    with pytest.raises(AnoragError):
        hs.fuse(dense=big)
