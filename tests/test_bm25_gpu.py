"""GPU parity: device BM25 (anr_bm25_*) against the golden vectors produced by the reference's own
utils/bm25_search.py and against the oracle on a larger random corpus.  Bar: bit-exact float64."""
import json
import os

import numpy as np
import pytest

from oracle import bm25 as obm

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases():
    with open(os.path.join(GOLD, "bm25_cases.json")) as f:
        return json.load(f)["cases"]


def _text(n):
    return f"{n.get('title', '')} {n.get('content', '')}"


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_bm25_scores_match_reference_golden(case):
    from anorag_hip import bm25_search as dbm
    corpus = dbm.build_bm25_corpus(case["notes"], _text)
    for q, exp, toks in zip(case["queries"], case["expected"], case["tokens"]):
        assert dbm.tokenize_text(q) == toks
        assert dbm.bm25_scores(corpus, case["notes"], q) == exp
    corpus.close()


def test_bm25_random_corpus_vs_oracle_and_sparse_form():
    from anorag_hip import bm25_search as dbm
    rng = np.random.default_rng(5)
    vocab = [f"w{i}" for i in range(3000)]
    probs = 1.0 / np.arange(1, 3001)
    probs /= probs.sum()
    notes = [{"title": "", "content": " ".join(rng.choice(vocab, size=rng.integers(0, 60), p=probs))} for _ in range(20000)]
    queries = [" ".join(rng.choice(vocab, size=rng.integers(1, 8), p=probs)) for _ in range(40)] + ["w1 w1 w2", "zzz"]
    dev = dbm.build_bm25_corpus(notes, _text)
    ref = obm.build_bm25_corpus(notes, _text)
    toks = [dbm.tokenize_text(q) for q in queries]
    raw = dev.scores_batch(toks, normalize=False)
    norm = dev.scores_batch(toks, normalize=True)
    for i, q in enumerate(queries):
        exp_raw = ref.get_scores(obm.tokenize_text(q))
        assert raw[i].tolist() == exp_raw                      # bit-exact, incl. repeated query tokens
        assert norm[i].tolist() == obm.bm25_scores(ref, notes, q)
    sparse = dev.nonzero_batch(toks, normalize=True, cap=20000)
    for i in range(len(queries)):
        ids, sc = sparse[i]
        nz = np.nonzero(norm[i])[0]
        assert set(ids.tolist()) == set(nz.tolist())
        assert np.array_equal(sc, norm[i][ids]) and np.all(np.diff(sc) <= 0)
    assert dev.get_scores(["zzz"]) == [0.0] * len(notes)
    dev.close()


def _field_cases():
    with open(os.path.join(GOLD, "bm25_field_cases.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _field_cases(), ids=lambda c: c["name"])
def test_field_weighted_bm25_matches_reference_golden(case):
    """FieldWeightedBM25 (utils/bm25_search.py:66-234): per-field device scoring + the device combine, bit-exact"""
    from anorag_hip import bm25_search as dbm
    corpus = dbm.build_field_weighted_bm25_corpus(case["notes"], case["field_weights"])
    raw = iter(case["raw"])
    for q, exp in zip(case["queries"], case["expected"]):
        assert dbm.field_weighted_bm25_scores(corpus, case["notes"], q) == exp
        if dbm.tokenize_text(q):
            assert corpus.get_scores(dbm.tokenize_text(q)) == next(raw)
    corpus.close()


def test_field_weighted_bm25_feeds_the_fusion_without_leaving_the_device():
    """DeviceFieldWeightedBM25.scores_device -> HybridSearcher.fuse_arrays: the N-vector stays on the device"""
    from anorag_hip import bm25_search as dbm
    from oracle import fusion as ofu
    from retrieval.hybrid_search import HybridSearcher
    rng = np.random.default_rng(8)
    vocab = [f"w{i}" for i in range(400)]
    notes = [{"title": " ".join(rng.choice(vocab, 3)), "entities": list(rng.choice(vocab, 2)),
              "content": " ".join(rng.choice(vocab, size=rng.integers(5, 40)))} for _ in range(9000)]
    queries = [" ".join(rng.choice(vocab, size=4)) for _ in range(6)]
    dev = dbm.build_field_weighted_bm25_corpus(notes)
    ref = obm.build_field_weighted_bm25_corpus(notes)
    arr = dev.scores_device([dbm.tokenize_text(q) for q in queries], normalize=True)
    host = arr.numpy()
    dense = [(rng.choice(9000, 50, replace=False).astype(np.int64), np.sort(rng.uniform(0.3, 0.9, 50))[::-1].copy())
             for _ in queries]
    w = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
    hs = HybridSearcher({"retrieval": {"candidate_pool": 40, "hybrid": {"fusion_method": "rrf", "rrf_k": 60, "weights": w}}})
    got = hs.fuse_arrays(len(queries), dense=dense, bm25=arr)
    arr.free()
    full = np.arange(9000, dtype=np.int64)
    for i, q in enumerate(queries):
        exp_scores = obm.field_weighted_bm25_scores(ref, notes, q)
        assert host[i].tolist() == exp_scores
        ids, fin = ofu.fuse_arrays(9000, (dense[i], (full, np.asarray(exp_scores)), None, None), [1.0, 0.5, 0.5, 0.1], "rrf", 60, 40)
        assert [r["note_id"] for r in got[i]] == ids.tolist() and [r["final_similarity"] for r in got[i]] == fin.tolist()
    dev.close()


def test_device_scores_carry_their_row_maxima_and_linear_fusion_uses_them():
    """scores_device leaves each row's maximum beside the N-vector (a by-product of the scoring pass); the linear
    fusion takes it instead of a max pass of its own — results identical to computing it, and to the oracle"""
    from anorag_hip import bm25_search as dbm
    from anorag_hip.fusion import fuse_dense
    from oracle import fusion as ofu
    rng = np.random.default_rng(21)
    vocab = [f"w{i}" for i in range(800)]
    notes = [{"title": " ".join(rng.choice(vocab, 2)), "entities": list(rng.choice(vocab, 2)),
              "content": " ".join(rng.choice(vocab, size=rng.integers(0, 30)))} for _ in range(12000)]
    queries = [" ".join(rng.choice(vocab, size=rng.integers(1, 6))) for _ in range(7)] + ["zzz qqq"]  # last: no hit at all
    toks = [dbm.tokenize_text(q) for q in queries]
    nq = len(queries)
    dense = [(rng.choice(12000, 60, replace=False).astype(np.int64), np.sort(rng.uniform(0.3, 0.9, 60))[::-1].copy())
             for _ in queries]
    w = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
    full = np.arange(12000, dtype=np.int64)
    for corpus in (dbm.build_bm25_corpus(notes, _text), dbm.build_field_weighted_bm25_corpus(notes)):
        for normalize in (False, True):
            arr = corpus.scores_device(toks, normalize=normalize)
            host = arr.numpy()
            assert arr.row_max is not None
            assert arr.row_max.numpy().reshape(-1).tolist() == host.max(axis=1).tolist()
            with_max = fuse_dense("linear", w, 60.0, 50, nq, {"dense": dense, "bm25": arr})
            kept, arr.row_max = arr.row_max, None
            without = fuse_dense("linear", w, 60.0, 50, nq, {"dense": dense, "bm25": arr})
            arr.row_max = kept
            for a, b in zip(with_max, without):
                assert np.array_equal(a, b, equal_nan=True)
            for i in range(nq):
                ids, fin = ofu.fuse_arrays(12000, (dense[i], (full, host[i]), None, None), [1.0, 0.5, 0.5, 0.1], "linear", 60, 50)
                assert with_max[1][i, :len(fin)].tolist() == fin.tolist()
            arr.free()
        corpus.close()
