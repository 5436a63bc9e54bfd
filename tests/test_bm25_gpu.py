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
