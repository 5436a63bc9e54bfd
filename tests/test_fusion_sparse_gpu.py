"""GPU parity of the SPARSE hand-off BM25 -> fusion (anr_bm25_sparse_dev -> anr_fuse_source.sparse_*): a BM25 row over N
notes given by the few thousand documents its postings touch instead of as an N-vector.

Pinned three ways: the reference-run golden cases of the N-array fusion whose bm25 vector is zero-filled
(tests/golden/fusion_long_cases.json, produced by the reference's own HybridSearcher.fuse), equality with the dense-array
form of the same rows on adversarial random rows (explicit zeros, negatives, an absent id, short lists overlapping the
entries and the implicit zeros), and the BM25 producer against anr_bm25_scores.  Bar: bit-exact float64, identical ids."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases():
    with open(os.path.join(GOLD, "fusion_long_cases.json")) as f:
        return [c for c in json.load(f)["cases"]
                if c["bm25_vec"] is not None and c["dense_vec"] is None and c["bm25_vec"]["fill"] == 0.0]


def _pairs(lst):
    if lst is None:
        return None
    return (np.array([p[0] for p in lst], dtype=np.int64), np.array([p[1] for p in lst], dtype=np.float64))


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_sparse_bm25_rows_match_reference_golden(case):
    from anorag_hip.fusion import SparseRows
    from retrieval.hybrid_search import HybridSearcher
    from test_fusion_dense_gpu import _compare
    hs = HybridSearcher(case["config"])
    v = case["bm25_vec"]
    rows = SparseRows.from_numpy([(np.asarray(v["idx"], dtype=np.int64), np.asarray(v["val"], dtype=np.float64))], v["n"])
    got = hs.fuse_arrays(1, dense=[_pairs(case["dense"])], bm25=rows, graph=[_pairs(case["graph"])],
                         path=[_pairs(case["path"])])[0]
    rows.free()
    _compare(got, case["expected"], hs.candidate_pool, hs.fusion_method)


def _random_rows(rng, nq, n, nnz_max, kinds):
    """rows with the entries the stage kernel treats differently: positives, explicit zeros, negatives, ties, a NaN"""
    rows = []
    for q in range(nq):
        nnz = int(rng.integers(0, min(nnz_max, n) + 1)) if q else min(nnz_max, n)
        ids = rng.choice(n, size=nnz, replace=False)
        val = np.abs(rng.standard_normal(nnz))
        if "ties" in kinds and nnz:
            val = np.round(val * 4) / 4                     # many equal values (and some exact zeros)
        if "zeros" in kinds and nnz:
            val[rng.random(nnz) < 0.1] = 0.0
            val[rng.random(nnz) < 0.02] = -0.0
        if "neg" in kinds and nnz:
            neg = rng.random(nnz) < 0.2
            val[neg] = -val[neg]
        if "nan" in kinds and nnz > 3:
            val[rng.integers(0, nnz)] = np.nan
        rows.append((ids.astype(np.int64), val))
    return rows


def _dense_of(rows, n):
    a = np.zeros((len(rows), n), dtype=np.float64)
    for i, (ids, val) in enumerate(rows):
        a[i, ids] = val
    return a


def _short_lists(rng, rows, n, m, nq):
    """per query a short list that holds listed ids (zero-valued ones too), unlisted low ids (implicit zeros that are
    candidates) and unlisted high ids"""
    out = []
    for q in range(nq):
        ids, val = rows[q]
        pick = []
        if len(ids):
            pick += rng.choice(ids, size=min(len(ids), m // 3), replace=False).tolist()
            z = ids[val == 0.0]
            pick += z[: m // 6].tolist()
        pick += rng.integers(0, min(n, 64), size=m // 4).tolist()
        pick += rng.integers(0, n, size=m // 4).tolist()
        pick = list(dict.fromkeys(int(x) for x in pick))[:m]
        rng.shuffle(pick)
        sc = np.sort(rng.standard_normal(len(pick)))[::-1].copy()
        out.append((np.asarray(pick, dtype=np.int64), sc))
    return out


@pytest.mark.parametrize("method", ["linear", "rrf"])
@pytest.mark.parametrize("n,nnz_max,pool,m,kinds", [
    (1_000_000, 3000, 80, 100, ("zeros",)),            # the C5 shape
    (50_000, 6000, 50, 300, ("zeros", "neg", "ties")),
    (5000, 5000, 200, 600, ("zeros", "neg", "nan")),   # rows that list every id
    (300, 300, 1024, 40, ("ties", "neg")),             # pool beyond N
    (7, 5, 10, 6, ("zeros",)),
    (1, 1, 5, 1, ()),
    (4096, 0, 30, 50, ()),                             # rows without a single entry
    (200_000, 8192, 100, 1000, ("zeros", "neg", "ties", "nan")),   # the largest row the fusion sorts itself, the longest short lists
    (1_000_000, 20_000, 80, 100, ("zeros",)),          # rows beyond the LDS sort (handed over sorted by id): a frequent-word query
    (70_000, 65_536, 100, 400, ("zeros", "neg", "ties", "nan")),   # the largest row there is
    (9000, 9000, 300, 900, ("zeros", "neg")),          # ... and one that lists every id
])
def test_sparse_rows_fuse_like_the_dense_array(method, n, nnz_max, pool, m, kinds):
    from anorag_hip.fusion import DeviceArray, SparseRows, fuse_dense
    rng = np.random.default_rng(n + pool + (1 if method == "rrf" else 0))
    nq = 5
    rows = _random_rows(rng, nq, n, nnz_max, kinds)
    dense = _short_lists(rng, rows, n, m, nq)
    graph = _short_lists(rng, rows, n, max(m // 8, 1), nq)
    w = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
    arr = DeviceArray.from_numpy(_dense_of(rows, n))
    sp = SparseRows.from_numpy(rows, n, cap=max(nnz_max, 1))
    for src in ({"dense": dense, "bm25": None, "graph": graph}, {"dense": dense, "bm25": None}, {"bm25": None}):
        a = fuse_dense(method, w, 60.0, pool, nq, {**src, "bm25": arr})
        b = fuse_dense(method, w, 60.0, pool, nq, {**src, "bm25": sp})
        assert np.array_equal(a[3], b[3])
        assert np.array_equal(a[0], b[0])
        assert a[1].tobytes() == b[1].tobytes()
        assert np.array_equal(a[2], b[2], equal_nan=True)
    arr.free()
    sp.free()


def test_sparse_source_in_the_dense_and_graph_slots_and_argument_errors():
    from anorag_hip import _lib
    from anorag_hip.fusion import DeviceArray, SparseRows, fuse_dense
    rng = np.random.default_rng(3)
    n, nq = 20_000, 3
    rows = _random_rows(rng, nq, n, 500, ("zeros", "neg"))
    lists = _short_lists(rng, rows, n, 60, nq)
    w = {"dense": 0.7, "bm25": 0.4, "graph": 0.9, "path": 0.2}
    arr = DeviceArray.from_numpy(_dense_of(rows, n))
    sp = SparseRows.from_numpy(rows, n)
    for slot in ("dense", "graph"):
        for method in ("linear", "rrf"):
            other = "bm25"
            a = fuse_dense(method, w, 10.0, 40, nq, {slot: arr, other: lists})
            b = fuse_dense(method, w, 10.0, 40, nq, {slot: sp, other: lists})
            assert np.array_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes()
            assert np.array_equal(a[2], b[2], equal_nan=True) and np.array_equal(a[3], b[3])
    with pytest.raises(_lib.AnoragError):      # a sparse source beside a dense array
        fuse_dense("linear", w, 10.0, 40, nq, {"dense": arr, "bm25": sp})
    with pytest.raises(_lib.AnoragError):      # the path source is a list
        fuse_dense("linear", w, 10.0, 40, nq, {"path": sp})
    sp2 = SparseRows.from_numpy(rows, n)
    with pytest.raises(_lib.AnoragError):      # two sparse sources
        fuse_dense("linear", w, 10.0, 40, nq, {"dense": sp2, "bm25": sp})
    for x in (arr, sp, sp2):
        x.free()


def _corpus(rng, n_docs, n_vocab, max_len):
    """Zipf-distributed words, 0 .. max_len - 1 per note (all tokens drawn at once: a draw per note took minutes at 150 k notes)"""
    vocab = [f"w{i}" for i in range(n_vocab)]
    probs = 1.0 / np.arange(1, n_vocab + 1)
    probs /= probs.sum()
    lens = rng.integers(0, max_len, size=n_docs)
    toks = rng.choice(n_vocab, size=int(lens.sum()), p=probs)
    words = np.array(vocab, dtype=object)
    offs = np.concatenate([[0], np.cumsum(lens)])
    notes = [{"title": "", "content": " ".join(words[toks[offs[i]:offs[i + 1]]])} for i in range(n_docs)]
    return vocab, probs, notes


def test_bm25_sparse_rows_equal_the_dense_scores_bit_for_bit():
    from anorag_hip import bm25_search as dbm
    rng = np.random.default_rng(11)
    vocab, probs, notes = _corpus(rng, 30_000, 20_000, 30)
    dev = dbm.build_bm25_corpus(notes, lambda n: f"{n.get('title', '')} {n.get('content', '')}")
    # rare words only (ranks >= 200): every query touches fewer than 6144 documents
    queries = [" ".join(rng.choice(vocab[200:], size=rng.integers(1, 9))) for _ in range(60)]
    queries += [f"{vocab[300]} {vocab[300]} {vocab[301]}", "zzz", ""]
    toks = [dbm.tokenize_text(q) for q in queries]
    for normalize in (True, False):
        full = dev.scores_batch(toks, normalize=normalize)
        sp = dev.scores_sparse_device(toks, normalize=normalize)
        assert sp is not None
        got = sp.numpy()
        for i in range(len(queries)):
            ids, val = got[i]
            assert len(set(ids.tolist())) == len(ids) == int(sp.counts[i])
            assert set(np.nonzero(full[i])[0].tolist()) <= set(ids.tolist())
            assert val.tobytes() == full[i][ids].tobytes()
            mask = np.ones(full.shape[1], dtype=bool)
            mask[ids] = False
            assert not full[i][mask].any()
        sp.free()
    # a frequent word touches more documents than ONE table holds: refused at that capacity, held by the sliced rows
    assert dev.scores_sparse_device([[vocab[0]]] + toks[:3], cap=dev.SPARSE_CAP) is None
    sp = dev.scores_sparse_device([[vocab[0]]] + toks[:3])
    assert sp is not None and sp.cap > dev.SPARSE_CAP and int(sp.counts[0]) > dev.SPARSE_CAP
    sp.free()
    dev.close()


def test_bm25_sliced_sparse_rows_equal_the_dense_scores_bit_for_bit():
    """rows of up to 65 536 documents (anr_bm25_sparse_dev beyond one LDS table: document-range slices, rank = slot):
    every entry equals anr_bm25_scores' bit for bit, every other document is 0.0, ids ascending"""
    from anorag_hip import bm25_search as dbm
    rng = np.random.default_rng(21)
    vocab, probs, notes = _corpus(rng, 150_000, 20_000, 14)     # more documents than one slice may span (131 072 ids)
    dev = dbm.build_bm25_corpus(notes, lambda n: f"{n.get('title', '')} {n.get('content', '')}")
    queries = [" ".join(rng.choice(vocab[4:300], size=rng.integers(1, 7), p=probs[4:300] / probs[4:300].sum())) for _ in range(20)]
    queries += [f"{vocab[3]} {vocab[3]} {vocab[4]} {vocab[900]}",   # a token twice
                " ".join(vocab[5:45]),                              # forty frequent words
                vocab[1], vocab[15000], "zzz", ""]
    toks = [dbm.tokenize_text(q) for q in queries]
    heavy = [dbm.tokenize_text(f"{vocab[0]} {vocab[1]}")]            # more than half the corpus: more than any row holds
    for normalize in (True, False):
        full = dev.scores_batch(toks, normalize=normalize)
        sp = dev.scores_sparse_device(toks, normalize=normalize, allow_overflow=True)
        assert sp.cap > dev.SPARSE_CAP
        lib_rows = sp.numpy()
        ids_raw = np.empty((sp.nq, sp.cap), dtype=np.uint32)
        from anorag_hip import _lib
        _lib.check(_lib.load().anr_device_copy(sp.device, ids_raw.ctypes.data, sp.ids_ptr, ids_raw.nbytes, 1), "anr_device_copy")
        n_big = 0
        for i in range(len(queries)):
            nz = np.nonzero(full[i])[0]
            if sp.counts[i] < 0:                                    # given up: it must really not fit
                assert len(nz) > sp.cap, (i, len(nz), sp.cap)
                continue
            ids, val = lib_rows[i]
            k = int(sp.counts[i])
            assert len(ids) == k and np.all(np.diff(ids_raw[i, :k].astype(np.int64)) > 0)     # ascending on the device
            assert set(nz.tolist()) <= set(ids.tolist())
            assert val.tobytes() == full[i][ids].tobytes()
            mask = np.ones(full.shape[1], dtype=bool)
            mask[ids] = False
            assert not full[i][mask].any()
            n_big += k > dev.SPARSE_CAP
        assert n_big >= 8 and (sp.counts >= 0).sum() >= len(queries) // 2, (n_big, sp.counts.tolist())
        sp.free()
    rows = dev.scores_sparse_device(heavy + toks[:2], allow_overflow=True)
    assert rows.cap == dev.SPARSE_CAP_MAX and rows.counts[0] == -1 and (rows.counts[1:] > 0).all()
    rows.free()
    # a capacity between the two forms that the query does not fit / fits
    one = [dbm.tokenize_text(vocab[40])]
    need = int(np.count_nonzero(dev.scores_batch(one, normalize=False)[0]))
    assert need > 100
    r = dev.scores_sparse_device(one, cap=7000, allow_overflow=True)
    assert r.counts.tolist() == ([need] if need <= 7000 else [-1])
    r.free()
    dev.close()


@pytest.mark.parametrize("method", ["linear", "rrf"])
def test_bm25_to_fusion_without_the_n_vector(method):
    """DeviceBM25.scores_sparse_device -> HybridSearcher.fuse_arrays == the same through scores_device"""
    from anorag_hip import bm25_search as dbm
    from retrieval.hybrid_search import HybridSearcher
    rng = np.random.default_rng(12)
    vocab, probs, notes = _corpus(rng, 40_000, 30_000, 25)
    dev = dbm.build_bm25_corpus(notes, lambda n: f"{n.get('title', '')} {n.get('content', '')}")
    queries = [" ".join(rng.choice(vocab[300:], size=rng.integers(1, 7))) for _ in range(24)] + ["zzz"]
    toks = [dbm.tokenize_text(q) for q in queries]
    nq = len(queries)
    dense = []
    for _ in range(nq):
        ids = rng.choice(len(notes), size=100, replace=False).astype(np.int64)
        dense.append((ids, np.sort(rng.random(100))[::-1].copy()))
    hs = HybridSearcher({"retrieval": {"candidate_pool": 80, "hybrid": {
        "enabled": True, "fusion_method": method, "rrf_k": 60,
        "weights": {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}}}})
    full = dev.scores_device(toks, normalize=True)
    sp = dev.scores_sparse_device(toks, normalize=True)
    assert sp is not None
    a = hs.fuse_arrays(nq, dense=dense, bm25=full)
    b = hs.fuse_arrays(nq, dense=dense, bm25=sp)
    assert a == b
    assert any(r["scores"]["bm25"] for res in b for r in res)
    full.free()
    sp.free()
    dev.close()


@pytest.mark.parametrize("method", ["linear", "rrf"])
def test_fuse_bm25_sends_heavy_queries_down_the_vector_path(method):
    """HybridSearcher.fuse_bm25: queries under the row capacity go sparse, a frequent word's query takes the N-vector;
    every result equals the all-vector pipeline's"""
    from anorag_hip import bm25_search as dbm
    from retrieval.hybrid_search import HybridSearcher
    rng = np.random.default_rng(13)
    vocab, probs, notes = _corpus(rng, 40_000, 5_000, 25)
    dev = dbm.build_bm25_corpus(notes, lambda n: f"{n.get('title', '')} {n.get('content', '')}")
    queries = [" ".join(rng.choice(vocab[100:], size=rng.integers(1, 5))) for _ in range(10)]
    queries[2] = f"{vocab[0]} {vocab[700]}"       # the most frequent word: tens of thousands of documents
    queries[7] = f"{vocab[1]} {vocab[2]}"
    toks = [dbm.tokenize_text(q) for q in queries]
    nq = len(queries)
    dense = []
    for _ in range(nq):
        ids = rng.choice(len(notes), size=60, replace=False).astype(np.int64)
        dense.append((ids, np.sort(rng.random(60))[::-1].copy()))
    hs = HybridSearcher({"retrieval": {"candidate_pool": 50, "hybrid": {
        "enabled": True, "fusion_method": method, "rrf_k": 60,
        "weights": {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}}}})
    rows = dev.scores_sparse_device(toks, cap=dev.SPARSE_CAP, allow_overflow=True)
    assert rows.counts[2] == -1 and rows.counts[7] == -1 and (rows.counts >= 0).sum() >= 6
    rows.free()
    full = dev.scores_device(toks, normalize=True)
    exp = hs.fuse_arrays(nq, dense=dense, bm25=full)
    full.free()
    # the rows as large as they come (round 4): no query leaves the sparse path
    rows = dev.scores_sparse_device(toks, allow_overflow=True)
    assert (rows.counts >= 0).all() and rows.counts[2] > dev.SPARSE_CAP
    rows.free()
    assert hs.fuse_bm25(dev, toks, dense=dense) == exp
    # ... and with rows too small for the frequent words: those two queries take the N-vector
    dev.SPARSE_CAP_MAX = 8192
    rows = dev.scores_sparse_device(toks, allow_overflow=True)
    assert rows.counts[2] == -1 and rows.counts[7] == -1 and (rows.counts >= 0).sum() >= 6
    rows.free()
    assert hs.fuse_bm25(dev, toks, dense=dense) == exp
    # mostly frequent-word queries: the whole batch takes the N-vector path
    toks2 = [dbm.tokenize_text(f"{vocab[i % 3]} {vocab[50 + i]}") for i in range(6)] + toks[:2]
    full2 = dev.scores_device(toks2, normalize=True)
    exp2 = hs.fuse_arrays(len(toks2), dense=dense[:len(toks2)], bm25=full2)
    full2.free()
    assert hs.fuse_bm25(dev, toks2, dense=dense[:len(toks2)]) == exp2
    dev.close()


def test_sparse_producer_and_consumer_refuse_what_they_cannot_hold():
    from anorag_hip import _lib, bm25_search as dbm
    from anorag_hip.fusion import SparseRows, fuse_dense
    rng = np.random.default_rng(14)
    vocab, probs, notes = _corpus(rng, 2000, 500, 12)
    dev = dbm.build_bm25_corpus(notes, lambda n: n["content"])
    toks = [dbm.tokenize_text(vocab[20])]
    with pytest.raises(_lib.AnoragError):          # more than a row can take
        dev.scores_sparse_device(toks, cap=dev.SPARSE_CAP_MAX + 1)
    with pytest.raises(_lib.AnoragError):
        dev.scores_sparse_device(toks, cap=0)
    assert dev.scores_sparse_device([], cap=16).nq == 0          # no queries: nothing to do
    rows = dev.scores_sparse_device(toks, cap=4)                  # a capacity the query does not fit
    assert rows is None
    rows = dev.scores_sparse_device(toks, cap=4, allow_overflow=True)
    assert rows.counts.tolist() == [-1]
    # ... and an overflow-marked row handed STRAIGHT to the fusion is refused, not fused as "every BM25 score
    # is 0.0": by the Python check when the counts are on the host, by the device's error word when they are not
    w0 = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
    dl = [(np.array([1, 2, 3], dtype=np.int64), np.array([0.9, 0.8, 0.7]))]
    with pytest.raises(ValueError):
        fuse_dense("linear", w0, 60.0, 5, 1, {"dense": dl, "bm25": rows})
    from retrieval.hybrid_search import HybridSearcher
    hs0 = HybridSearcher({"retrieval": {"candidate_pool": 5, "hybrid": {"enabled": True, "fusion_method": "linear", "weights": w0}}})
    with pytest.raises((ValueError, _lib.AnoragError)):
        hs0.fuse_arrays(1, dense=dl, bm25=rows)
    wrapped = SparseRows.wrap(rows.ids_ptr, rows.scores_ptr, rows.count_ptr, 1, rows.n, rows.cap, rows.device)  # counts unknown here
    for method in ("linear", "rrf"):
        with pytest.raises(_lib.AnoragError, match="count < 0"):
            fuse_dense(method, w0, 60.0, 5, 1, {"dense": dl, "bm25": wrapped})
    rows.free()
    dev.close()
    # ids outside the array, ids listed twice, a count beyond the capacity
    for bad, what in (((np.array([5, 100]), np.array([1.0, 2.0])), "array_len"), ((np.array([7, 3, 7]), np.array([1.0, 2.0, 3.0])), "twice")):
        r = SparseRows.from_numpy([bad], 50)
        with pytest.raises(_lib.AnoragError, match=what):
            fuse_dense("linear", w0, 60.0, 5, 1, {"dense": dl, "bm25": r})
        r.free()
    # rows beyond the fusion's own sort must arrive in id order
    big_ids = np.arange(9000, dtype=np.uint32)
    big_ids[[10, 11]] = big_ids[[11, 10]]
    r = SparseRows.from_numpy([(np.arange(9000), np.ones(9000))], 20_000)
    _lib.check(_lib.load().anr_device_copy(0, r.ids_ptr, big_ids.ctypes.data, big_ids.nbytes, 0), "anr_device_copy")
    with pytest.raises(_lib.AnoragError, match="not sorted"):
        fuse_dense("linear", w0, 60.0, 5, 1, {"dense": dl, "bm25": r})
    r.free()
    r = SparseRows.from_numpy([(np.array([1, 2]), np.array([1.0, 2.0]))], 50, cap=4)
    cnt = np.array([9], dtype=np.int32)
    _lib.check(_lib.load().anr_device_copy(0, r.count_ptr, cnt.ctypes.data, 4, 0), "anr_device_copy")
    r.counts = None
    with pytest.raises(_lib.AnoragError, match="capacity"):
        fuse_dense("rrf", w0, 60.0, 5, 1, {"dense": dl, "bm25": r})
    r.free()
    ok = SparseRows.from_numpy([(np.array([7, 3]), np.array([1.0, 2.0]))], 50)   # a well-formed row still fuses
    assert fuse_dense("linear", w0, 60.0, 5, 1, {"dense": dl, "bm25": ok})[3][0] == 5
    ok.free()
    w = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
    big = SparseRows(1, 100_000, 65537)            # beyond the fusion's 65 536 entries per row
    with pytest.raises(_lib.AnoragError):
        fuse_dense("linear", w, 60.0, 10, 1, {"bm25": big})
    big.free()
    far = SparseRows.from_numpy([(np.array([5]), np.array([1.0]))], 2**28)   # rrf ranks are packed into 28 bits
    with pytest.raises(_lib.AnoragError):
        fuse_dense("rrf", w, 60.0, 10, 1, {"bm25": far})
    far.free()
