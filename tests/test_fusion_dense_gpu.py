"""GPU parity of the N-array fusion (anr_fuse_dense: HybridSearcher.fuse where bm25 is the full-corpus score vector)
against golden vectors produced by the REFERENCE's own HybridSearcher.fuse on the equivalent N-entry lists
(tests/golden/fusion_long_cases.json, tests/golden/make_golden.py) and, at N = 1 M, against the oracle.
Bar: bit-exact final_similarity (float64), identical ids and order (modulo the reference's set-order ties in
`linear`)."""
import json
import os

import numpy as np
import pytest

from oracle import fusion as ofu

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases():
    with open(os.path.join(GOLD, "fusion_long_cases.json")) as f:
        return json.load(f)["cases"]


def _unpack(v):
    if v is None:
        return None
    a = np.full(v["n"], v["fill"], dtype=np.float64)
    a[v["idx"]] = v["val"]
    return a


def _pairs(lst):
    if lst is None:
        return None
    return (np.array([p[0] for p in lst], dtype=np.int64), np.array([p[1] for p in lst], dtype=np.float64))


def _compare(got, exp, pool, method):
    assert len(got) == len(exp)
    assert [r["final_similarity"] for r in got] == [r["final_similarity"] for r in exp]
    if method == "rrf":
        assert [r["note_id"] for r in got] == [r["note_id"] for r in exp]
    else:
        groups, ggot = {}, {}
        for r in exp:
            groups.setdefault(r["final_similarity"], set()).add(r["note_id"])
        for r in got:
            ggot.setdefault(r["final_similarity"], set()).add(r["note_id"])
        last = exp[-1]["final_similarity"] if exp else None
        for sc, ids in groups.items():
            if sc == last and len(exp) == pool:
                continue  # which of the tied ids make the cut is set-iteration order in the reference
            assert ggot[sc] == ids
    by_id = {r["note_id"]: r for r in exp}
    for r in got:
        if r["note_id"] in by_id:
            assert r["scores"] == by_id[r["note_id"]]["scores"]
            assert r["tags"] == by_id[r["note_id"]]["tags"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_fuse_arrays_matches_reference_golden(case):
    """device arrays for the full-length sources, short lists for the rest (the C5 data path)"""
    from anorag_hip.fusion import DeviceArray
    from retrieval.hybrid_search import HybridSearcher
    hs = HybridSearcher(case["config"])
    bm = _unpack(case["bm25_vec"])
    dv = _unpack(case["dense_vec"])
    held = []

    def arr(v):
        a = DeviceArray.from_numpy(v[None, :])
        held.append(a)
        return a

    got = hs.fuse_arrays(1, dense=arr(dv) if dv is not None else [_pairs(case["dense"])],
                         bm25=arr(bm) if bm is not None else None,
                         graph=[_pairs(case["graph"])], path=[_pairs(case["path"])])[0]
    for a in held:
        a.free()
    _compare(got, case["expected"], hs.candidate_pool, hs.fusion_method)


@pytest.mark.parametrize("name", ["linear_bm25_full_0", "rrf_bm25_full_0", "rrf_bm25_ties", "linear_two_full_vectors",
                                  "rrf_bm25_dense_overlap", "linear_bm25_negative_fill"])
def test_fuse_with_long_lists_matches_reference_golden(name):
    """the drop-in call itself: HybridSearcher.fuse with the N-entry (note_id, score) lists the reference was given —
    more than the LDS kernel's 4096 entries, so the long source is streamed as an array"""
    from retrieval.hybrid_search import HybridSearcher
    case = next(c for c in _cases() if c["name"] == name)
    hs = HybridSearcher(case["config"])
    bm = _unpack(case["bm25_vec"])
    dv = _unpack(case["dense_vec"])
    n = case["n"]
    dense = [(i, float(dv[i])) for i in range(n)] if dv is not None else [tuple(p) for p in case["dense"]]
    bm25 = [(i, float(bm[i])) for i in range(n)] if bm is not None else None
    got = hs.fuse(dense=dense, bm25=bm25, graph=[tuple(p) for p in case["graph"]], path=[tuple(p) for p in case["path"]])
    _compare(got, case["expected"], hs.candidate_pool, hs.fusion_method)


def test_fuse_long_lists_with_string_ids_and_an_unsorted_long_list():
    """string note ids, the long list in an order unrelated to any id numbering (rrf ties follow LIST order), ids
    that occur only in the short lists — against the oracle restatement (pinned by the goldens above)"""
    from retrieval.hybrid_search import HybridSearcher
    rng = np.random.default_rng(5)
    n = 7000
    ids = [f"note_{i:05d}" for i in rng.permutation(n)]
    sc = np.round(np.abs(rng.standard_normal(n)), 1)            # heavy ties
    sc[rng.random(n) < 0.9] = 0.0
    bm25 = [(i, float(s)) for i, s in zip(ids, sc)]
    dense = [(f"note_{int(i):05d}", float(s)) for i, s in zip(rng.choice(n + 50, 100, replace=False), rng.uniform(0.2, 0.9, 100))]
    graph = [(f"note_{int(i):05d}", float(s)) for i, s in zip(rng.choice(n, 20, replace=False), rng.uniform(0, 1, 20))]
    path = [(f"note_{int(i):05d}", float(s)) for i, s in zip(rng.choice(n + 50, 8, replace=False), rng.uniform(0, 1, 8))]
    for method in ("rrf", "linear"):
        cfg = {"retrieval": {"candidate_pool": 70, "hybrid": {"enabled": True, "fusion_method": method, "rrf_k": 60,
                                                             "weights": {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}}}}
        hs = HybridSearcher(cfg)
        got = hs.fuse(dense, bm25, graph, path)
        exp = ofu.fuse(dense, bm25, graph, path, candidate_pool=70, fusion_method=method, weights=hs.weights, rrf_k=60)
        _compare(got, exp, 70, method)


def _long_lists(rng, n, sizes):
    """(dense, bm25, graph, path) lists with the given entry counts over note ids note_00000 .. note_{n-1}"""
    out = []
    for m in sizes:
        pick = rng.choice(n, m, replace=False)
        out.append([(f"note_{int(i):05d}", float(s)) for i, s in zip(pick, rng.uniform(0.01, 1.0, m))])
    return out


@pytest.mark.parametrize("sizes,pool", [((5000, 5000, 900, 0), 50),      # two long lists + 900 graph hits: linear must
                                         ((5000, 5000, 900, 300), 50),    # promote the third source too
                                         ((4100, 700, 600, 500), 80),
                                         ((6000, 10, 5, 3), 1500),         # candidate_pool beyond the kernel's 1024 results
                                         ((3000, 2500, 40, 20), 2600)])
@pytest.mark.parametrize("method", ["linear", "rrf"])
def test_fuse_accepts_every_combination_of_long_lists(sizes, pool, method):
    """HybridSearcher.fuse must not refuse an input the reference accepts: several long lists beside many short
    entries (n_arr * (pool + 2 m) + m <= 4096 inside anr_fuse_dense decides how many sources go as arrays) and
    candidate_pool > 1024 (linear: rounds of 1024 with the selected ids masked; rrf: the sorting path)"""
    from retrieval.hybrid_search import HybridSearcher
    rng = np.random.default_rng(sum(sizes) + pool)
    dense, bm25, graph, path = _long_lists(rng, 9000, sizes)
    w = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
    hs = HybridSearcher({"retrieval": {"candidate_pool": pool, "hybrid": {"enabled": True, "fusion_method": method, "rrf_k": 60,
                                                                         "weights": w}}})
    got = hs.fuse(dense, bm25, graph, path)
    exp = ofu.fuse(dense, bm25, graph, path, candidate_pool=pool, fusion_method=method, weights=w, rrf_k=60)
    _compare(got, exp, pool, method)


def test_rrf_negative_weight_on_the_long_list_takes_the_sorting_path():
    """a negative weight on the streamed source turns its ranking round (the best finals are the LOWEST-valued ids):
    anr_fuse_dense rejects it, HybridSearcher.fuse routes the query to anr_fuse_rrf_long — same result as the reference"""
    from anorag_hip import AnoragError
    from anorag_hip.fusion import DeviceArray, fuse_dense
    from retrieval.hybrid_search import HybridSearcher
    rng = np.random.default_rng(77)
    dense, bm25, graph, path = _long_lists(rng, 8000, (100, 6000, 20, 5))
    w = {"dense": 1.0, "bm25": -0.5, "graph": 0.5, "path": 0.1}
    hs = HybridSearcher({"retrieval": {"candidate_pool": 60, "hybrid": {"enabled": True, "fusion_method": "rrf", "rrf_k": 60,
                                                                       "weights": w}}})
    got = hs.fuse(dense, bm25, graph, path)
    exp = ofu.fuse(dense, bm25, graph, path, candidate_pool=60, fusion_method="rrf", weights=w, rrf_k=60)
    _compare(got, exp, 60, "rrf")
    arr = DeviceArray.from_numpy(rng.random((1, 5000)))
    with pytest.raises(AnoragError):
        fuse_dense("rrf", w, 60.0, 10, 1, {"bm25": arr})
    arr.free()


@pytest.mark.parametrize("method", ["rrf", "linear"])
def test_c5_shape_1m_notes_device_bm25_vector(method):
    """C5 shape: N = 1 M notes, dense = a top-100 list per query, bm25 = abs(normal) kept at ~0.1 %
    of the ids and divided by its maximum, weights {1.0, 0.5, 0.5, 0.1}, rrf_k 60, pool 80; the stated 200 queries in ONE
    call (every one checked against the oracle), one of them with an all-zero bm25 vector, one whose dense hits are the first ids (they win the zero ties)."""
    from anorag_hip.fusion import DeviceArray
    from retrieval.hybrid_search import HybridSearcher
    rng = np.random.default_rng(99)
    n, nq, pool = 1_000_000, 200, 80
    w = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
    hs = HybridSearcher({"retrieval": {"candidate_pool": pool, "hybrid": {"enabled": True, "fusion_method": method,
                                                                         "rrf_k": 60, "weights": w}}})
    bm = np.zeros((nq, n), dtype=np.float64)
    dense, graph, path = [], [], []
    for q in range(nq):
        if q != 3:
            nz = rng.choice(n, size=1000, replace=False)
            v = np.abs(rng.standard_normal(1000))
            bm[q, nz] = v / v.max()
        d_ids = np.arange(100) if q == 5 else rng.choice(n, 100, replace=False)
        dense.append((d_ids.astype(np.int64), np.sort(rng.uniform(0.2, 0.9, 100))[::-1].copy()))
        graph.append((rng.choice(n, 15, replace=False).astype(np.int64), rng.uniform(0, 1, 15)))
        path.append((rng.choice(n, 5, replace=False).astype(np.int64), rng.uniform(0, 1, 5)))
    arr = DeviceArray.from_numpy(bm)
    got, st = hs.fuse_arrays(nq, dense=dense, bm25=arr, graph=graph, path=path, want_stats=True)
    arr.free()
    assert st["scan_bytes"] == nq * n * 8 and st["n_queries"] == nq
    full = np.arange(n, dtype=np.int64)
    for q in range(nq):
        ids, fin = ofu.fuse_arrays(n, (dense[q], (full, bm[q]), graph[q], path[q]), [w[k] for k in ("dense", "bm25", "graph", "path")],
                                   method, 60, pool)
        assert [r["final_similarity"] for r in got[q]] == fin.tolist()
        if method == "rrf":
            assert [r["note_id"] for r in got[q]] == ids.tolist()
        else:
            assert sorted(r["note_id"] for r in got[q] if r["final_similarity"] > fin[-1]) == \
                sorted(int(i) for i, f in zip(ids, fin) if f > fin[-1])


def test_adversarial_orders_keep_the_candidate_lists_bounded():
    """ascending and constant score vectors: chunk 0's threshold admits everything after it, so every chunk has to
    select its own K' best and raise the running threshold — results still exact"""
    from anorag_hip.fusion import DeviceArray, fuse_dense
    n, pool = 300_000, 64
    w = {"dense": 1.0, "bm25": 0.5, "graph": 0.0, "path": 0.0}
    asc = np.linspace(0.0, 1.0, n)
    const = np.full(n, 0.25)
    steps = np.repeat(np.arange(n // 1000), 1000).astype(np.float64)  # 1000-way ties, ascending plateaus
    bm = np.stack([asc, const, steps])
    dense = [(np.array([5, 7, n - 1], dtype=np.int64), np.array([0.9, 0.8, 0.7]))] * 3
    arr = DeviceArray.from_numpy(bm)
    for method in ("linear", "rrf"):
        ids, fin, _, cnt, st = fuse_dense(method, w, 60.0, pool, 3, {"dense": dense, "bm25": arr}, want_stats=True)
        full = np.arange(n, dtype=np.int64)
        for q in range(3):
            e_ids, e_fin = ofu.fuse_arrays(n, (dense[q], (full, bm[q]), None, None), [1.0, 0.5, 0.0, 0.0], method, 60, pool)
            assert cnt[q] == pool and fin[q].tolist() == e_fin.tolist()
            assert ids[q].tolist() == e_ids.tolist()
        assert st["n_candidates"] <= 3 * ((n + 4095) // 4096) * 1100
    arr.free()


def test_fuse_dense_argument_errors():
    from anorag_hip import AnoragError
    from anorag_hip.fusion import DeviceArray, fuse_dense
    a = DeviceArray.from_numpy(np.zeros((1, 5000)))
    w = {"dense": 1.0, "bm25": 0.5}
    with pytest.raises(AnoragError):  # rrf with two array sources
        fuse_dense("rrf", w, 60.0, 10, 1, {"dense": a, "bm25": a})
    with pytest.raises(AnoragError):  # more short-list entries than the arrays' side lists may hold
        fuse_dense("linear", w, 60.0, 10, 1, {"bm25": a, "dense": [(np.arange(2000), np.ones(2000))]})
    with pytest.raises(AnoragError):  # no array at all
        fuse_dense("linear", w, 60.0, 10, 1, {"dense": [(np.arange(5), np.ones(5))]})
    a.free()


@pytest.mark.parametrize("nq", [1, 3, 5])
def test_linear_max_pass_with_an_odd_number_of_queries(nq):
    """the device max pass (k_fd_max: a 64-bit atomicMax per source and query) with an ODD number of queries and no
    maxima supplied: the 64-bit work words sit behind a 4100-byte-per-query histogram and must stay 8-byte aligned —
    a 4-byte-aligned 64-bit atomic is a bus error (the SIGBUS seen once during round 2's arena rewrite)"""
    from anorag_hip.fusion import DeviceArray, fuse_dense
    rng = np.random.default_rng(nq)
    n, pool = 30_000, 40
    a = rng.random((nq, n))
    arr = DeviceArray.from_numpy(a)            # no row maxima: the kernel finds them
    dense = [(rng.choice(n, 50, replace=False).astype(np.int64), rng.random(50)) for _ in range(nq)]
    w = {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}
    ids, fin, _, cnt = fuse_dense("linear", w, 60.0, pool, nq, {"dense": dense, "bm25": arr})
    arr.free()
    full = np.arange(n, dtype=np.int64)
    for q in range(nq):
        _, f = ofu.fuse_arrays(n, (dense[q], (full, a[q]), None, None), [1.0, 0.5, 0.5, 0.1], "linear", 60, pool)
        assert cnt[q] == pool and fin[q].tolist() == f.tolist()


def test_host_supplied_row_maxima_match_the_device_max_pass():
    """DeviceArray.from_numpy(with_max=True): rows with NaN (absent ids), an all-NaN row, an all-negative row, an
    all-zero row — the linear fusion with the supplied maxima equals the one that runs k_fd_max"""
    from anorag_hip.fusion import DeviceArray, fuse_dense
    rng = np.random.default_rng(31)
    n, nq, pool = 30_000, 5, 48
    a = rng.standard_normal((nq, n))
    a[0, rng.random(n) < 0.5] = np.nan
    a[1, :] = np.nan
    a[2, :] = -np.abs(a[2]) - 0.5
    a[3, :] = 0.0
    dense = [(rng.choice(n, 40, replace=False).astype(np.int64), rng.uniform(0.1, 0.9, 40)) for _ in range(nq)]
    w = {"dense": 1.0, "bm25": 0.7, "graph": 0.0, "path": 0.0}
    plain = DeviceArray.from_numpy(a)
    known = DeviceArray.from_numpy(a, with_max=True)
    assert known.row_max is not None and plain.row_max is None
    r0 = fuse_dense("linear", w, 60.0, pool, nq, {"dense": dense, "bm25": plain})
    r1 = fuse_dense("linear", w, 60.0, pool, nq, {"dense": dense, "bm25": known})
    for x, y in zip(r0, r1):
        assert np.array_equal(x, y, equal_nan=True)
    plain.free()
    known.free()


def test_one_long_query_drains_the_deferred_rank_searches_inside_the_scan():
    """ONE query over 48 M notes at 1 % density: every workgroup walks ~23 chunks of the same query and every wave defers
    ~5 rank searches per chunk, so the per-wave LDS segments fill and are searched INSIDE the scan (the drain protocol:
    flag raised one chunk ahead, acted on after the next chunk's barrier) — the 1 M-note shapes above never get there,
    they finish a query before a segment fills.  rrf (exact ranks of the short-list ids among all 48 M) and linear."""
    from anorag_hip.fusion import DeviceArray, fuse_dense
    rng = np.random.default_rng(77)
    n, pool = 48_000_000, 60
    bm = np.zeros((1, n), dtype=np.float64)
    nz = rng.choice(n, n // 100, replace=False)
    bm[0, nz] = np.round(np.abs(rng.standard_normal(len(nz))), 3) + 0.001    # ties among the non-zero values too
    d_ids = np.concatenate([rng.choice(nz, 40, replace=False), rng.choice(n, 60, replace=False)]).astype(np.int64)
    d_ids = np.unique(d_ids)
    dense = [(d_ids, np.sort(rng.uniform(0.2, 0.9, len(d_ids)))[::-1].copy())]
    w = {"dense": 1.0, "bm25": 0.5, "graph": 0.0, "path": 0.0}
    arr = DeviceArray.from_numpy(bm, with_max=True)
    full = np.arange(n, dtype=np.int64)
    for method in ("rrf", "linear"):
        ids, fin, _, cnt, st = fuse_dense(method, w, 60.0, pool, 1, {"dense": dense, "bm25": arr}, want_stats=True)
        e_ids, e_fin = ofu.fuse_arrays(n, (dense[0], (full, bm[0]), None, None), [1.0, 0.5, 0.0, 0.0], method, 60, pool)
        assert cnt[0] == pool and fin[0].tolist() == e_fin.tolist()
        if method == "rrf":
            assert ids[0].tolist() == e_ids.tolist()
        else:
            assert sorted(ids[0][fin[0] > e_fin[-1]].tolist()) == sorted(e_ids[e_fin > e_fin[-1]].tolist())
        assert st["scan_bytes"] == n * 8
    arr.free()


@pytest.mark.parametrize("method", ["rrf", "linear"])
def test_float32_score_arrays_equal_the_float64_arrays_of_the_same_values(method):
    """array_dtype = 1: a float32 vector stands for the float64 values it converts to exactly — same results as the
    float64 array of those values (and as the oracle on them); mixed with a second, float64, array source for linear;
    lengths that end inside a chunk"""
    from anorag_hip.fusion import DeviceArray, fuse_dense
    rng = np.random.default_rng(13)
    n, nq, pool = 150_001, 3, 50
    v32 = np.zeros((nq, n), dtype=np.float32)
    for q in range(nq):
        nz = rng.choice(n, 900, replace=False)
        v32[q, nz] = np.abs(rng.standard_normal(900)).astype(np.float32)
    v32[1, rng.choice(n, 200, replace=False)] = np.nan  # absent ids
    v64 = v32.astype(np.float64)
    dense = [(rng.choice(n, 70, replace=False).astype(np.int64), np.sort(rng.uniform(0.1, 0.9, 70))[::-1].copy())
             for _ in range(nq)]
    w = {"dense": 1.0, "bm25": 0.5, "graph": 0.3, "path": 0.0}
    a32, a64 = DeviceArray.from_numpy(v32), DeviceArray.from_numpy(v64)
    assert a32.dtype == np.float32
    r32 = fuse_dense(method, w, 60.0, pool, nq, {"dense": dense, "bm25": a32})
    r64 = fuse_dense(method, w, 60.0, pool, nq, {"dense": dense, "bm25": a64})
    for x, y in zip(r32, r64):
        assert np.array_equal(x, y, equal_nan=True)
    full = np.arange(n, dtype=np.int64)
    for q in range(nq):
        ok = ~np.isnan(v64[q])
        e_ids, e_fin = ofu.fuse_arrays(n, (dense[q], (full[ok], v64[q][ok]), None, None), [1.0, 0.5, 0.3, 0.0], method, 60, pool)
        assert r32[1][q].tolist() == e_fin.tolist()
    if method == "linear":  # a second array source (graph), float64 and shorter than the first
        g64 = np.abs(rng.standard_normal((nq, 100_000)))
        g64[:, rng.random(100_000) < 0.99] = 0.0
        ag = DeviceArray.from_numpy(g64)
        m32 = fuse_dense("linear", w, 60.0, pool, nq, {"dense": dense, "bm25": a32, "graph": ag})
        m64 = fuse_dense("linear", w, 60.0, pool, nq, {"dense": dense, "bm25": a64, "graph": ag})
        for x, y in zip(m32, m64):
            assert np.array_equal(x, y, equal_nan=True)
        for q in range(nq):
            ok = ~np.isnan(v64[q])
            e_ids, e_fin = ofu.fuse_arrays(n, (dense[q], (full[ok], v64[q][ok]), (np.arange(100_000, dtype=np.int64), g64[q]), None),
                                           [1.0, 0.5, 0.3, 0.0], "linear", 60, pool)
            assert m32[1][q].tolist() == e_fin.tolist()
        ag.free()
    a32.free()
    a64.free()


def _rrf_multi_cases():
    with open(os.path.join(GOLD, "fusion_rrf_multi_long_cases.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _rrf_multi_cases(), ids=lambda c: c["name"])
def test_rrf_with_two_or_three_full_corpus_lists_matches_reference_golden(case):
    """HybridSearcher.fuse, method rrf, where two or three lists hold one entry per note (anr_fuse_rrf_long: every ranked
    list sorted on the device in its own order) — the reference's own output on the same lists: ids, order, finals
    bit for bit, per-source scores and tags"""
    from retrieval.hybrid_search import HybridSearcher
    hs = HybridSearcher(case["config"])
    n = case["n"]
    lists = {}
    for k in ("dense", "bm25", "graph", "path"):
        if k in case["vectors"]:
            lists[k] = [(i, v) for i, v in enumerate(case["vectors"][k])]
        else:
            lists[k] = [tuple(p) for p in case["short"].get(k, [])]
    got = hs.fuse(dense=lists["dense"], bm25=lists["bm25"], graph=lists["graph"], path=lists["path"])
    exp = case["expected"]
    assert [r["note_id"] for r in got] == [r["note_id"] for r in exp]
    assert [r["final_similarity"] for r in got] == [r["final_similarity"] for r in exp]
    assert [r["scores"] for r in got] == [r["scores"] for r in exp] and [r["tags"] for r in got] == [r["tags"] for r in exp]
    assert len(got) <= hs.candidate_pool and n >= len(got)


def test_rrf_long_lists_in_unrelated_orders_with_string_ids_and_heavy_ties():
    """two long lists whose orders have nothing to do with each other (a stable sort keeps EACH list's own order among
    equal scores), string ids, ids missing from one list or the other, a long graph list as well — against the oracle
    restatement (pinned by the goldens above)"""
    from retrieval.hybrid_search import HybridSearcher
    rng = np.random.default_rng(17)
    n = 9000
    names = [f"n{i:05d}" for i in range(n)]
    dense = [(names[i], float(s)) for i, s in zip(rng.permutation(n)[:8000], np.round(rng.uniform(0, 1, 8000), 2))]
    bm25 = [(names[i], float(s)) for i, s in zip(rng.permutation(n)[:8500], np.round(np.abs(rng.standard_normal(8500)), 1))]
    graph = [(names[i], float(s)) for i, s in zip(rng.permutation(n)[:5000], np.round(rng.uniform(0, 1, 5000), 1))]
    path = [(names[i], float(s)) for i, s in zip(rng.choice(n, 12, replace=False), rng.uniform(0, 1, 12))] + [("only_in_path", 0.9)]
    for w in ({"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}, {"dense": 0.3, "bm25": 1.0, "graph": 0.0, "path": 2.0}):
        cfg = {"retrieval": {"candidate_pool": 90, "hybrid": {"enabled": True, "fusion_method": "rrf", "rrf_k": 7, "weights": w}}}
        hs = HybridSearcher(cfg)
        got = hs.fuse(dense, bm25, graph, path)
        exp = ofu.fuse(dense, bm25, graph, path, candidate_pool=90, fusion_method="rrf", weights=hs.weights, rrf_k=7)
        assert [r["note_id"] for r in got] == [r["note_id"] for r in exp]
        assert [r["final_similarity"] for r in got] == [r["final_similarity"] for r in exp]
        assert [r["scores"] for r in got] == [r["scores"] for r in exp]
    assert all(r["note_id"] != "only_in_path" for r in got)


@pytest.mark.parametrize("slot", ["dense", "graph", "path"])
def test_linear_with_the_one_array_in_any_slot_takes_the_barrier_free_pass(slot):
    """linear with ONE array source (k_fd_scan_free; the path slot has its own arithmetic: w * x, no normalisation): a
    sparse vector — nothing overflows — and a dense one — every query is flagged and re-done by the regular scan —
    against the oracle over several chunks"""
    from anorag_hip.fusion import DeviceArray, fuse_dense
    rng = np.random.default_rng({"dense": 1, "graph": 2, "path": 3}[slot])
    n, nq, pool = 30_000, 3, 60
    w = {"dense": 0.8, "bm25": 0.6, "graph": 0.4, "path": 0.3}
    other = "bm25"
    for density in (0.002, 1.0):
        a = np.zeros((nq, n))
        for q in range(nq):
            ids = rng.choice(n, size=max(1, int(n * density)), replace=False)
            a[q, ids] = np.abs(rng.standard_normal(len(ids)))
        lists = []
        for q in range(nq):
            ids = rng.choice(n, size=50, replace=False).astype(np.int64)
            lists.append((ids, np.sort(rng.random(50))[::-1].copy()))
        arr = DeviceArray.from_numpy(a)
        got = fuse_dense("linear", w, 60.0, pool, nq, {slot: arr, other: lists})
        arr.free()
        full = np.arange(n, dtype=np.int64)
        for q in range(nq):
            src = {slot: (full, a[q]), other: lists[q]}
            ids, fin = ofu.fuse_arrays(n, tuple(src.get(k) for k in ("dense", "bm25", "graph", "path")),
                                       [w["dense"], w["bm25"], w["graph"], w["path"]], "linear", 60.0, pool)
            cnt = int(got[3][q])
            assert cnt == len(fin) and got[1][q, :cnt].tolist() == fin.tolist(), (slot, density, q)
            assert got[0][q, :cnt].tolist() == ids.tolist(), (slot, density, q)
