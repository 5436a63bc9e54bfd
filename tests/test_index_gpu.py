"""GPU parity tests of the exact flat index (HIP path through the C ABI) against the oracle.

Bar (BASELINE.json north_star): identical top-k id sets, scores within 1e-4 (float32).  The oracle
ranks by the float64-accumulated score; rows closer than 1e-6 to the k-th score are near-ties that no
float32 reference resolves reproducibly, so id sets are compared modulo those (none occur on the seeds
used here unless a test says it builds them on purpose).
"""
import numpy as np
import pytest

from oracle import flat_index as orc

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-4


def _data(n, d, nq, seed=1234, qseed=4321):
    x = np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32)
    q = np.random.default_rng(qseed).standard_normal((nq, d), dtype=np.float32)
    return x, q


def _check(index, x, q, k, metric_name, normalize):
    D, I = index.search(q, k)
    xm = orc.preprocess_vectors(x, "cosine" if normalize else "raw")
    qm = orc.preprocess_vectors(q, "cosine" if normalize else "raw")
    Dr, Ir = orc.flat_search(qm, xm, k, metric_name)
    s64 = orc.exact_scores(qm, xm, metric_name)
    assert I.shape == Ir.shape and D.dtype == np.float32 and I.dtype == np.int64
    assert orc.near_tie_equal(I, Ir, s64, k, 1e-6), "top-k id sets differ beyond near-ties"
    # scores: compare position-wise (sorted best-first on both sides)
    valid = Ir >= 0
    assert np.array_equal(I >= 0, valid)
    assert np.max(np.abs(D[valid] - Dr[valid]), initial=0.0) <= SCORE_TOL
    # best-first order
    if metric_name == "ip":
        assert np.all(np.diff(D, axis=1)[valid[:, 1:]] <= 0)
    else:
        assert np.all(np.diff(D, axis=1)[valid[:, 1:]] >= 0)
    return D, I, Dr, Ir


def test_c1_shape_dense_path():
    """C1: 10k x 384 (all-MiniLM-L6-v2 shape), batch-1 top-10, cosine."""
    from anorag_hip import FlatIndex, METRIC_IP
    x, q = _data(10_000, 384, 1)
    idx = FlatIndex(384, METRIC_IP, normalize=True)
    idx.add(x)
    assert idx.ntotal == 10_000
    D, I, Dr, Ir = _check(idx, x, q, 10, "ip", True)
    assert np.array_equal(I, Ir)
    st = idx.last_stats()
    assert st["n_fallback"] == 0
    idx.close()


def test_sparse_path_batch64_top100():
    """C2 shape at reduced N: 200k x 768, batch-64 top-100 — exercises sample, ladder, candidate lists."""
    from anorag_hip import FlatIndex, METRIC_IP
    x, q = _data(200_000, 768, 64)
    idx = FlatIndex(768, METRIC_IP, normalize=True)
    idx.add(x)
    D, I, Dr, Ir = _check(idx, x, q, 100, "ip", True)
    st = idx.last_stats()
    assert st["sample_rows"] > 0, "expected the threshold-gated scan"
    assert st["n_overflow"] == 0
    assert st["n_fallback"] <= 2, st
    # stored rows are the normalised rows
    rec = idx.reconstruct_n(123, 5)
    ref = orc.preprocess_vectors(x[123:128], "cosine")
    assert np.max(np.abs(rec - ref)) < 1e-6
    idx.close()


def test_force_exact_path_matches():
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip._lib import OPT_FORCE_EXACT
    x, q = _data(20_000, 256, 7)
    idx = FlatIndex(256, METRIC_IP, normalize=True)
    idx.add(x)
    D1, I1 = idx.search(q, 50)
    idx.set_option(OPT_FORCE_EXACT, 1)
    D2, I2 = idx.search(q, 50)
    assert idx.last_stats()["n_fallback"] == 7
    assert np.array_equal(I1, I2)
    assert np.array_equal(D1, D2)
    _check(idx, x, q, 50, "ip", True)
    idx.close()


def test_l2_metric():
    from anorag_hip import FlatIndex, METRIC_L2
    x, q = _data(50_000, 128, 16)
    idx = FlatIndex(128, METRIC_L2, normalize=False)
    idx.add(x)
    D, I, Dr, Ir = _check(idx, x, q, 20, "l2", False)
    idx.close()


def test_incremental_add_and_ragged_sizes():
    """adds that do not end on a 32-row tile boundary, k > ntotal padding, reset."""
    from anorag_hip import FlatIndex, METRIC_IP
    x, q = _data(1000, 100, 3)   # dim not a multiple of 16
    idx = FlatIndex(100, METRIC_IP, normalize=True)
    idx.add(x[:7])
    D, I = idx.search(q, 10)
    assert np.all(I[:, 7:] == -1) and np.all(I[:, :7] >= 0)
    assert np.all(D[:, 7:] == -orc.FLT_MAX)
    idx.add(x[7:45])
    idx.add(x[45:46])
    idx.add(x[46:])
    assert idx.ntotal == 1000
    _check(idx, x, q, 10, "ip", True)
    idx.reset()
    assert idx.ntotal == 0
    D, I = idx.search(q, 4)
    assert np.all(I == -1)
    idx.add(x[:33])
    _check(idx, x[:33], q, 40, "ip", True)
    idx.close()


def test_duplicates_and_zero_rows_tie_order():
    """exact duplicate rows (the reference's "Empty note" placeholder, embedding_manager.py:448) and a
    zero row: ties must come out in ascending id order and zero-norm rows stay zero."""
    from anorag_hip import FlatIndex, METRIC_IP
    x, q = _data(10_000, 384, 4)
    x[5000:5016] = x[17]          # 16 duplicates of row 17
    x[42] = 0.0
    q[0] = x[17] + 0.01 * q[0]    # make the duplicate group the best match of query 0
    idx = FlatIndex(384, METRIC_IP, normalize=True)
    idx.add(x)
    D, I, Dr, Ir = _check(idx, x, q, 10, "ip", True)
    assert np.array_equal(I[0], Ir[0])
    assert I[0][0] == 17 and list(I[0][1:10]) == list(range(5000, 5009))
    assert np.all(idx.reconstruct_n(42, 1) == 0.0)
    idx.close()


def test_async_pipelined_search_equals_sync():
    """anr_index_search_dev_async + anr_index_sync: overlapping batches on rotating streams give the same
    answers as the synchronous call; a multi-batch synchronous call (nq > 64) uses the same pipeline."""
    import torch
    from anorag_hip import FlatIndex, METRIC_IP
    x, q = _data(150_000, 256, 64 * 5 + 7)
    idx = FlatIndex(256, METRIC_IP, normalize=True)
    idx.add(x)
    D_all, I_all = idx.search(q, 30)                      # 6 batches through the pipelined path
    xm, qm = orc.preprocess_vectors(x), orc.preprocess_vectors(q)
    Dr, Ir = orc.flat_search(qm, xm, 30, "ip")
    assert orc.near_tie_equal(I_all, Ir, orc.exact_scores(qm, xm, "ip"), 30, 1e-6)
    assert np.max(np.abs(D_all - Dr)) <= SCORE_TOL
    dev = torch.device("cuda", 0)
    qt = torch.from_numpy(q).to(dev)
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = []
    idx.reset_stats()
    for b in range(5):
        D = torch.empty((64, 30), device=dev)
        I = torch.empty((64, 30), device=dev, dtype=torch.int64)
        idx.search_device_async(qt[b * 64:(b + 1) * 64].data_ptr(), 64, 30, D.data_ptr(), I.data_ptr(),
                                streams[b % 3].cuda_stream)
        outs.append((D, I))
    idx.sync()
    torch.cuda.synchronize()
    for b, (D, I) in enumerate(outs):
        assert np.array_equal(I.cpu().numpy(), I_all[b * 64:(b + 1) * 64])
        assert np.array_equal(D.cpu().numpy(), D_all[b * 64:(b + 1) * 64])
    assert idx.last_stats()["n_queries"] == 320
    # ANR_OPT_STREAMS = 1: batches strictly one after the other on a single stream — same answers
    from anorag_hip._lib import OPT_STREAMS, AnoragError
    idx.set_option(OPT_STREAMS, 1)
    D1, I1 = idx.search(q, 30)
    assert np.array_equal(I1, I_all) and np.array_equal(D1, D_all)
    with pytest.raises(AnoragError):
        idx.set_option(OPT_STREAMS, 4)
    idx.close()


def test_c2_full_size_1m_x_768_batch64_top100():
    """BASELINE config C2 at full size: 1M x 768, batch-64 top-100, against the float64 oracle, plus the
    size-independent properties (sortedness, unique in-range ids, agreement with the independent dense exact
    path, self-retrieval)."""
    import torch
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip._lib import OPT_FORCE_EXACT
    n, d, nq, k = 1_000_000, 768, 64, 100
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    idx = FlatIndex(d, METRIC_IP, normalize=True)
    idx.reserve(n)
    host = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, 250_000):
        xb = torch.randn((250_000, d), generator=g, device=dev)
        torch.cuda.synchronize()
        idx.add_device(xb.data_ptr(), xb.shape[0])
        host[s:s + 250_000] = xb.cpu().numpy()
    del xb
    q = np.random.default_rng(4321).standard_normal((nq, d), dtype=np.float32)
    q[7] = host[123_456]                                  # self-retrieval probe
    D, I = idx.search(q, k)
    st = idx.last_stats()
    assert st["n_overflow"] == 0 and st["n_fallback"] <= 2, st
    # properties
    assert np.all(np.diff(D, axis=1) <= 0)
    assert all(len(set(r)) == k for r in I.tolist()) and I.min() >= 0 and I.max() < n
    assert I[7, 0] == 123_456 and abs(D[7, 0] - 1.0) < 1e-5
    # oracle (float64 arbitration) at full size
    xn, qn = orc.preprocess_vectors(host), orc.preprocess_vectors(q)
    del host
    s64 = orc.exact_scores(qn, xn, "ip", block=100_000)
    for i in range(nq):
        kth = np.partition(s64[i], n - k)[n - k]
        ref = set(np.nonzero(s64[i] >= kth)[0].tolist())
        got = set(I[i].tolist())
        assert all(abs(s64[i][r] - kth) <= 1e-6 for r in got ^ ref), f"query {i}: id set differs beyond near-ties"
        assert np.max(np.abs(D[i] - s64[i][I[i]].astype(np.float32))) <= SCORE_TOL
    # the dense exact path is an independent algorithm: identical output
    idx.set_option(OPT_FORCE_EXACT, 1)
    D2, I2 = idx.search(q[:8], k)
    assert np.array_equal(I2, I[:8]) and np.array_equal(D2, D[:8])
    idx.close()


def test_score_rows_matches_oracle():
    """anr_index_score_rows: gather(rows, ids) . q with the index's query preprocessing (cosine / L2)."""
    from anorag_hip import FlatIndex, METRIC_IP, METRIC_L2
    x, q = _data(5000, 200, 70)
    ids = np.random.default_rng(3).integers(0, 5000, size=(70, 13))
    ids[0, 0] = -1
    ids[1, 1] = 5000
    idx = FlatIndex(200, METRIC_IP, normalize=True)
    idx.add(x)
    got = idx.score_rows(q, ids)
    xn, qn = orc.preprocess_vectors(x), orc.preprocess_vectors(q)
    ref = np.einsum("qd,qjd->qj", qn.astype(np.float64), xn[np.clip(ids, 0, 4999)].astype(np.float64))
    assert np.isnan(got[0, 0]) and np.isnan(got[1, 1])
    ok = np.ones_like(got, dtype=bool)
    ok[0, 0] = ok[1, 1] = False
    assert np.max(np.abs(got[ok] - ref[ok])) <= 1e-6
    idx.close()
    l2 = FlatIndex(200, METRIC_L2, normalize=False)
    l2.add(x)
    got = l2.score_rows(q, ids)
    ref = ((q[:, None, :].astype(np.float64) - x[np.clip(ids, 0, 4999)].astype(np.float64)) ** 2).sum(-1)
    assert np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) <= 1e-6
    l2.close()


def test_tight_clusters_second_pass_is_exact():
    """near-duplicate neighbourhoods: hundreds of rows within the f16 error bound of the k-th score, so the
    certificate fails and the fixed-threshold second pass must deliver the exact answer (cosine and L2)."""
    from anorag_hip import FlatIndex, METRIC_IP, METRIC_L2
    rng = np.random.default_rng(11)
    n, d, nq, k = 150_000, 256, 64, 50
    cent = rng.standard_normal((300, d)).astype(np.float32)
    x = (cent[rng.integers(0, 300, n)] + 0.02 * rng.standard_normal((n, d))).astype(np.float32)
    q = (cent[rng.integers(0, 300, nq)] + 0.02 * rng.standard_normal((nq, d))).astype(np.float32)
    idx = FlatIndex(d, METRIC_IP, normalize=True)
    idx.add(x)
    D, I, Dr, Ir = _check(idx, x, q, k, "ip", True)
    st = idx.last_stats()
    assert st["n_fallback"] > 0, "this data is meant to defeat the certificate"
    assert st["n_dense_exact"] == 0, st
    assert st["n_from_lists"] > 0, st        # theta above the scan's own threshold: recovered without a second scan
    # adaptive overfetch: after batches in which most certificates failed, the automatic K' grows (here to its
    # 4x cap, 512) so that later batches certify without the second scan — and the answers stay exact
    assert st["overfetch"] == 128
    for _ in range(3):
        _check(idx, x, q, k, "ip", True)
    st = idx.last_stats()
    assert st["overfetch"] == 512, st
    assert st["n_fallback"] < nq // 2, st    # far fewer certificates fail with 512 candidates than with 128
    idx.close()
    l2 = FlatIndex(d, METRIC_L2, normalize=False)
    l2.add(x)
    _check(l2, x, q, k, "l2", False)
    assert l2.last_stats()["n_dense_exact"] == 0
    l2.close()


def test_two_shards_id_offset_packed_exchange_and_strided_merge():
    """The N > 1 data path of bench.py on one device: two row shards with ANR_OPT_ID_OFFSET, results written
    into packed [B*k f32 | B*k i64] buffers, the two buffers laid out as one all-gather would leave them,
    anr_merge_topk_strided_dev — equals the single-index oracle; -1 padding is not shifted by the offset."""
    import ctypes as C
    import torch
    from anorag_hip import FlatIndex, METRIC_IP, _lib
    from anorag_hip._lib import OPT_ID_OFFSET
    n, d, B, k = 90_000, 128, 64, 20
    x, q = _data(n, d, B)
    cut = 50_016
    dev = torch.device("cuda", 0)
    qt = torch.from_numpy(q).to(dev)
    from anorag_hip.sharded import packed_layout
    nres = B * k
    id_off, part = packed_layout(nres)
    packed = torch.zeros(2 * part, device=dev, dtype=torch.uint8)
    shards = []
    for r, (lo, hi) in enumerate(((0, cut), (cut, n))):
        idx = FlatIndex(d, METRIC_IP, normalize=True)
        idx.add(x[lo:hi])
        idx.set_option(OPT_ID_OFFSET, lo)
        base = packed.data_ptr() + r * part
        idx.search_device(qt.data_ptr(), B, k, base, base + id_off)
        shards.append(idx)
    torch.cuda.synchronize()
    Dm = torch.empty((B, k), device=dev)
    Im = torch.empty((B, k), device=dev, dtype=torch.int64)
    lib = _lib.load()
    _lib.check(lib.anr_merge_topk_strided_dev(0, C.c_void_p(packed.data_ptr()), C.c_void_p(packed.data_ptr() + id_off),
                                              part // 4, part // 8, 2, B, k, 1, C.c_void_p(Dm.data_ptr()),
                                              C.c_void_p(Im.data_ptr()), C.c_void_p(0)), "merge")
    torch.cuda.synchronize()
    xm, qm = orc.preprocess_vectors(x), orc.preprocess_vectors(q)
    Dr, Ir = orc.flat_search(qm, xm, k, "ip")
    assert orc.near_tie_equal(Im.cpu().numpy(), Ir, orc.exact_scores(qm, xm, "ip"), k, 1e-6)
    assert np.max(np.abs(Dm.cpu().numpy() - Dr)) <= SCORE_TOL
    # second shard alone: ids are global; asking for more than it holds pads with -1 (not offset - 1)
    tiny = FlatIndex(d, METRIC_IP, normalize=True)
    tiny.add(x[:5])
    tiny.set_option(OPT_ID_OFFSET, 1000)
    D, I = tiny.search(q[:3], 8)
    assert np.array_equal(np.sort(I[:, :5], axis=1), np.tile(np.arange(1000, 1005), (3, 1)))
    assert np.all(I[:, 5:] == -1)
    for s in shards:
        s.close()
    tiny.close()


def test_sample_that_overestimates_the_corpus_still_gives_exact_results():
    """The scan's start threshold is a guess from a strided tile sample.  Here the guess is as wrong as it can
    be: every row that scores high sits in a SAMPLED tile (tile index a multiple of the sample stride), so the
    sample predicts thousands of rows above the threshold while the corpus holds only the planted ones — fewer
    than K' candidates come out of the scan, the certificate must fail and the second pass must restore the
    exact answer.  Also the opposite order: all high rows packed into one UNSAMPLED stretch (threshold far too
    low for that region) must stay exact without overflowing into the dense path."""
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip._lib import OPT_SAMPLE_ROWS
    rng = np.random.default_rng(5)
    n, d, nq, k = 200_000, 64, 8, 10
    sample_rows = 12288                      # 384 sample tiles -> stride 6250 // 384 = 16 tiles
    stride = (n // 32) // (sample_rows // 32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    x = rng.standard_normal((n, d)).astype(np.float32)
    for t in range(0, (sample_rows // 32)):  # one near-copy of every query in each sampled tile
        r0 = t * stride * 32
        for j in range(nq):
            x[r0 + j] = q[j] + 0.02 * (t + 1) * rng.standard_normal(d).astype(np.float32)  # cosine falls with t
    idx = FlatIndex(d, METRIC_IP, normalize=True)
    idx.add(x)
    idx.set_option(OPT_SAMPLE_ROWS, sample_rows)
    _check(idx, x, q, k, "ip", True)
    st = idx.last_stats()
    assert st["n_fallback"] == nq, st        # every query: fewer than K' candidates above the guessed threshold
    assert st["n_dense_exact"] == 0, st      # ... and the recovery (not the dense path) answered them
    assert st["n_from_lists"] == nq, st      # theta (10th best - eps) is above the scan threshold (16th best)
    idx.close()

    # same construction with near-identical planted rows: the 10th best is within eps of the scan's threshold, so
    # theta falls BELOW it and the lists cannot be trusted to hold every row above theta -> the theta re-scan runs
    x3 = rng.standard_normal((n, d)).astype(np.float32)
    for t in range(0, (sample_rows // 32)):
        r0 = t * stride * 32
        for j in range(nq):
            x3[r0 + j] = q[j] + 1e-4 * rng.standard_normal(d).astype(np.float32)
    idx = FlatIndex(d, METRIC_IP, normalize=True)
    idx.add(x3)
    idx.set_option(OPT_SAMPLE_ROWS, sample_rows)
    D, I = idx.search(q, k)
    st = idx.last_stats()
    assert st["n_fallback"] == nq and st["n_from_lists"] < nq, st
    s64 = orc.exact_scores(orc.preprocess_vectors(q), orc.preprocess_vectors(x3), "ip")
    kth = np.sort(s64, axis=1)[:, -k]
    got = np.take_along_axis(s64, I, axis=1)
    assert np.all(got >= kth[:, None] - 1e-6)          # every returned row is a true top-k row up to near-ties
    assert np.max(np.abs(D - np.sort(s64, axis=1)[:, ::-1][:, :k])) <= SCORE_TOL
    idx.close()

    x2 = rng.standard_normal((n, d)).astype(np.float32)
    lo = 5 * 32                              # tiles 5..14 lie between two sampled tiles (0 and 16)
    for j in range(300):
        x2[lo + j] = q[j % nq] + 0.2 * rng.standard_normal(d).astype(np.float32)
    idx = FlatIndex(d, METRIC_IP, normalize=True)
    idx.add(x2)
    idx.set_option(OPT_SAMPLE_ROWS, sample_rows)
    _check(idx, x2, q, k, "ip", True)
    assert idx.last_stats()["n_dense_exact"] == 0
    idx.close()


def test_add_npy_streams_a_memory_mapped_file(tmp_path):
    """embeddings.npy -> index without loading the file whole: float32 and float16 files, chunk boundaries that are
    not multiples of the 32-row tile, equal to adding the array directly"""
    from anorag_hip import FlatIndex, METRIC_IP
    x, q = _data(10_007, 96, 5)
    p32 = str(tmp_path / "embeddings.npy")
    np.save(p32, x)
    a = FlatIndex(96, METRIC_IP, normalize=True)
    assert a.add_npy(p32, chunk_rows=3001) == 10_007 and a.ntotal == 10_007
    b = FlatIndex(96, METRIC_IP, normalize=True)
    b.add(x)
    Da, Ia = a.search(q, 10)
    Db, Ib = b.search(q, 10)
    assert np.array_equal(Ia, Ib) and np.array_equal(Da, Db)
    p16 = str(tmp_path / "embeddings16.npy")
    np.save(p16, x.astype(np.float16))
    c = FlatIndex(96, METRIC_IP, normalize=True)
    c.add_npy(p16)
    _check(c, x.astype(np.float16).astype(np.float32), q, 10, "ip", True)
    wrong = FlatIndex(64, METRIC_IP)
    with pytest.raises(ValueError):
        wrong.add_npy(p32)                     # dimension mismatch
    wrong.close()
    for i in (a, b, c):
        i.close()


def test_randomised_shapes_against_the_oracle():
    """a seeded sweep over corpus size, dimension, batch size, k and metric (both the dense and the sparse pipeline,
    ragged tiles, k > n padding, dimensions that are not multiples of 16) — every case against the oracle"""
    from anorag_hip import FlatIndex, METRIC_IP, METRIC_L2
    rng = np.random.default_rng(2024)
    sizes = [1, 31, 33, 1000, 4097, 50_000, 120_001]
    for case in range(14):
        n = int(sizes[case % len(sizes)])
        d = int(rng.choice([8, 17, 64, 100, 200, 384]))
        nq = int(rng.choice([1, 3, 64, 65, 130]))
        k = int(rng.choice([1, 7, 50, 200]))
        metric = "l2" if case % 3 == 2 else "ip"
        normalize = metric == "ip" and bool(case % 2)
        x = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal((nq, d)).astype(np.float32)
        idx = FlatIndex(d, METRIC_L2 if metric == "l2" else METRIC_IP, normalize=normalize)
        idx.add(x[: n // 2])
        idx.add(x[n // 2:])
        try:
            _check(idx, x, q, k, metric, normalize)
        except AssertionError as e:  # say which case
            raise AssertionError(f"case {case}: n={n} d={d} nq={nq} k={k} metric={metric} normalize={normalize}: {e}")
        idx.close()


def test_concurrent_searches_from_worker_threads():
    """the reference searches from ThreadPoolExecutor workers (query_processor.py:2761-2766): handles serialise
    their own calls, ctypes releases the GIL, every thread gets the single-threaded answer"""
    from concurrent.futures import ThreadPoolExecutor
    from anorag_hip import FlatIndex, METRIC_IP
    x, q = _data(60_000, 128, 64)
    idx = FlatIndex(128, METRIC_IP, normalize=True)
    idx.add(x)
    ref = [idx.search(q[i:i + 4], 10) for i in range(0, 64, 4)]
    with ThreadPoolExecutor(max_workers=8) as ex:
        got = list(ex.map(lambda i: idx.search(q[i:i + 4], 10), range(0, 64, 4)))
    for (Dr, Ir), (Dg, Ig) in zip(ref, got):
        assert np.array_equal(Ir, Ig) and np.array_equal(Dr, Dg)
    idx.close()


def test_k_beyond_the_select_window_uses_the_device_sort():
    """k > 1024 (faiss accepts any k; retrieve() over-fetches top_k x 3): exact scores of every row + device radix
    sort; cosine and L2, k > n padding, ties by ascending id, id offset"""
    from anorag_hip import FlatIndex, METRIC_IP, METRIC_L2
    from anorag_hip._lib import OPT_ID_OFFSET
    x, q = _data(20_011, 96, 5)
    x[100:110] = x[7]                                    # duplicate rows: equal scores, ascending ids expected
    q[0] = x[7]                                          # ... and query 0 finds them first
    idx = FlatIndex(96, METRIC_IP, normalize=True)
    idx.add(x)
    D, I, Dr, Ir = _check(idx, x, q, 3000, "ip", True)
    assert idx.last_stats()["n_dense_exact"] == 5
    assert list(I[0, :11]) == [7] + list(range(100, 110))
    idx.set_option(OPT_ID_OFFSET, 5000)
    D2, I2 = idx.search(q[:2], 2048)
    assert np.array_equal(I2, I[:2, :2048] + 5000) and np.array_equal(D2, D[:2, :2048])
    idx.close()
    l2 = FlatIndex(96, METRIC_L2, normalize=False)
    l2.add(x[:1500])
    _check(l2, x[:1500], q, 2000, "l2", False)          # k > n: -1 padding
    l2.close()


def test_candidate_list_overflow_falls_back_to_the_dense_exact_path():
    """a list capacity far too small for the data (ANR_OPT_CAND_CAP = 64 entries per workgroup and query, every row
    of a 40 000-row cluster that the sample never sees is above the threshold): the overflow is detected and the dense exact path answers, exactly"""
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip._lib import OPT_CAND_CAP
    rng = np.random.default_rng(21)
    n, d, nq, k = 200_000, 64, 6, 20
    x = rng.standard_normal((n, d)).astype(np.float32)
    c = rng.standard_normal(d).astype(np.float32)
    from anorag_hip._lib import OPT_SAMPLE_ROWS
    sample_rows = 12288                                  # 384 sample tiles -> every 16th tile is sampled
    tiles = np.arange(n // 32)
    unsampled = tiles[tiles % ((n // 32) // (sample_rows // 32)) != 0]
    hot = (rng.choice(unsampled, 1250, replace=False)[:, None] * 32 + np.arange(32)[None, :]).ravel()  # 40 000 rows
    x[hot] = c + 0.15 * rng.standard_normal((len(hot), d)).astype(np.float32)  # a cluster the sample never sees
    q = (c + 0.15 * rng.standard_normal((nq, d))).astype(np.float32)
    idx = FlatIndex(d, METRIC_IP, normalize=True)
    idx.add(x)
    idx.set_option(OPT_SAMPLE_ROWS, sample_rows)
    idx.set_option(OPT_CAND_CAP, 64)
    _check(idx, x, q, k, "ip", True)
    st = idx.last_stats()
    assert st["n_overflow"] > 0 and st["n_dense_exact"] > 0, st
    idx.close()


def test_tiny_corpus_single_launch_path_equals_the_pipeline_and_the_oracle():
    """C1's regime (a few queries, host buffers, a corpus of up to ~350 MB): ONE kernel per call (tiny_kernels.hpp).  Same
    bits as the five-kernel pipeline (ANR_OPT_TINY = 0) and the oracle's ids — cosine and L2, a dimension that is not
    a multiple of 4, k larger than the corpus, an id offset, 1 to 4 queries, repeated calls (completion words)."""
    from anorag_hip import FlatIndex, METRIC_IP, METRIC_L2
    from anorag_hip._lib import OPT_ID_OFFSET, OPT_TINY
    for (n, d, k, metric, name, norm) in ((10_000, 384, 10, METRIC_IP, "ip", True), (7_777, 130, 100, METRIC_L2, "l2", False),
                                          (37, 64, 50, METRIC_IP, "ip", True), (16_001, 96, 128, METRIC_IP, "ip", False),
                                          (80_000, 768, 10, METRIC_IP, "ip", True)):  # > 64 K rows: 204 workgroups
        x, q = _data(n, d, 4, seed=n, qseed=n + 1)
        idx = FlatIndex(d, metric, normalize=norm)
        idx.add(x)
        idx.set_option(OPT_TINY, 2)   # wherever the path is able, also where the pipeline is measured faster (k > 64)
        for nq in (1, 3, 4):
            for rep in range(3):
                D, I = idx.search(q[:nq], k)
                st = idx.last_stats()
                assert st["n_dense_exact"] == nq and st["n_candidates"] == 0, st   # the single-launch path ran
            idx.set_option(OPT_TINY, 0)
            D0, I0 = idx.search(q[:nq], k)
            idx.set_option(OPT_TINY, 2)
            assert np.array_equal(I, I0) and np.array_equal(D, D0)
            _check(idx, x, q[:nq], k, name, norm)
        idx.set_option(OPT_ID_OFFSET, 5_000_000)
        D1, I1 = idx.search(q[:1], k)
        assert np.array_equal(np.where(I1 >= 0, I1 - 5_000_000, -1), I[:1]) and np.array_equal(D1, D[:1])
        idx.close()
    # beyond the path's limits the pipeline answers (5 queries; more rows per workgroup than it ranks: 30 000 rows at
    # k = 100 are 20 partial lists of 1500 rows)
    x, q = _data(30_000, 384, 5)
    idx = FlatIndex(384, METRIC_IP, normalize=True)
    idx.add(x)
    idx.search(q, 10)
    assert idx.last_stats()["n_dense_exact"] == 0
    idx.search(q[:1], 100)
    assert idx.last_stats()["n_dense_exact"] == 0
    idx.search(q[:1], 10)
    assert idx.last_stats()["n_dense_exact"] == 1   # (46 MB at k = 10 is inside them)
    idx.search(q[:1], 80)                           # (default mode: k > 64 goes to the pipeline, measured faster there)
    assert idx.last_stats()["n_dense_exact"] == 0
    idx.close()


def test_fused_post_kernel_equals_the_three_launch_pipeline():
    """k_post (select + exact re-score + finalize in one launch) against the separate k_select / k_rescore / k_finalize
    launches (ANR_OPT_FUSED_POST = 0): identical bits — threshold-gated scan, dense small-corpus scan, L2, and tight
    clusters whose certificates fail (so flags / theta feed the recovery pass)"""
    from anorag_hip import FlatIndex, METRIC_IP, METRIC_L2
    from anorag_hip._lib import OPT_FUSED_POST, OPT_TINY
    rng = np.random.default_rng(3)
    cent = rng.standard_normal((200, 256)).astype(np.float32)
    cl = (cent[rng.integers(0, 200, 120_000)] + 0.02 * rng.standard_normal((120_000, 256))).astype(np.float32)
    clq = (cent[rng.integers(0, 200, 64)] + 0.02 * rng.standard_normal((64, 256))).astype(np.float32)
    cases = [(_data(150_000, 768, 64), METRIC_IP, True, 100), (_data(3_000, 384, 17), METRIC_IP, True, 10),
             (_data(90_000, 130, 33), METRIC_L2, False, 50), (_data(40_000, 768, 3), METRIC_IP, True, 20),
             ((cl, clq), METRIC_IP, True, 50)]
    for (x, q), metric, norm, k in cases:
        res = []
        for fused in (2, 0):   # 2 = k_post for every batch (the default, 1, leaves batches of <= 4 queries to the launches)
            idx = FlatIndex(x.shape[1], metric, normalize=norm)
            idx.set_option(OPT_TINY, 0)
            idx.set_option(OPT_FUSED_POST, fused)
            idx.add(x)
            D, I = idx.search(q, k)
            res.append((D, I, idx.last_stats()))
            idx.close()
        assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0], res[1][0])
        for key in ("n_fallback", "n_candidates", "n_from_lists", "n_dense_exact"):
            assert res[0][2][key] == res[1][2][key], (key, res[0][2], res[1][2])
    assert res[0][2]["n_fallback"] > 0     # the clustered case exercised the failed-certificate hand-off


def test_shadow_side_kernels_and_role_streams_equal_the_wide_pipeline():
    """The shadow-sized side kernels (k_sample, k_select_shadow, k_rescore_shadow + k_finalize: what a pipelined caller's
    batches use so that they run BESIDE the previous batch's resident scan) and the role-stream schedule against the wide
    kernels on one stream per batch: identical bits and identical statistics — IP at 768-d, a dimension that is not a
    multiple of four, L2 (row bias in the sample), a partial batch, and tight clusters whose certificates fail."""
    from anorag_hip import FlatIndex, METRIC_IP, METRIC_L2
    from anorag_hip._lib import OPT_SHADOW, OPT_SCHEDULE, OPT_TINY, OPT_FUSED_POST
    rng = np.random.default_rng(5)
    cent = rng.standard_normal((300, 256)).astype(np.float32)
    cl = (cent[rng.integers(0, 300, 260_000)] + 0.02 * rng.standard_normal((260_000, 256))).astype(np.float32)
    clq = (cent[rng.integers(0, 300, 64)] + 0.02 * rng.standard_normal((64, 256))).astype(np.float32)
    cases = [(_data(300_000, 768, 128), METRIC_IP, True, 100), (_data(280_000, 102, 70), METRIC_IP, True, 10),
             (_data(270_000, 384, 64), METRIC_L2, False, 50), (_data(260_000, 130, 5), METRIC_L2, False, 20),
             ((cl, clq), METRIC_IP, True, 50)]
    for (x, q), metric, norm, k in cases:
        res = []
        for shadow, schedule, fused in ((0, 0, 0), (2, 1, 0), (2, 0, 1), (0, 1, 2)):
            idx = FlatIndex(x.shape[1], metric, normalize=norm)
            idx.set_option(OPT_TINY, 0)
            idx.set_option(OPT_SHADOW, shadow)
            idx.set_option(OPT_SCHEDULE, schedule)
            idx.set_option(OPT_FUSED_POST, fused)
            idx.add(x)
            D, I = idx.search(q, k)
            st = idx.last_stats()
            assert st["sample_rows"] > 0, "the case must take the threshold-gated scan"
            rec, _ = idx.batch_log(8, correlate=False)
            assert rec.shape[0] >= 1 and bool(rec[-1, 12] & 1) == (shadow == 2)
            res.append((D, I, st))
            idx.close()
        for other in res[1:]:
            assert np.array_equal(res[0][1], other[1]) and np.array_equal(res[0][0], other[0])
            for key in ("n_fallback", "n_candidates", "n_from_lists", "n_dense_exact", "overfetch"):
                assert res[0][2][key] == other[2][key], (key, res[0][2], other[2])
    assert res[0][2]["n_fallback"] > 0     # the clustered case exercised the failed-certificate hand-off
    # and the results are the oracle's
    x, q = _data(300_000, 768, 16)
    idx = FlatIndex(768, METRIC_IP, normalize=True)
    idx.set_option(OPT_SHADOW, 2)
    idx.add(x)
    _check(idx, x, q, 100, "ip", True)
    idx.close()


def test_pipelined_batches_with_failed_certificates_recover_beside_the_pipeline():
    """Asynchronous batches on a corpus of tight clusters (every certificate fails; some queries need the fixed-threshold second
    scan): the recovery of a completed batch is launched while later batches are still being enqueued (round 4:
    launch_recovery / finish_recovery) and must leave exactly the results of the one-batch-at-a-time search — and the
    oracle's."""
    import torch
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip._lib import OPT_OVERFETCH
    rng = np.random.default_rng(12)
    n, d, B, k, nb = 400_000, 256, 64, 50, 9
    cent = rng.standard_normal((400, d)).astype(np.float32)
    x = (cent[rng.integers(0, 400, n)] + 0.02 * rng.standard_normal((n, d))).astype(np.float32)
    q = (cent[rng.integers(0, 400, (nb, B))] + 0.02 * rng.standard_normal((nb, B, d))).astype(np.float32)
    q[:, ::7] = rng.standard_normal((nb, len(range(0, B, 7)), d)).astype(np.float32)   # some queries far from every cluster
    idx = FlatIndex(d, METRIC_IP, normalize=True)
    idx.set_option(OPT_OVERFETCH, 64)    # a fixed, small K': certificates keep failing in every batch (no adaptation)
    idx.add(x)
    dev = torch.device("cuda", 0)
    Q = torch.from_numpy(q).to(dev)
    ref = []
    for b in range(nb):
        D = torch.empty((B, k), device=dev)
        I = torch.empty((B, k), device=dev, dtype=torch.int64)
        idx.search_device(Q[b].data_ptr(), B, k, D.data_ptr(), I.data_ptr())
        torch.cuda.synchronize()
        ref.append((D.cpu().numpy(), I.cpu().numpy()))
    st0 = idx.last_stats()
    assert st0["n_fallback"] > 0
    idx.reset_stats()
    S = [torch.cuda.Stream() for _ in range(3)]
    Ds = [torch.empty((B, k), device=dev) for _ in range(nb)]
    Is = [torch.empty((B, k), device=dev, dtype=torch.int64) for _ in range(nb)]
    for rep in range(2):
        for b in range(nb):
            idx.search_device_async(Q[b].data_ptr(), B, k, Ds[b].data_ptr(), Is[b].data_ptr(), S[b % 3].cuda_stream)
        idx.sync()
        torch.cuda.synchronize()
        for b in range(nb):
            assert np.array_equal(Is[b].cpu().numpy(), ref[b][1]) and np.array_equal(Ds[b].cpu().numpy(), ref[b][0]), (rep, b)
    st = idx.last_stats()
    assert st["n_fallback"] >= nb * 10 and st["n_dense_exact"] == 0, st
    assert st["n_from_lists"] > 0
    rec, _ = idx.batch_log(2 * nb, correlate=False)
    assert rec.shape[0] == 2 * nb
    xn = orc.preprocess_vectors(x)
    qn = orc.preprocess_vectors(q[nb - 1])
    Dr, Ir = orc.flat_search(qn, xn, k, "ip")
    assert orc.near_tie_equal(ref[nb - 1][1], Ir, orc.exact_scores(qn, xn, "ip"), k, 1e-6)
    idx.close()


@pytest.mark.parametrize("when", ["before_add", "after_add"])
def test_twelve_bit_scan_image_gives_the_exact_top_k(when):
    """ANR_OPT_SCAN_BITS 12: the streaming pass reads every stored f16 rounded to its top 12 bits (25 % fewer bytes); the
    certificate takes the coarser image's error norm, so the answers stay the exact top-k of the float32 rows — Gaussian
    rows, a ragged row count, cosine / raw inner product / L2, the option set before the rows arrive (k_add writes both
    images) and on a filled index (k_build12), incremental adds after the switch, and tight clusters whose certificates
    fail (recovery from the lists, theta re-scan)."""
    from anorag_hip import FlatIndex, METRIC_IP, METRIC_L2
    from anorag_hip._lib import OPT_SCAN_BITS
    n, d, nq, k = 200_001, 768, 64, 100
    x, q = _data(n, d, nq, seed=77, qseed=78)
    for metric, name, normalize in ((METRIC_IP, "ip", True), (METRIC_IP, "ip", False), (METRIC_L2, "l2", False)):
        idx = FlatIndex(d, metric, normalize=normalize)
        if when == "before_add":
            idx.set_option(OPT_SCAN_BITS, 12)
        idx.add(x[:150_000])
        if when == "after_add":
            idx.set_option(OPT_SCAN_BITS, 12)
        idx.add(x[150_000:])                       # rows appended after the switch land in both images
        D, I, Dr, Ir = _check(idx, x, q, k, name, normalize)
        st = idx.last_stats()
        assert st["sample_rows"] > 0 and st["n_dense_exact"] == 0, st
        tiles = -(-n // 32)
        assert st["scan_bytes"] == tiles * 32 * 768 * 3 // 2, st      # 1.5 bytes per stored value
        if name == "ip" and normalize:
            idx.set_option(OPT_SCAN_BITS, 16)      # and back: the f16 image, the same answers
            D16, I16 = idx.search(q, k)
            assert np.array_equal(I16, I) and np.array_equal(D16, D)
            assert idx.last_stats()["scan_bytes"] == tiles * 32 * 768 * 2
        idx.close()
    rng = np.random.default_rng(11)
    n, d, nq, k = 150_000, 256, 64, 50
    cent = rng.standard_normal((300, d)).astype(np.float32)
    x = (cent[rng.integers(0, 300, n)] + 0.02 * rng.standard_normal((n, d))).astype(np.float32)
    q = (cent[rng.integers(0, 300, nq)] + 0.02 * rng.standard_normal((nq, d))).astype(np.float32)
    idx = FlatIndex(d, METRIC_IP, normalize=True)
    if when == "before_add":
        idx.set_option(OPT_SCAN_BITS, 12)
    idx.add(x)
    idx.set_option(OPT_SCAN_BITS, 12)
    # near-duplicate neighbourhoods: hundreds of rows within the 12-bit error bound of the k-th score — exact answers from the
    # first batch on (lists, theta re-scan, dense exact path); K' climbs its ladder, and after two batches that failed at the
    # largest K' the index goes back to the f16 image by itself (whose own K' then adapts as in test_tight_clusters_...)
    seen = []
    for _ in range(14):
        _check(idx, x, q, k, "ip", True)
        st = idx.last_stats()
        seen.append((st["scan_bytes"], st["n_fallback"], st["n_dense_exact"], st["overfetch"], st["n_from_lists"]))
    tiles = -(-n // 32)
    b12, b16 = tiles * 32 * 256 * 3 // 2, tiles * 32 * 256 * 2
    # (either outcome is sound: the ladder finds a K' at which most certificates hold on the 12-bit image, or two batches
    # fail at the top level and the f16 image takes over; what must hold is that the failures die down)
    assert seen[0][0] == b12 and seen[-1][0] in (b12, b16), seen
    assert seen[-1][1] < nq // 2, [t[1:] for t in seen]   # as the f16 image does on this data at its largest K' (test_tight_clusters_...)
    assert any(a[1] > b[1] for a, b in zip(seen, seen[1:])), seen
    idx.reset()                                    # an emptied index keeps the option (and tries the 12-bit image again)
    idx.add(x[:40_000])
    _check(idx, x[:40_000], q, k, "ip", True)
    idx.close()


def test_twelve_bit_scan_pipelined_batches_equal_the_sixteen_bit_ones():
    """the asynchronous pipeline with the 12-bit image: three batches in flight, results identical to the f16 image's"""
    import torch
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip._lib import OPT_SCAN_BITS
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    n, d, nb, B, k = 400_000, 768, 9, 64, 100
    idx = FlatIndex(d, METRIC_IP, normalize=True)
    idx.reserve(n)
    for _ in range(4):
        xb = torch.randn((n // 4, d), generator=g, device=dev)
        torch.cuda.synchronize()
        idx.add_device(xb.data_ptr(), xb.shape[0])
    Q = torch.randn((nb, B, d), generator=g, device=dev)
    out = {}
    for bits in (16, 12):
        idx.set_option(OPT_SCAN_BITS, bits)
        D = torch.empty((nb, B, k), device=dev)
        I = torch.empty((nb, B, k), device=dev, dtype=torch.int64)
        torch.cuda.synchronize()
        for b in range(nb):
            idx.search_device_async(Q[b].data_ptr(), B, k, D[b].data_ptr(), I[b].data_ptr(), 0)
        idx.sync()
        torch.cuda.synchronize()
        out[bits] = (D.cpu().numpy(), I.cpu().numpy())
    assert np.array_equal(out[12][1], out[16][1])
    assert np.array_equal(out[12][0], out[16][0])      # the scores are the exact re-scored ones either way
    idx.close()


@pytest.mark.parametrize("d,n,nq,k", [(100, 120_000, 7, 10), (384, 150_000, 33, 20), (1024, 70_000, 64, 5)])
def test_twelve_bit_scan_at_other_widths(d, n, nq, k):
    """the 12-bit loader's other instantiations: padded dimensions (100 -> 128), the four-block operand buffers (384: 24
    k-blocks), a wide row (1024); forced with ANR_OPT_SCAN_BITS 12 on corpora below the automatic threshold"""
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip._lib import OPT_SCAN_BITS
    x, q = _data(n, d, nq, seed=d, qseed=d + 1)
    idx = FlatIndex(d, METRIC_IP, normalize=True)
    idx.set_option(OPT_SCAN_BITS, 12)
    idx.add(x)
    _check(idx, x, q, k, "ip", True)
    st = idx.last_stats()
    dimp = -(-d // 128) * 128
    assert st["sample_rows"] > 0, st                                    # the threshold-gated pipeline, not the small-corpus path
    assert st["scan_bytes"] == -(-n // 32) * 32 * dimp * 3 // 2, st     # ... reading the 12-bit image
    idx.close()


def test_twelve_bit_scan_on_low_rank_embeddings():
    """rows that live near a 48-dimensional subspace with a few dominant directions (what sentence embeddings look like more
    than an isotropic Gaussian does): scores bunch up near the top, the 12-bit batches climb their K' ladder or hand over to
    the f16 image — and every answer is the exact top-k"""
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip._lib import OPT_SCAN_BITS
    rng = np.random.default_rng(5)
    n, d, nq, k = 180_000, 768, 64, 100
    basis = rng.standard_normal((48, d)).astype(np.float32) * (1.0 / np.sqrt(np.arange(1, 49)))[:, None].astype(np.float32)
    x = (rng.standard_normal((n, 48)).astype(np.float32) @ basis + 0.05 * rng.standard_normal((n, d)).astype(np.float32))
    q = (rng.standard_normal((nq, 48)).astype(np.float32) @ basis + 0.05 * rng.standard_normal((nq, d)).astype(np.float32))
    idx = FlatIndex(d, METRIC_IP, normalize=True)
    idx.set_option(OPT_SCAN_BITS, 12)
    idx.add(x)
    seen = []
    for _ in range(6):
        _check(idx, x, q, k, "ip", True)
        st = idx.last_stats()
        seen.append((st["scan_bytes"] * 2 // (-(-n // 32) * 32 * d), st["n_fallback"], st["overfetch"]))
    assert seen[0][0] == 3, seen                    # started on the 12-bit image (3 half-bytes... 1.5 bytes per value)
    assert seen[-1][1] <= seen[0][1], seen          # the failures do not grow
    idx.close()


@pytest.mark.parametrize("d,normalize", [(768, True), (100, False)])
def test_the_scan_images_and_their_error_norms_equal_the_oracles(d, normalize):
    """what the certificate of a 12-bit batch rests on: the device's 12-bit image is, bit for bit, the oracle's restatement
    (float16 of the stored row, rounded to nearest-even to its top 12 bits), and the error norm the index tracks for it
    bounds the true one from above by at most the 1e-4 slack of its float32 accumulation — rows written by k_add (before the
    switch: k_build12 converts them) and after it, large values near the float16 range included"""
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip._lib import AnoragError, OPT_SCAN_BITS
    rng = np.random.default_rng(3)
    n = 6_000
    x = rng.standard_normal((n, d)).astype(np.float32)
    if not normalize:
        x[:50] *= 1.0e3                      # values up to a few thousand
        x[50, :8] = [65504.0, -65504.0, 65500.0, 65472.0, -65488.0, 6.1e-5, -5.9e-8, 0.0]   # the float16 edges
    idx = FlatIndex(d, METRIC_IP, normalize=normalize)
    with pytest.raises(AnoragError):
        idx.add(x[:10]) or idx.reconstruct_scan_image(12, 0, 10)     # no 12-bit image yet
    idx.add(x[10:3000])
    idx.set_option(OPT_SCAN_BITS, 12)        # k_build12 for the 3000 rows stored so far
    idx.add(x[3000:])                        # k_add for the rest
    stored = idx.reconstruct_n(0, n)         # the float32 rows the index keeps (normalised on the device)
    for bits in (16, 12):
        img = idx.reconstruct_scan_image(bits, 0, n)
        ref = orc.scan_image(stored, bits)
        assert img.tobytes() == ref.tobytes(), bits
    st = idx.scan_image_stats()
    assert st["bits"] == 12
    for bits, key in ((16, "max_err16"), (12, "max_err12")):
        true = orc.scan_image_error(stored, bits)
        assert true <= st[key] <= true * 1.0003 + 1e-12, (bits, true, st[key])
    norm = float(np.sqrt((stored.astype(np.float64) ** 2).sum(axis=1)).max())
    assert norm <= st["max_norm"] <= norm * 1.0003
    part = idx.reconstruct_scan_image(12, 2990, 37)    # a range that straddles tiles and the two writers
    assert part.tobytes() == orc.scan_image(stored[2990:3027], 12).tobytes()
    idx.close()
