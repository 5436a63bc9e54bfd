"""GPU parity of the row-sharded path (BASELINE.json config C3) through the C ABI, on ONE device:
P handles stand in for the P GPUs of a node — the data path (ANR_OPT_ID_OFFSET, packed [B*k f32 | B*k i64] result
buffers laid out as one all-gather leaves them, anr_index_wait before the exchange, anr_merge_topk_strided_dev) is
exactly bench.py's; only the RCCL transport is absent.  Checked against the float64-arbitrated oracle."""
import ctypes as C

import numpy as np
import pytest

from oracle import flat_index as orc

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-4  # BASELINE.json north_star: scores within 1e-4 fp32


def _gen(rows, dim, seed, dev, chunk=262_144, centroids=None, sigma=0.0):
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    done = 0
    while done < rows:
        m = min(chunk, rows - done)
        x = torch.randn((m, dim), generator=g, device=dev, dtype=torch.float32)
        if centroids is not None:
            pick = torch.randint(0, centroids.shape[0], (m,), generator=g, device=dev)
            x = centroids[pick] + sigma * x
        yield x
        done += m


def _build_shards(P, per, dim, dev, seed0, **kw):
    import torch
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip._lib import OPT_ID_OFFSET
    shards = []
    for p in range(P):
        idx = FlatIndex(dim, METRIC_IP, normalize=True)
        idx.reserve(per)
        for xb in _gen(per, dim, seed0 + p, dev, **kw):
            torch.cuda.synchronize()
            idx.add_device(xb.data_ptr(), xb.shape[0])
        idx.set_option(OPT_ID_OFFSET, p * per)
        shards.append(idx)
    return shards


def _oracle(P, per, dim, dev, seed0, q_host, k, **kw):
    """(exact float64 scores, ids) of the best k + 16 rows per query: the extra rows let near-ties be checked"""
    top = orc.BlockedTopK(orc.preprocess_vectors(q_host), k + 16, "ip")
    for p in range(P):
        base = p * per
        for xb in _gen(per, dim, seed0 + p, dev, chunk=524_288, **kw):
            top.push(orc.preprocess_vectors(xb.cpu().numpy()), base)
            base += xb.shape[0]
    return top.result()


def _pipeline(shards, Q, B, k, dev):
    """bench.py's N > 1 step for every batch of Q [nb, B, dim]: async searches into the packed buffers of a slot,
    anr_index_wait (final) two batches behind, strided merge; returns merged (D, I) per batch and the stats"""
    import torch
    from anorag_hip import _lib
    lib = _lib.load()
    from anorag_hip.sharded import packed_layout
    P, nres, NSLOT = len(shards), B * k, 3
    id_off, part = packed_layout(nres)
    packed = [torch.zeros(P * part, device=dev, dtype=torch.uint8) for _ in range(NSLOT)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NSLOT)]
    out, pending = {}, []

    def exchange(i):
        s = i % NSLOT
        for sh in shards:
            sh.wait(len(pending))
        Dm = torch.empty((B, k), device=dev)
        Im = torch.empty((B, k), device=dev, dtype=torch.int64)
        with torch.cuda.stream(streams[s]):
            _lib.check(lib.anr_merge_topk_strided_dev(
                0, C.c_void_p(packed[s].data_ptr()), C.c_void_p(packed[s].data_ptr() + id_off), part // 4,
                part // 8, P, B, k, 1, C.c_void_p(Dm.data_ptr()), C.c_void_p(Im.data_ptr()),
                C.c_void_p(streams[s].cuda_stream)), "merge")
        out[i] = (Dm, Im)

    for i in range(Q.shape[0]):
        s = i % NSLOT
        for r, sh in enumerate(shards):
            base = packed[s].data_ptr() + r * part
            sh.search_device_async(Q[i].data_ptr(), B, k, base, base + id_off, streams[s].cuda_stream)
        pending.append(i)
        if len(pending) > NSLOT - 1:
            exchange(pending.pop(0))
    for sh in shards:
        sh.sync()
    while pending:
        exchange(pending.pop(0))
    torch.cuda.synchronize()
    return out


def _same_ids(I_gpu, S_ref, I_ref, k, tol=1e-6):
    """identical id sets, modulo rows within `tol` of the k-th exact score (no float32 reference resolves those)"""
    for r in range(I_ref.shape[0]):
        a, b = set(I_gpu[r].tolist()), set(I_ref[r, :k].tolist())
        if a == b:
            continue
        kth = S_ref[r, k - 1]
        sc = dict(zip(I_ref[r].tolist(), S_ref[r].tolist()))
        for rid in a ^ b:
            if rid in sc and abs(sc[rid] - kth) > tol:
                return False
            if rid not in sc:  # a row the oracle ranks below its own list end: it must tie with the k-th
                return False
    return True


def test_c3_full_shape_eight_shards_10m_rows():
    """C3 at its real shape: 10 M x 768 as 8 shards of 1.25 M rows, batch 64, top-100, five pipelined batches — all
    64 queries of the last batch against the oracle (ids and scores), one batch against the synchronous host
    path; no certificate is expected to fail on Gaussian rows."""
    import torch
    dev = torch.device("cuda", 0)
    P, per, dim, B, k = 8, 1_250_000, 768, 64, 100
    shards = _build_shards(P, per, dim, dev, 5000)
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    Q = torch.randn((5, B, dim), generator=g, device=dev)
    out = _pipeline(shards, Q, B, k, dev)
    fb = sum(sh.last_stats()["n_fallback"] for sh in shards)
    S_ref, I_ref = _oracle(P, per, dim, dev, 5000, Q[4].cpu().numpy(), k)
    Dm, Im = out[4]
    assert _same_ids(Im.cpu().numpy(), S_ref, I_ref, k)
    assert np.max(np.abs(Dm.cpu().numpy() - S_ref[:, :k].astype(np.float32))) <= SCORE_TOL
    assert (Im.cpu().numpy() >= 0).all() and (Im.cpu().numpy() < P * per).all()
    # the pipelined answers equal the synchronous single-batch path on every shard + host merge
    from anorag_hip.sharded import merge_topk_host_c
    qh = Q[2].cpu().numpy()
    parts = [sh.search(qh, k) for sh in shards]
    Dh, Ih = merge_topk_host_c(np.stack([d for d, _ in parts]), np.stack([i for _, i in parts]), True)
    assert np.array_equal(Ih, out[2][1].cpu().numpy()) and np.array_equal(Dh, out[2][0].cpu().numpy())
    assert fb == 0
    for sh in shards:
        sh.close()


def test_clustered_shards_failed_certificates_are_final_before_the_merge():
    """Dense neighbourhoods (tight clusters: hundreds of rows within the f16 error bound of the k-th score) make
    certificates fail; the exchange must only see a shard's batch after anr_index_wait ran its recovery.  The merged
    lists of every pipelined batch equal the oracle's."""
    import torch
    dev = torch.device("cuda", 0)
    P, per, dim, B, k = 8, 160_000, 768, 64, 100
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    cent = torch.randn((320, dim), generator=g, device=dev)  # ~500 rows per cluster in every shard: more than K'
    kw = dict(centroids=cent, sigma=0.02)
    shards = _build_shards(P, per, dim, dev, 7000, **kw)
    Q = cent[torch.randint(0, 320, (4, B), generator=g, device=dev)] + 0.02 * torch.randn((4, B, dim), generator=g, device=dev)
    out = _pipeline(shards, Q, B, k, dev)
    stats = [sh.last_stats() for sh in shards]
    assert sum(s["n_fallback"] for s in stats) > 0, stats  # the case this test exists for
    for i in (1, 3):
        S_ref, I_ref = _oracle(P, per, dim, dev, 7000, Q[i].cpu().numpy(), k, **kw)
        Dm, Im = out[i]
        assert _same_ids(Im.cpu().numpy(), S_ref, I_ref, k, tol=2e-6)
        assert np.max(np.abs(Dm.cpu().numpy() - S_ref[:, :k].astype(np.float32))) <= SCORE_TOL
    for sh in shards:
        sh.close()


def test_sharded_flat_index_single_process_and_vector_index_devices_key(tmp_path):
    """ShardedFlatIndex (one process, several handles): two add calls (several segments per shard -> host id
    mapping), search / reconstruct / score_rows vs the oracle; then the drop-in VectorIndex with
    anorag_hip.devices = [0, 0, 0] incl. save / load."""
    from anorag_hip import METRIC_IP
    from anorag_hip.compat import config
    from anorag_hip.sharded import ShardedFlatIndex
    n, d, nq, k = 30_011, 96, 7, 25
    x = np.random.default_rng(1234).standard_normal((n, d), dtype=np.float32)
    x[20_000] = x[3]  # duplicate rows in different shards: the tie goes to the lower global id
    q = np.random.default_rng(4321).standard_normal((nq, d), dtype=np.float32)
    xn, qn = orc.preprocess_vectors(x), orc.preprocess_vectors(q)
    Dr, Ir = orc.flat_search(qn, xn, k, "ip")
    s64 = orc.exact_scores(qn, xn, "ip")
    sh = ShardedFlatIndex(d, METRIC_IP, normalize=True, devices=[0, 0, 0])
    sh.add(x)
    D, I = sh.search(q, k)
    assert orc.near_tie_equal(I, Ir, s64, k, 1e-6) and np.max(np.abs(D - Dr)) <= SCORE_TOL
    sh.reset()
    sh.add(x[:12_345])
    sh.add(x[12_345:])
    assert sh.ntotal == n
    D, I = sh.search(q, k)
    assert orc.near_tie_equal(I, Ir, s64, k, 1e-6) and np.max(np.abs(D - Dr)) <= SCORE_TOL
    rows = sh.reconstruct_n(12_000, 700)
    assert np.max(np.abs(rows - xn[12_000:12_700])) <= 1e-6
    ids = np.array([[0, 12_345, n - 1, -1]] * nq, dtype=np.int64)
    sc = sh.score_rows(q, ids)
    assert np.allclose(sc[:, :3], s64[:, [0, 12_345, n - 1]], atol=SCORE_TOL) and np.isnan(sc[:, 3]).all()
    sh.close()

    from vector_store.vector_index import VectorIndex
    old = config.get("anorag_hip.devices", None)
    config.set("anorag_hip.devices", [0, 0, 0])
    config.set("storage.vector_index_path", str(tmp_path))
    try:
        vi = VectorIndex(embedding_dim=d)
        assert vi.create_index("Flat") and vi.add_vectors(x)
        assert type(vi.index).__name__ == "ShardedFlatIndex"
        got = vi.search(q, top_k=k)
        assert [[h["index"] for h in r] for r in got] == I.tolist()
        path = vi.save_index()
        vj = VectorIndex(embedding_dim=d)
        assert path and vj.load_index(path.split("/")[-1])
        assert vj.search(q, top_k=k) == got
        vi.cleanup()
        vj.cleanup()
    finally:
        config.set("anorag_hip.devices", old)


def test_sharded_searcher_device_path_under_an_nccl_group_of_one():
    """ShardedSearcher's RCCL leg (packed [B*k f32 | pad | B*k i64] buffer, anr_index_wait, all_gather_into_tensor,
    anr_merge_topk_strided_dev out of the receive buffer) executed under an initialised nccl process group of world
    size 1 — the most one GPU allows; odd B*k included (the id block must stay 8-byte aligned, the part stride exact).
    Results == the oracle with the shard's id offset applied."""
    import socket
    import torch
    import torch.distributed as dist
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip.sharded import ShardedSearcher
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n, d, off = 30_000, 96, 1_000_000
        x = np.random.default_rng(5).standard_normal((n, d), dtype=np.float32)
        idx = FlatIndex(d, METRIC_IP, normalize=True)
        idx.add(x)
        srch = ShardedSearcher(idx, off, force_device=True)
        assert srch.backend == "nccl" and srch.world == 1
        xn = orc.preprocess_vectors(x)
        for B, k in ((1, 5), (3, 7), (64, 100), (5, 1)):
            q = np.random.default_rng(100 + B).standard_normal((B, d), dtype=np.float32)
            D, I = srch.search(q, k)
            qn = orc.preprocess_vectors(q)
            Dr, Ir = orc.flat_search(qn, xn, k, "ip")
            s64 = orc.exact_scores(qn, xn, "ip")
            assert orc.near_tie_equal(I - off, Ir, s64, k, 1e-6), (B, k)
            assert np.max(np.abs(D - Dr)) <= SCORE_TOL
        # the pipelined form (bench.py's N > 1 loop): searches left in flight, two and three batches per all-gather, a
        # short last group; device tensors out
        for B, k, group, n_batches in ((64, 100, 2, 9), (3, 7, 3, 7), (5, 1, 1, 4)):
            stream = srch.stream(B, k, lag=2, group=group)
            qs = np.random.default_rng(200 + B).standard_normal((n_batches, B, d), dtype=np.float32)
            qd = torch.from_numpy(qs).cuda()
            got = {}
            for b in range(n_batches):
                for tag, Dt, It in stream.submit(qd[b] if b % 2 else qs[b], tag=b):   # device tensors and numpy arrays
                    got[tag] = (Dt.cpu().numpy(), It.cpu().numpy())
            idx.sync()
            for tag, Dt, It in stream.flush():
                got[tag] = (Dt.cpu().numpy(), It.cpu().numpy())
            assert sorted(got) == list(range(n_batches))
            for b in range(n_batches):
                qn = orc.preprocess_vectors(qs[b])
                Dr, Ir = orc.flat_search(qn, xn, k, "ip")
                s64 = orc.exact_scores(qn, xn, "ip")
                assert orc.near_tie_equal(got[b][1] - off, Ir, s64, k, 1e-6), (B, k, b)
                assert np.max(np.abs(got[b][0] - Dr)) <= SCORE_TOL
        # queries PRODUCED by un-synchronised work on the caller's stream: a long matrix product sits in front of
        # the copy that fills q, submit() is called right away — the search must wait for the producer, not read the zeros
        stream = srch.stream(64, 10, lag=2, group=2)
        qs = np.random.default_rng(77).standard_normal((6, 64, d), dtype=np.float32)
        src = torch.from_numpy(qs).cuda()
        big = torch.randn((6144, 6144), device="cuda")
        torch.cuda.synchronize()
        got = {}
        for b in range(6):
            qd = torch.zeros((64, d), device="cuda")
            for _ in range(3):
                big = (big @ big) * 1e-4      # milliseconds of work in front of ...
            qd.copy_(src[b])                   # ... the kernel that writes the queries
            for tag, Dt, It in stream.submit(qd, tag=b):
                got[tag] = (Dt.cpu().numpy(), It.cpu().numpy())
            del qd                            # the allocator may hand the block out again: record_stream keeps it safe
        idx.sync()
        for tag, Dt, It in stream.flush():
            got[tag] = (Dt.cpu().numpy(), It.cpu().numpy())
        for b in range(6):
            qn = orc.preprocess_vectors(qs[b])
            Dr, Ir = orc.flat_search(qn, xn, 10, "ip")
            assert orc.near_tie_equal(got[b][1] - off, Ir, orc.exact_scores(qn, xn, "ip"), 10, 1e-6), b
        idx.close()
    finally:
        dist.destroy_process_group()
