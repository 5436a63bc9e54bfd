"""GPU tests of the drop-in classes (reference API surface): VectorIndex result shaping and persistence,
EmbeddingManager text handling, VectorRetriever build/search/retrieve against the oracle pipeline."""
import os

import numpy as np
import pytest

from oracle import encoder as oenc
from oracle import flat_index as orc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def cfg(tmp_path):
    from anorag_hip.compat import config
    config.reset()
    config.set("storage.vector_index_path", str(tmp_path / "vidx"))
    config.set("storage.vector_store_path", str(tmp_path / "vstore"))
    config.set("storage.embedding_cache_path", str(tmp_path / "ecache"))
    config.set("vector_store.index_type", "Flat")
    yield config
    config.reset()


def test_vector_index_surface(cfg, tmp_path):
    from vector_store import VectorIndex
    x = np.random.default_rng(0).standard_normal((500, 96)).astype(np.float64)   # float64 in, cast like the reference
    q = np.random.default_rng(1).standard_normal((3, 96)).astype(np.float32)
    vi = VectorIndex(96)
    assert vi.search(q) == []                     # no index yet
    assert vi.create_index() is True and vi.is_trained
    assert vi.search(q) == []                     # empty index
    assert vi.add_vectors(x, np.arange(500)) is True and vi.total_vectors == 500
    res = vi.search(q, top_k=7)
    xn, qn = orc.preprocess_vectors(x), orc.preprocess_vectors(q)
    Dr, Ir = orc.flat_search(qn, xn, 7, "ip")
    exp = orc.shape_results(Dr, Ir, "cosine")
    assert [[h["index"] for h in r] for r in res] == [[h["index"] for h in r] for r in exp]
    for r, e in zip(res, exp):
        for a, b in zip(r, e):
            assert set(a) == {"index", "score", "rank", "similarity"}
            assert a["rank"] == b["rank"] and abs(a["score"] - b["score"]) <= 1e-4 and a["similarity"] == a["score"]
    single = vi.search(q[:1], top_k=3)            # one query -> flat list (vector_index.py:255-257)
    assert isinstance(single[0], dict) and len(single) == 3
    assert vi.search(q[0], top_k=3) == []         # 1-D input fails inside -> [] like the reference
    assert vi.add_vectors(x[0]) is False
    big = vi.search(q[:1], top_k=600)             # k > ntotal: -1 ids dropped
    assert len(big) == 500
    # persistence under the reference's file names
    path = vi.save_index()
    assert path.endswith("index_Flat_96d.faiss") and os.path.exists(path.replace(".faiss", "_metadata.json"))
    vi2 = VectorIndex(96)
    assert vi2.load_index("index_Flat_96d.faiss") is True and vi2.total_vectors == 500
    assert [h["index"] for h in vi2.search(q[:1], top_k=7)] == [h["index"] for h in res[0]]
    assert vi2.load_index("missing.faiss") is False
    st = vi.get_index_stats()
    assert st["ntotal"] == 500 and st["index_type"] == "Flat"
    assert vi.remove_vectors(np.array([0, 1])) is True and vi.index.ntotal == 498
    vi.reset_index()
    assert vi.total_vectors == 0 and vi.is_trained is False
    # IVF bookkeeping: training shrinks nlist for tiny corpora, custom ids are honoured
    cfg.set("vector_store.index_type", "IVFFlat")
    iv = VectorIndex(96)
    assert iv.create_index() and not iv.is_trained
    assert iv.add_vectors(x[:50], np.arange(1000, 1050)) and iv.is_trained and iv.nlist == 25
    assert iv.search(q[:1], top_k=1)[0]["index"] >= 1000
    # L2: any metric other than 'cosine' (vector_index.py:69-74)
    cfg.set("vector_store.similarity_metric", "dot_product")
    l2 = VectorIndex(96)
    l2.create_index("Flat")
    l2.add_vectors(x)
    r = l2.search(q[:1], top_k=2)
    Dl, Il = orc.flat_search(q[:1], x.astype(np.float32), 2, "l2")
    assert [h["index"] for h in r] == Il[0].tolist()
    assert abs(r[0]["similarity"] - 1.0 / (1.0 + r[0]["score"])) < 1e-12
    assert l2.create_index("Bogus") is False
    for v in (vi, vi2, iv, l2):
        v.cleanup()


@pytest.fixture()
def model_dir(tmp_path):
    return oenc.make_synthetic_model(str(tmp_path / "bge-small-synth"), layers=2, hidden=128, heads=4,
                                     intermediate=512, pooling="cls", max_pos=128)


def test_embedding_manager_and_retriever(cfg, model_dir):
    from vector_store import EmbeddingManager, VectorRetriever
    EmbeddingManager._reset_singleton()
    cfg.set("embedding.model_path", model_dir)
    cfg.set("embedding.max_length", 64)
    cfg.set("embedding.batch_size", 8)
    cfg.set("vector_store.similarity_threshold", 0.0001)
    em = EmbeddingManager()
    assert em is EmbeddingManager() and em.embedding_dim == 128 and em.model_name == model_dir
    assert em.encode_texts([]).size == 0 and em.encode_queries([]).size == 0
    words = oenc.synthetic_sentences(model_dir, 40, seed=11, min_words=4, max_words=12)
    e1 = em.encode_texts(words[:5])
    ref = oenc.encode(model_dir, words[:5], normalize=True, max_seq_length=64)
    assert e1.dtype == np.float32 and np.sum(e1 * ref, axis=1).min() > 0.9995
    # 'bge' in the *path* -> queries get the instruction prefix (embedding_manager.py:552-559)
    qe = em.encode_queries(words[:2])
    qref = oenc.encode(model_dir, ["Represent this sentence for searching relevant passages: " + w for w in words[:2]],
                       normalize=True, max_seq_length=64)
    assert np.sum(qe * qref, axis=1).min() > 0.9995
    assert em._preprocess_texts(["  x  ", "", "a" * 1000]) == ["x", "Empty content", "a" * 256]
    notes = [{"note_id": f"n{i}", "title": f"t{i}", "content": w, "entities": ["e1", "e2"]} for i, w in enumerate(words)]
    notes[3]["content"] = ""
    t = em._extract_title_raw_span_text(notes[0], [], {})
    assert t == f"t0 || {words[0]} || ENTITIES: e1, e2"
    sim = em.compute_similarity(e1, e1)
    assert sim.shape == (5, 5) and np.allclose(np.diag(sim), 1.0, atol=1e-4)
    # any number of candidates, all three metrics, against the reference's formulas (embedding_manager.py:602-620)
    rng = np.random.default_rng(0)
    A = rng.standard_normal((7, 128)).astype(np.float32)
    Bm = rng.standard_normal((5003, 128)).astype(np.float32)
    An = A / (np.linalg.norm(A, axis=1, keepdims=True) + 1e-8)
    Bn = Bm / (np.linalg.norm(Bm, axis=1, keepdims=True) + 1e-8)
    assert np.allclose(em.compute_similarity(A, Bm, "cosine"), An @ Bn.T, atol=2e-6)
    assert np.allclose(em.compute_similarity(A, Bm, "dot"), A.astype(np.float64) @ Bm.astype(np.float64).T, rtol=1e-5, atol=1e-4)
    dist = np.sqrt(((A[:, None, :].astype(np.float64) - Bm[None, :, :]) ** 2).sum(-1))
    assert np.allclose(em.compute_similarity(A, Bm, "euclidean"), 1.0 / (1.0 + dist), atol=1e-5)
    top = em.find_most_similar(e1[0], e1, top_k=2)
    assert top[0]["index"] == 0 and set(top[0]) == {"index", "similarity"}
    ok, _ = em.validate_model_consistency()
    assert ok

    vr = VectorRetriever()
    assert vr.search(["x"]) == [[]] and vr.retrieve("x") == []
    assert vr.build_index([]) is False
    assert vr.build_index(notes) is True
    assert vr.note_embeddings.shape == (40, 128) and vr.note_id_to_index["n7"] == 7
    assert not hasattr(vr, "id_to_index")          # reference quirk kept
    for f in ("atomic_notes.json", "note_embeddings.npz", "id_mappings.json", "index_Flat_128d.faiss"):
        assert os.path.exists(os.path.join(vr.data_dir if "faiss" not in f else vr.vector_index.index_dir, f))
    res = vr.search(words[:3], top_k=5)
    assert len(res) == 3 and all(len(r) <= 5 for r in res)
    # expected ids from the oracle pipeline: oracle encoder of the same texts + oracle flat search
    note_texts = [em._preprocess_embedding_text(em._extract_title_raw_span_text(n, [], {}), {}) for n in notes]
    xn = oenc.encode(model_dir, [s if len(s.strip()) >= 3 else "Empty note" for s in note_texts], normalize=True,
                     max_seq_length=64)
    qn = oenc.encode(model_dir, ["Represent this sentence for searching relevant passages: " + w for w in words[:3]],
                     normalize=True, max_seq_length=64)
    s64 = orc.exact_scores(orc.preprocess_vectors(qn), orc.preprocess_vectors(xn), "ip")
    for qi, hits in enumerate(res):
        info = hits[0]["retrieval_info"]
        assert set(info) == {"similarity", "score", "rank", "query", "retrieval_method"}
        assert info["query"] == words[qi] and info["retrieval_method"] == "vector_search"
        got = [vr.note_id_to_index[h["note_id"]] for h in hits]
        # f16 encoder vs f32 oracle embeddings: compare through the oracle scores with a tolerance
        best = np.sort(s64[qi])[::-1][:5]
        assert np.allclose([s64[qi][g] for g in got], best[:len(got)], atol=5e-3)
        assert all(abs(h["retrieval_info"]["similarity"] - s64[qi][g]) < 5e-3 for h, g in zip(hits, got))
    slim = vr.search_single(words[0], top_k=2, include_metadata=False)
    assert set(slim[0]) == {"note_id", "content", "paragraph_idxs", "retrieval_info"}
    boosted = vr.retrieve(words[0], top_k=3, boost_entities=[words[0].split()[0]], must_have_terms=["zzzz"])
    assert boosted and "original_similarity" in boosted[0]["retrieval_info"]
    assert "downweighted_missing_terms" in boosted[0]["retrieval_info"]["adjustments"]
    assert vr.add_notes([{"note_id": "new1", "title": "x", "content": words[0]}]) is True
    assert vr.vector_index.total_vectors == 41 and vr.note_embeddings.shape[0] == 41
    assert vr.get_similar_notes("n0", top_k=3) is not None
    assert vr.remove_notes(["new1"]) is True and len(vr.atomic_notes) == 40
    vr2 = VectorRetriever()
    vr2.data_dir, vr2.vector_index.index_dir = vr.data_dir, vr.vector_index.index_dir
    assert vr2.build_index(notes) is True            # reloads the saved index (same count + first id)
    assert vr2.vector_index.total_vectors == 40
    vr.clear_index()
    assert vr.search(["x"]) == [[]]
    vr.cleanup()
    EmbeddingManager._reset_singleton()


def test_device_resident_encode_to_index_handoff(cfg, model_dir):
    """encoder -> index without the host round trip (anr_encoder_forward_dev -> anr_index_add_dev /
    anr_index_search_devq): same embeddings and the same hits as the host-array path, rows in input order although
    the encoder batches by length, tokenisation overlapped with the forward of the previous batch"""
    from vector_store import EmbeddingManager, VectorRetriever
    EmbeddingManager._reset_singleton()
    cfg.set("embedding.model_path", model_dir)
    cfg.set("embedding.max_length", 64)
    cfg.set("embedding.batch_size", 8)           # 61 notes -> 8 length-sorted batches
    cfg.set("vector_store.similarity_threshold", 0.0001)
    em = EmbeddingManager()
    words = oenc.synthetic_sentences(model_dir, 61, seed=21, min_words=2, max_words=30)
    notes = [{"note_id": f"n{i}", "title": f"t{i}", "content": w} for i, w in enumerate(words)]
    host = em.encode_atomic_notes(notes)
    dev = em.encode_texts_device(em._assemble_note_texts(notes))
    assert dev.nq == 61 and dev.n == em.embedding_dim
    assert np.array_equal(dev.numpy(), host)      # same kernels, same rows, input order
    dev.free()
    vr = VectorRetriever()
    assert vr._build_device_resident(notes) is True
    assert np.array_equal(vr.note_embeddings, host) and vr.vector_index.total_vectors == 61
    vr.atomic_notes = notes
    vr._build_id_mappings()
    q = words[:5]
    a = vr._search_device_resident(q, 7)
    b = vr.vector_index.search(em.encode_queries(q), top_k=7)
    assert a is not None and a == b
    assert vr.search(q, top_k=7) and len(vr.search(q, top_k=7)[0]) == 7
    # a sharded (multi-handle) index cannot take device rows: the build falls back to the host path, same result
    cfg.set("anorag_hip.devices", [0, 0])
    vs = VectorRetriever()
    assert vs.build_index(notes, force_rebuild=True, save_index=False) is True
    assert type(vs.vector_index.index).__name__ == "ShardedFlatIndex"
    assert np.array_equal(vs.note_embeddings, host)
    assert [[h["note_id"] for h in r] for r in vs.search(q, top_k=7)] == [[h["note_id"] for h in r] for r in vr.search(q, top_k=7)]
    vr.cleanup()
    vs.cleanup()
    EmbeddingManager._reset_singleton()


def em_row(vr, note):
    return vr.embedding_manager.encode_atomic_notes([note], include_metadata=True)[0]


def test_update_note_optimize_params_and_tfidf_namespace_fallback(cfg, model_dir, monkeypatch):
    """rows c6 / b9 / c10 of SURVEY.md §8a: update_note (re-encode + rebuild), optimize_search_params (the nprobe sweep
    of an exact scan: recall 1.0 at the first value), the TF-IDF "BM25 fallback" and search_with_namespace_fallback
    (reference retriever.py:924-1062; the reference's utils.dataset_guard helper is stood in for by a stub with its
    filtering rule: a note belongs when its dataset / qid fields match)"""
    import sys
    import types
    from vector_store import EmbeddingManager, VectorIndex, VectorRetriever
    EmbeddingManager._reset_singleton()
    cfg.set("embedding.model_path", model_dir)
    cfg.set("embedding.max_length", 64)
    cfg.set("embedding.batch_size", 16)
    cfg.set("vector_store.similarity_threshold", 0.0001)
    words = oenc.synthetic_sentences(model_dir, 30, seed=31, min_words=4, max_words=10)
    notes = [{"note_id": f"n{i}", "title": f"t{i}", "content": w, "dataset": "ds", "qid": "q1" if i < 15 else "q2"}
             for i, w in enumerate(words)]
    vr = VectorRetriever()
    assert vr.build_index(notes, save_index=False)
    # update_note: the row's embedding changes, the index is rebuilt and finds the new text
    before = vr.note_embeddings[4].copy()
    new_note = dict(notes[4], content=words[20] + " " + words[21])
    assert vr.update_note("n4", new_note) is True
    assert not np.allclose(vr.note_embeddings[4], before) and vr.get_note_by_id("n4")["content"] == new_note["content"]
    assert vr.update_note("missing", new_note) is False
    assert np.array_equal(vr.note_embeddings[4], em_row(vr, new_note))          # the re-encoded row, and the rebuilt index
    assert vr.vector_index.search(vr.note_embeddings[4:5], top_k=1)[0]["index"] == 4    # holds it at the same position
    # TF-IDF fallback ("BM25" in the reference): built with the index, cosine over TF-IDF rows
    vr.bm25_enabled = True
    vr._build_bm25_index(vr.atomic_notes)
    tf = vr._bm25_search(words[7], top_k=3)
    assert tf and tf[0]["note_id"] == "n7" and tf[0]["retrieval_info"]["retrieval_method"] == "bm25_fallback"
    assert [h["retrieval_info"]["rank"] for h in tf] == list(range(1, len(tf) + 1))
    # namespace fallback: dense hits outside the namespace are dropped; an empty result falls back to TF-IDF
    guard = types.ModuleType("utils.dataset_guard")
    guard.filter_notes_by_namespace = lambda hits, dataset, qid: [h for h in hits if h.get("dataset") == dataset and h.get("qid") == qid]
    monkeypatch.setitem(sys.modules, "utils", types.ModuleType("utils"))
    monkeypatch.setitem(sys.modules, "utils.dataset_guard", guard)
    res = vr.search_with_namespace_fallback([words[2], words[25]], "ds", "q2", top_k=30)
    assert all(h["qid"] == "q2" for r in res for h in r) and res[1] and res[1][0]["note_id"] == "n25"
    orig_search = vr.search
    vr.search = lambda q, *a, **k: [[] for _ in q]        # dense search finds nothing in the namespace
    res = vr.search_with_namespace_fallback([words[25]], "ds", "q2", top_k=5)
    vr.search = orig_search
    assert res[0] and res[0][0]["retrieval_info"]["retrieval_method"] == "bm25_fallback" and res[0][0]["note_id"] == "n25"
    vr.cleanup()
    EmbeddingManager._reset_singleton()
    # optimize_search_params: only for the IVF types; the exact scan reaches recall 1.0 at the first nprobe tried
    x = np.random.default_rng(0).standard_normal((3000, 64)).astype(np.float32)
    q = np.random.default_rng(1).standard_normal((6, 64)).astype(np.float32)
    vi = VectorIndex(64)
    assert vi.create_index("Flat") and vi.add_vectors(x)
    assert vi.optimize_search_params(q, np.zeros((6, 5), dtype=np.int64)) == {}
    vj = VectorIndex(64)
    assert vj.create_index("IVFFlat") and vj.add_vectors(x)
    _, truth = orc.flat_search(orc.preprocess_vectors(q), orc.preprocess_vectors(x), 10, "ip")
    best = vj.optimize_search_params(q, truth, target_recall=0.9)
    assert best == {"nprobe": 1, "recall": 1.0} and vj.nprobe == 1
    vi.cleanup()
    vj.cleanup()


def _same_hits(D1, I1, D2, I2, tol=5e-4):
    """two top-k results over embeddings that agree to rounding: scores within tol; ids equal except where the scores of
    the swapped entries are within tol of each other (a near tie may flip)"""
    assert D1.shape == D2.shape and np.allclose(D1, D2, atol=tol)
    for r in range(I1.shape[0]):
        for p in np.nonzero(I1[r] != I2[r])[0]:
            where = np.nonzero(I1[r] == I2[r, p])[0]
            other = D1[r, where[0]] if len(where) else D1[r, -1]   # (pushed out of the list: compare with its end)
            assert abs(float(other) - float(D1[r, p])) <= 2 * tol


def test_streamed_and_sharded_builds_equal_the_one_shot_build(cfg, model_dir, tmp_path):
    """anorag_hip.offline_build: chunked device-resident encode -> add (with the embeddings.npy side file), and the
    per-rank shard builds (two ranks played one after the other on this device) whose merged search equals the single
    index.  The chunked builds run forwards of other sizes than the one-shot encode (other GEMM kernel variants,
    other summation orders whose f32 results round to different f16 activations), so the embeddings agree to the encoder's
    tolerance (cosine >= 0.99999 here), not bit for bit."""
    from anorag_hip import FlatIndex, METRIC_IP
    from anorag_hip.offline_build import sharded_build, stream_build
    from anorag_hip.sharded import merge_topk_host_c
    from vector_store import EmbeddingManager
    EmbeddingManager._reset_singleton()
    cfg.set("embedding.model_path", model_dir)
    cfg.set("embedding.max_length", 64)
    cfg.set("embedding.batch_size", 16)
    em = EmbeddingManager()
    words = oenc.synthetic_sentences(model_dir, 150, seed=41, min_words=3, max_words=20)
    notes = [{"note_id": f"n{i}", "title": f"t{i}", "content": w} for i, w in enumerate(words)]
    ref = em.encode_atomic_notes(notes)
    one = FlatIndex(em.embedding_dim, METRIC_IP, normalize=True)
    one.add(ref)
    idx = FlatIndex(em.embedding_dim, METRIC_IP, normalize=True)
    npy = str(tmp_path / "embeddings.npy")
    assert stream_build(idx, em, notes, chunk_notes=64, embeddings_npy=npy) == 150 and idx.ntotal == 150
    got = np.load(npy)
    assert got.shape == ref.shape and np.max(np.abs(got - ref)) <= 1e-3 and np.sum(got * ref, axis=1).min() >= 0.99999
    q = em.encode_queries(words[:9])
    D1, I1 = one.search(q, 12)
    D2, I2 = idx.search(q, 12)
    _same_hits(D1, I1, D2, I2)
    parts = []
    for rank in range(2):
        searcher, shard, (lo, hi) = sharded_build(em, notes, world=2, rank=rank, chunk_notes=50)
        assert shard.ntotal == hi - lo and searcher.row_offset == lo
        parts.append(shard.search(q, 12))          # global ids (ANR_OPT_ID_OFFSET set by the searcher)
        shard.close()
    Dm, Im = merge_topk_host_c(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), True)
    _same_hits(D1, I1, Dm, Im)
    one.close()
    idx.close()
    EmbeddingManager._reset_singleton()


def test_worker_threads_share_the_encoder_the_index_and_the_fusion(cfg, model_dir):
    """The reference answers questions from ThreadPoolExecutor workers that share the EmbeddingManager singleton
    (main_musique.py:487-494, embedding_manager.py:64-70): encode -> search -> BM25 -> fuse from eight threads at once
    must give every thread the single-threaded answer."""
    from concurrent.futures import ThreadPoolExecutor
    from anorag_hip import bm25_search as dbm
    from retrieval.hybrid_search import HybridSearcher
    from vector_store import EmbeddingManager, VectorRetriever
    EmbeddingManager._reset_singleton()
    cfg.set("embedding.model_path", model_dir)
    cfg.set("embedding.max_length", 64)
    cfg.set("embedding.batch_size", 16)
    cfg.set("vector_store.similarity_threshold", 0.0001)
    words = oenc.synthetic_sentences(model_dir, 300, seed=21, min_words=4, max_words=14)
    notes = [{"note_id": f"n{i}", "title": f"t{i}", "content": w, "entities": []} for i, w in enumerate(words)]
    vr = VectorRetriever()
    assert vr.build_index(notes) is True
    corpus = dbm.build_bm25_corpus(notes, lambda n: n["content"])
    hs = HybridSearcher({"retrieval": {"candidate_pool": 20, "hybrid": {
        "enabled": True, "fusion_method": "linear", "weights": {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}}}})
    queries = [words[i] for i in range(0, 240, 5)]

    def answer(q):
        hits = vr.search([q], top_k=10)[0]
        dense = [(h["note_id"], h["retrieval_info"]["similarity"]) for h in hits]
        sc = dbm.bm25_scores(corpus, notes, q)
        bm = [(notes[i]["note_id"], s) for i, s in enumerate(sc) if s > 0][:200]
        return [(r["note_id"], r["final_similarity"]) for r in hs.fuse(dense=dense, bm25=bm)]

    expected = [answer(q) for q in queries]
    assert all(len(e) > 0 for e in expected)
    for _ in range(3):
        with ThreadPoolExecutor(max_workers=8) as ex:
            got = list(ex.map(answer, queries))
        assert got == expected
    corpus.close()
