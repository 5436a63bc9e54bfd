"""CPU tests (no GPU needed): the C-ABI shared library loads, exports every symbol include/anorag.h declares,
fails loudly (no silent fallback) when there is no device; host-side logic of the drop-in classes."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "anorag.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(anr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from anorag_hip import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/anorag.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.anr_version().startswith(b"anorag-hip")


def test_no_device_fails_loudly():
    from anorag_hip import AnoragError, FlatIndex, _lib
    if _lib.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(AnoragError) as e:
        FlatIndex(64)
    assert "no HIP device" in str(e.value)
    # argument validation happens before any device work
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.anr_index_create(0, 0, 1, 0, C.byref(h)) == -1 and b"dim" in lib.anr_last_error()
    assert lib.anr_index_create(64, 7, 1, 0, C.byref(h)) == -1
    assert lib.anr_index_search(None, None, 1, 1, None, None) == -1
    from retrieval.hybrid_search import HybridSearcher
    hs = HybridSearcher({"retrieval": {"hybrid": {"weights": {"dense": 1.0}}}})
    with pytest.raises(AnoragError):
        hs.fuse(dense=[("a", 1.0)])          # the product path never falls back to Python arithmetic
    from vector_store.vector_index import VectorIndex
    vi = VectorIndex(32)
    assert vi.use_gpu is False and vi.create_index("Flat") is False and vi.search(np.zeros((1, 32))) == []


def test_hybrid_searcher_config_surface():
    from retrieval.hybrid_search import HybridSearcher, create_hybrid_searcher
    hs = create_hybrid_searcher({})
    assert (hs.candidate_pool, hs.enabled, hs.fusion_method, hs.weights, hs.rrf_k) == (50, True, "linear", {}, 60)

    class Obj:
        def load_config(self):
            return {"retrieval": {"candidate_pool": 80, "hybrid": {"enabled": False, "fusion_method": "rrf",
                                                                    "weights": {"dense": 1.0}, "rrf_k": 10}}}
    hs = HybridSearcher(Obj())
    assert hs.candidate_pool == 80 and hs.fusion_method == "rrf" and hs.rrf_k == 10
    assert hs.fuse(dense=[("a", 1.0)]) == []       # disabled -> [] without touching the device
    assert hs._normalize({}) == {} and hs._normalize({"a": 0.0}) == {"a": 0.0}
    assert hs._normalize({"a": 2.0, "b": 1.0}) == {"a": 1.0, "b": 0.5}


def test_vector_index_host_view_matches_oracle_preprocess():
    from oracle import flat_index as orc
    from vector_store.vector_index import VectorIndex
    vi = VectorIndex(16)
    x = np.random.default_rng(0).standard_normal((20, 16))
    x[3] = 0
    a = vi._preprocess_vectors(np.asfortranarray(x))
    b = orc.preprocess_vectors(x)
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and np.array_equal(a, b)
    assert vi.index_type == "IVFFlat" and vi.similarity_metric == "cosine" and vi.nlist == 100 and vi.nprobe == 10
    assert vi.add_vectors(x) is False and vi.save_index() == "" and vi.get_index_stats() == {}
    assert vi.train_index(x) is False and vi.remove_vectors(np.array([1])) is False
    assert vi._calculate_recall([[{"index": 1}, {"index": 2}]], np.array([[1, 3]])) == 0.5


def test_embedding_manager_text_rules_without_model():
    from vector_store.embedding_manager import EmbeddingManager
    em = object.__new__(EmbeddingManager)   # text helpers only; the constructor needs a model + device
    em.max_length = 8
    assert em._preprocess_texts([" ab ", "", "x" * 40]) == ["ab", "Empty content", "x" * 32]
    note = {"title": " T ", "content": "", "raw_span": " body ", "entities": ["a", "", "b"]}
    assert em._extract_title_raw_span_text(note, [], {}) == "T || body || ENTITIES: a, b"
    long = {"title": "t", "content": "c" * 600}
    assert len(em._extract_title_raw_span_text(long, [], {})) == 512
    assert em._extract_title_raw_span_text(long, [], {})[-14:] == " || ENTITIES: "
    assert em._preprocess_embedding_text("a \n  b\x07 Ａ", {}) == "a b A"
    assert em._should_skip_note("ab", {}) and not em._should_skip_note("abc", {})
    assert em._extract_title_content_text({"title": "a", "content": "b"}, "content", {}) == "a b"


def test_model_dir_reader_and_name_map(tmp_path):
    from anorag_hip.encoder import map_hf_name, read_model_dir
    from oracle import encoder as oenc
    d = oenc.make_synthetic_model(str(tmp_path / "m"), layers=1, hidden=64, heads=2, intermediate=128, vocab=300,
                                  max_pos=64, pooling="cls")
    info = read_model_dir(d)
    assert info["pooling"] == "cls" and info["normalize_module"] and info["pos_offset"] == 0
    assert info["max_seq_length"] == 64
    assert map_hf_name("bert.encoder.layer.3.attention.self.query.weight") == "L3.q.w"
    assert map_hf_name("encoder.layer.0.output.LayerNorm.bias") == "L0.ln2.b"
    assert map_hf_name("roberta.embeddings.word_embeddings.weight") == "emb.word"
    assert map_hf_name("pooler.dense.weight") is None


def test_mpnet_relative_bias_table_matches_transformers():
    """host logic: the per-offset bias table handed to the encoder (rel.bias) equals what transformers' MPNetEncoder
    computes with its bucket function, for every (query, key) pair of a 512-position model"""
    import numpy as np
    import torch
    from transformers.models.mpnet.modeling_mpnet import MPNetEncoder
    from anorag_hip.encoder import relative_bias_table
    w = np.random.default_rng(0).standard_normal((32, 12)).astype(np.float32)
    span = 512
    tab = relative_bias_table(w, span)
    assert tab.shape == (12, 2 * span - 1) and tab.dtype == np.float32
    q = torch.arange(span)[:, None]
    k = torch.arange(span)[None, :]
    ref = w[MPNetEncoder.relative_position_bucket(k - q, num_buckets=32).numpy()]      # [q, k, heads]
    qi, ki = np.meshgrid(np.arange(span), np.arange(span), indexing="ij")
    assert np.array_equal(tab[:, (ki - qi) + span - 1].transpose(1, 2, 0), ref)


def test_mpnet_parameter_names_map_to_encoder_tensors():
    from anorag_hip.encoder import map_hf_name
    assert map_hf_name("mpnet.encoder.layer.3.attention.attn.q.weight") == "L3.q.w"
    assert map_hf_name("encoder.layer.0.attention.attn.o.bias") == "L0.o.b"
    assert map_hf_name("encoder.layer.11.attention.LayerNorm.weight") == "L11.ln1.g"
    assert map_hf_name("encoder.layer.11.output.LayerNorm.bias") == "L11.ln2.b"
    assert map_hf_name("encoder.relative_attention_bias.weight") == "rel.weight"
    assert map_hf_name("embeddings.word_embeddings.weight") == "emb.word"
    assert map_hf_name("pooler.dense.weight") is None


def test_bfloat16_and_float16_checkpoints_load_as_float32(tmp_path):
    """host logic: safetensors checkpoints stored in f16 or bf16 (no numpy dtype for the latter) arrive as float32"""
    import os
    import torch
    from safetensors.torch import load_file, save_file
    from oracle import encoder as oenc
    from anorag_hip import encoder as penc
    d = oenc.make_synthetic_model(str(tmp_path / "m"), layers=1, hidden=64, heads=2, intermediate=128, vocab=300, max_pos=64)
    ref = penc.load_weights(d)
    st = os.path.join(d, "model.safetensors")
    t = load_file(st)
    for dt, tol in ((torch.float16, 2e-3), (torch.bfloat16, 2e-2)):
        save_file({k: v.to(dt) for k, v in t.items()}, st)
        w = penc.load_weights(d)
        assert set(w) == set(ref)
        for k in ref:
            assert w[k].dtype == np.float32 and w[k].shape == ref[k].shape
            assert np.max(np.abs(w[k] - ref[k])) <= tol * max(1.0, float(np.max(np.abs(ref[k]))))


def test_c_result_shaping_builds_the_same_objects_as_the_python_loops():
    """csrc/pyshape.c (host-side formatting, no device work): shape_hits == VectorIndex's loop, shape_fused ==
    HybridSearcher's loop — same keys in the same order, same Python types, same values"""
    from anorag_hip import _pyshape
    from retrieval import hybrid_search as hsm
    from vector_store import vector_index as vim
    rng = np.random.default_rng(3)
    nq, pool = 7, 12
    ids = rng.integers(0, 50, size=(nq, pool)).astype(np.int64)
    fin = rng.standard_normal((nq, pool))
    src = rng.standard_normal((nq, pool, 4))
    src[rng.random((nq, pool, 4)) < 0.4] = np.nan
    cnt = rng.integers(0, pool + 1, size=nq).astype(np.int32)
    cnt[0], cnt[1] = 0, pool
    names = [f"note_{i}" for i in range(50)]
    for note_ids in (None, names):
        got = _pyshape.shape_fused(ids, fin, src, cnt, nq, pool, note_ids)
        exp = hsm._shape_fused_py(ids, fin, src, cnt, note_ids)
        assert got == exp
        for g, e in zip(got, exp):
            for a, b in zip(g, e):
                assert list(a) == list(b) and list(a["scores"]) == list(b["scores"]) and list(a["tags"]) == list(b["tags"])
                assert type(a["note_id"]) is type(b["note_id"]) and type(a["tags"]["is_bridge"]) is bool
    assert hsm._shape_fused(ids, fin, src, cnt, names) == hsm._shape_fused_py(ids, fin, src, cnt, names)
    with pytest.raises(ValueError):
        _pyshape.shape_fused(ids, fin, src, cnt, nq + 1, pool, None)
    with pytest.raises(IndexError):
        _pyshape.shape_fused(ids, fin, src, cnt, nq, pool, names[:3])
    # VectorIndex.search's shaping, cosine and L2
    I = rng.integers(-1, 30, size=(4, 9)).astype(np.int64)
    S = rng.random((4, 9)).astype(np.float32)
    for cosine in (True, False):
        sim = S.tolist() if cosine else (1.0 / (1.0 + S.astype(np.float64))).tolist()
        exp = [[{"index": i, "score": s, "rank": r, "similarity": m} for r, (i, s, m) in enumerate(zip(a, b, c)) if i != -1]
               for a, b, c in zip(I.tolist(), S.tolist(), sim)]
        assert vim._shape_hits(I, S, cosine) == exp


def test_sentence_encoder_token_slices_bound_a_forward():
    """host logic: a tokenised batch is cut into forwards of at most max_forward_tokens padded tokens (255 long
    notes under a long-context model must not become one 2 M-token forward)"""
    from anorag_hip.encoder import SentenceEncoder
    enc = object.__new__(SentenceEncoder)
    enc.max_forward_tokens = 1000
    lens = np.array([500, 400, 100, 90, 33, 10, 5])
    cuts = list(enc._token_slices(lens))
    assert cuts == [(0, 1, 500), (1, 3, 400), (3, 7, 90)]
    for a, b, L in cuts:
        assert L == lens[a:b].max() and ((b - a) * ((L + 31) // 32 * 32) <= 1000 or b - a == 1)
    enc.max_forward_tokens = 1 << 18
    assert list(enc._token_slices(lens)) == [(0, 7, 500)]
    assert list(enc._token_slices(np.array([], dtype=np.int64))) == []
    enc.max_forward_tokens = 1          # a single sentence always goes through
    assert list(enc._token_slices(np.array([40, 30]))) == [(0, 1, 40), (1, 2, 30)]


@pytest.mark.parametrize("sizes,pool,method", [((5000, 5000, 900, 0), 50, "linear"), ((5000, 5000, 900, 300), 50, "linear"),
                                               ((4100, 700, 600, 500), 80, "linear"), ((6000, 10, 5, 3), 1500, "linear"),
                                               ((6000, 10, 5, 3), 1500, "rrf"), ((5000, 900, 0, 0), 50, "rrf"),
                                               ((5000, 1200, 0, 0), 50, "rrf"), ((5000, 10, 0, 5), 50, "rrf")])
def test_fuse_long_routing_respects_the_kernel_limits(monkeypatch, sizes, pool, method):
    """host logic of HybridSearcher._fuse_long (no device): which sources go to anr_fuse_dense as arrays — the call must
    satisfy the limits the C side checks (short-list entries <= 1024, n_arr * (pool + 2 m) + m <= 4096, pool <= 1024, rrf:
    one array with a non-negative weight) or go to the sorting path / the rounds"""
    from retrieval import hybrid_search as hsm
    calls = []

    class FakeArray:
        def __init__(self, a):
            self.a, self.row_max, self.nq, self.n, self.dtype = a, None, a.shape[0], a.shape[1], a.dtype

        @classmethod
        def from_numpy(cls, a, device=0, with_max=False):
            return cls(np.asarray(a))

        def free(self):
            pass

    def fake_fuse_dense(m, weights, rrf_k, pl, nq, sources, device=0, want_stats=False):
        n_arr = sum(isinstance(v, FakeArray) for v in sources.values())
        short = sum(len(v[0][0]) for v in sources.values() if not isinstance(v, FakeArray))
        calls.append((m, pl, n_arr, short))
        assert pl <= 1024 and short <= 1024 and n_arr * (pl + 2 * short) + short <= 4096 and n_arr >= 1
        if m == "rrf":
            assert n_arr == 1
        return (np.full((1, pl), -1, np.int64), np.zeros((1, pl)), np.zeros((1, pl, 4)), np.zeros(1, np.int32))

    monkeypatch.setattr(hsm, "DeviceArray", FakeArray)
    monkeypatch.setattr(hsm, "fuse_dense", fake_fuse_dense)
    sorted_path = []
    monkeypatch.setattr(hsm.HybridSearcher, "_fuse_rrf_long", lambda self, dicts, pl: sorted_path.append(pl) or [])
    rng = np.random.default_rng(1)
    lists = []
    for m in sizes:
        ids = rng.choice(9000, m, replace=False)
        lists.append([(f"n{int(i)}", float(s)) for i, s in zip(ids, rng.random(m))])
    hs = hsm.HybridSearcher({"retrieval": {"candidate_pool": pool, "hybrid": {"fusion_method": method, "rrf_k": 60,
                                                                               "weights": {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}}}})
    assert hs._fuse_long(tuple(lists), pool) == []
    if method == "rrf" and (pool > 1024 or sizes[1] > 1024):
        assert sorted_path and not calls          # several long lists / a big pool: ranked by device sorts
    else:
        assert calls and not sorted_path
