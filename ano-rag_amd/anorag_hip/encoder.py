"""Host side of the device sentence encoder: model-directory reader, tokeniser front end and the
``SentenceEncoder`` object the drop-in ``EmbeddingManager`` holds in ``self.model`` (where the reference holds
a ``sentence_transformers.SentenceTransformer``, embedding_manager.py:350).

What runs where: tokenisation (``tokenizers``) and batching on the host; embedding gather, transformer
blocks, pooling and L2 normalisation on the device through ``anr_encoder_*`` (libanorag_hip.so).  There is no
CPU forward: without the HIP library or a device, construction raises.

The pipeline restates ``SentenceTransformer.encode`` (sentence-transformers >= 2.2, not vendored in the
reference): sort by text length (longest first), batches of ``batch_size``, tokenise with truncation to
``max_seq_length`` and padding to the longest of the batch, Transformer -> Pooling (mode from the model's
``1_Pooling/config.json``) -> optional Normalize module, then ``normalize_embeddings``, results restored to
the input order as float32.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import threading
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import EncoderConfig

_PREFIXES = ("bert.", "roberta.", "xlm_roberta.", "mpnet.", "model.", "0.auto_model.", "auto_model.")


def _strip(name: str) -> str:
    changed = True
    while changed:
        changed = False
        for p in _PREFIXES:
            if name.startswith(p):
                name = name[len(p):]
                changed = True
    return name


def map_hf_name(name: str) -> Optional[str]:
    """HF BERT/RoBERTa parameter name -> tensor name of anr_encoder_set_tensor (None = not used)."""
    n = _strip(name)
    table = {
        "embeddings.word_embeddings.weight": "emb.word",
        "embeddings.position_embeddings.weight": "emb.pos",
        "embeddings.token_type_embeddings.weight": "emb.type",
        "embeddings.LayerNorm.weight": "emb.ln.g",
        "embeddings.LayerNorm.bias": "emb.ln.b",
    }
    if n in table:
        return table[n]
    if n == "encoder.relative_attention_bias.weight":
        return "rel.weight"  # MPNet: [buckets, heads]; load_weights turns it into the per-offset table "rel.bias"
    if not n.startswith("encoder.layer."):
        return None
    rest = n[len("encoder.layer."):]
    li, _, tail = rest.partition(".")
    sub = {
        "attention.self.query": "q", "attention.self.key": "k", "attention.self.value": "v",
        "attention.output.dense": "o", "intermediate.dense": "ffn1", "output.dense": "ffn2",
        # MPNet names
        "attention.attn.q": "q", "attention.attn.k": "k", "attention.attn.v": "v", "attention.attn.o": "o",
    }
    for k, v in sub.items():
        if tail == k + ".weight":
            return f"L{li}.{v}.w"
        if tail == k + ".bias":
            return f"L{li}.{v}.b"
    ln = {"attention.output.LayerNorm": "ln1", "output.LayerNorm": "ln2", "attention.LayerNorm": "ln1"}
    for k, v in ln.items():
        if tail == k + ".weight":
            return f"L{li}.{v}.g"
        if tail == k + ".bias":
            return f"L{li}.{v}.b"
    return None


def read_model_dir(path: str) -> Dict:
    """config + pooling + tokenizer location of a sentence-transformers / HF model directory."""
    with open(os.path.join(path, "config.json")) as f:
        hf = json.load(f)
    mtype = hf.get("model_type", "bert")
    if mtype not in ("bert", "roberta", "xlm-roberta", "mpnet"):
        raise ValueError(f"model_type {mtype!r} is not supported by the HIP encoder (bert / roberta / xlm-roberta / mpnet)")
    if hf.get("position_embedding_type", "absolute") != "absolute":
        raise ValueError("only absolute position embeddings are supported")
    act = hf.get("hidden_act", "gelu")
    if act != "gelu":
        raise ValueError(f"hidden_act {act!r} is not supported (gelu)")
    pooling, normalize_module, max_seq = "mean", False, None
    mod_file = os.path.join(path, "modules.json")
    if os.path.exists(mod_file):
        with open(mod_file) as f:
            for m in json.load(f):
                t = m.get("type", "")
                if t.endswith("Pooling"):
                    with open(os.path.join(path, m.get("path", "1_Pooling"), "config.json")) as pf:
                        pc = json.load(pf)
                    if pc.get("pooling_mode_cls_token"):
                        pooling = "cls"
                    elif pc.get("pooling_mode_mean_tokens", True):
                        pooling = "mean"
                    else:
                        raise ValueError("unsupported pooling mode (only mean / cls)")
                elif t.endswith("Normalize"):
                    normalize_module = True
    sb = os.path.join(path, "sentence_bert_config.json")
    if os.path.exists(sb):
        with open(sb) as f:
            max_seq = json.load(f).get("max_seq_length")
    pad_idx = hf.get("pad_token_id", 0 if mtype == "bert" else 1)
    return {
        "hf": hf,
        "model_type": mtype,
        "pooling": pooling,
        "normalize_module": normalize_module,
        "max_seq_length": max_seq,
        "pos_offset": 0 if mtype == "bert" else int(pad_idx) + 1,
        "pad_token_id": int(pad_idx),
    }


def relative_bias_table(weight: np.ndarray, span: int, max_distance: int = 128) -> np.ndarray:
    """MPNet's relative position bias as a per-offset table: out[h][(key - query) + span - 1] (float32
    [heads][2 span - 1]) from relative_attention_bias.weight [buckets][heads], with the bucket function of
    transformers' MPNetEncoder.relative_position_bucket (bidirectional T5 buckets)."""
    import math
    import torch  # same float32 log as transformers uses: bucket edges must not move by an ulp
    buckets, heads = weight.shape
    rel = torch.arange(-(span - 1), span, dtype=torch.long)  # key - query
    n = -rel
    half = buckets // 2
    ret = (n < 0).to(torch.long) * half
    n = torch.abs(n)
    max_exact = half // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (half - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, half - 1))
    ret = ret + torch.where(n < max_exact, n, large)
    return np.ascontiguousarray(weight[ret.numpy()].T, dtype=np.float32)  # [heads][2 span - 1]


def load_weights(path: str) -> Dict[str, np.ndarray]:
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        try:
            from safetensors.numpy import load_file
            raw = load_file(st)
        except (TypeError, ValueError):  # bfloat16 checkpoints have no numpy dtype: go through torch
            from safetensors.torch import load_file as load_torch
            raw = {k: v.float().numpy() for k, v in load_torch(st).items()}
    else:
        import torch
        raw = {k: v.float().numpy() for k, v in torch.load(os.path.join(path, "pytorch_model.bin"),
                                                           map_location="cpu").items()}
    out = {}
    for k, v in raw.items():
        name = map_hf_name(k)
        if name is not None:
            out[name] = np.ascontiguousarray(v, dtype=np.float32)
    if "rel.weight" in out:  # MPNet
        with open(os.path.join(path, "config.json")) as f:
            hf = json.load(f)
        span = int(hf["max_position_embeddings"]) - (int(hf.get("pad_token_id", 1)) + 1)
        out["rel.bias"] = relative_bias_table(out.pop("rel.weight"), span)
        if "emb.type" not in out:  # no token-type embeddings in MPNet: a zero row keeps the embedding kernel uniform
            out["emb.type"] = np.zeros((1, out["emb.word"].shape[1]), dtype=np.float32)
    return out


def load_tokenizer(path: str):
    from tokenizers import Tokenizer
    tj = os.path.join(path, "tokenizer.json")
    if os.path.exists(tj):
        return Tokenizer.from_file(tj)
    vocab = os.path.join(path, "vocab.txt")
    if os.path.exists(vocab):
        from tokenizers import BertWordPieceTokenizer
        lower = True
        tc = os.path.join(path, "tokenizer_config.json")
        if os.path.exists(tc):
            with open(tc) as f:
                lower = json.load(f).get("do_lower_case", True)
        return BertWordPieceTokenizer(vocab, lowercase=lower)._tokenizer
    raise FileNotFoundError(f"no tokenizer.json or vocab.txt under {path}")


class DeviceEncoder:
    """anr_encoder handle + weights."""

    def __init__(self, hf_config: Dict, tensors: Dict[str, np.ndarray], pooling: str = "mean", pos_offset: int = 0,
                 device: int = 0):
        self._lib = _lib.load()
        cfg = EncoderConfig(
            n_layers=int(hf_config["num_hidden_layers"]), hidden=int(hf_config["hidden_size"]),
            n_heads=int(hf_config["num_attention_heads"]), intermediate=int(hf_config["intermediate_size"]),
            vocab_size=int(hf_config["vocab_size"]), max_positions=int(hf_config["max_position_embeddings"]),
            type_vocab_size=int(hf_config.get("type_vocab_size", 1)), pos_offset=int(pos_offset),
            pooling=1 if pooling == "cls" else 0, act=0, ln_eps=float(hf_config.get("layer_norm_eps", 1e-12)),
        )
        self.hidden = cfg.hidden
        self.max_positions = cfg.max_positions - cfg.pos_offset
        h = C.c_void_p()
        _lib.check(self._lib.anr_encoder_create(C.byref(cfg), int(device), C.byref(h)), "anr_encoder_create")
        self._h = h
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            _lib.check(self._lib.anr_encoder_set_tensor(self._h, name.encode(), a.ctypes.data, a.size),
                       f"anr_encoder_set_tensor({name})")
        _lib.check(self._lib.anr_encoder_finalize(self._h), "anr_encoder_finalize")

    def forward(self, ids: np.ndarray, lengths: np.ndarray, type_ids: Optional[np.ndarray] = None,
                normalize: bool = False, shared: bool = False) -> np.ndarray:
        """``shared``: through ``anr_encoder_forward_shared`` — calls from other threads that arrive while a forward runs
        are merged into the next one (small requests only; bit-identical results)"""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        B, L = ids.shape
        out = np.empty((B, self.hidden), dtype=np.float32)
        tp = None
        if type_ids is not None:
            type_ids = np.ascontiguousarray(type_ids, dtype=np.int32)
            tp = type_ids.ctypes.data
        fn = self._lib.anr_encoder_forward_shared if shared else self._lib.anr_encoder_forward
        _lib.check(fn(self._h, ids.ctypes.data, lengths.ctypes.data, tp, B, L,
                      int(bool(normalize)), out.ctypes.data), "anr_encoder_forward")
        return out

    def shared_stats(self):
        """(forwards run, requests served) by the combining queue of ``forward(shared=True)``"""
        f, r = C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib.anr_encoder_shared_stats(self._h, C.byref(f), C.byref(r)), "anr_encoder_shared_stats")
        return int(f.value), int(r.value)

    def forward_device(self, ids: np.ndarray, lengths: np.ndarray, type_ids: Optional[np.ndarray], normalize: bool,
                       out_ptr: int, out_rows: Optional[np.ndarray] = None) -> None:
        """the same forward, sequence b written to row out_rows[b] of the float32 [*, hidden] DEVICE buffer at out_ptr"""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        B, L = ids.shape
        tp = rows = None
        if type_ids is not None:
            type_ids = np.ascontiguousarray(type_ids, dtype=np.int32)
            tp = type_ids.ctypes.data
        if out_rows is not None:
            out_rows = np.ascontiguousarray(out_rows, dtype=np.int32)
            rows = out_rows.ctypes.data
        _lib.check(self._lib.anr_encoder_forward_dev(self._h, ids.ctypes.data,
                                                     lengths.ctypes.data, tp, B, L, int(bool(normalize)),
                                                     C.c_void_p(out_ptr), rows), "anr_encoder_forward_dev")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.anr_encoder_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class SentenceEncoder:
    """The subset of the SentenceTransformer surface the reference uses: ``encode``,
    ``get_sentence_embedding_dimension``, ``max_seq_length`` (embedding_manager.py:353-362, :392-399)."""

    def __init__(self, model_path: str, device: int = 0, trust_remote_code: bool = True, **_):
        info = read_model_dir(model_path)
        self.model_path = model_path
        self._info = info
        self.tokenizer = load_tokenizer(model_path)
        self._enc = DeviceEncoder(info["hf"], load_weights(model_path), pooling=info["pooling"],
                                  pos_offset=info["pos_offset"], device=device)
        self.max_seq_length = int(info["max_seq_length"] or min(512, self._enc.max_positions))
        self._pad = info["pad_token_id"]
        self.device = device
        self._pool = None  # tokeniser worker (created on first multi-batch encode)
        self._tok_lock = threading.Lock()
        self._tok_max_len = None
        # padded tokens (sequences x padded length) one forward may hold: the workspace is 16 H + 2 I bytes per token
        # (6.4 GB at this budget for XLM-R large) — the role SentenceTransformer.encode's batch_size plays as a memory knob
        self.max_forward_tokens = 1 << 18
        # encodes of a few sentences (a query at a time from worker threads) go through the library's combining queue
        # (anr_encoder_forward_shared, csrc/combine.hpp)
        self.combine_max_sentences = 8

    def get_sentence_embedding_dimension(self) -> int:
        return self._enc.hidden

    def tokenize(self, texts: Sequence[str]):
        tok = self.tokenizer
        max_len = int(min(self.max_seq_length, self._enc.max_positions))
        # One tokenisation at a time, and the truncation / padding settings touched only when max_seq_length changed: the
        # reference encodes from ThreadPoolExecutor workers that share the model (main_musique.py:487-494), and a
        # `tokenizers` object that is reconfigured while another thread encodes raises "Already borrowed"
        # (encode_batch is parallel inside, so nothing is lost by taking turns).
        with self._tok_lock:
            if self._tok_max_len != max_len:
                tok.enable_truncation(max_length=max_len)
                tok.no_padding()
                self._tok_max_len = max_len
            encs = tok.encode_batch([str(t) for t in texts])
        lists = [e.ids for e in encs]  # ONE conversion per sentence (each `.ids` / `.type_ids` access builds a new list)
        lens = np.fromiter((len(x) for x in lists), dtype=np.int32, count=len(lists))
        L = int(lens.max())
        ids = np.full((len(lists), L), self._pad, dtype=np.int32)
        for i, x in enumerate(lists):
            ids[i, :len(x)] = x
        # single sentences: every token is of segment 0 (the reference never encodes pairs, embedding_manager.py:392)
        types = np.zeros((len(lists), L), dtype=np.int32)
        return ids, lens, types

    def _batches(self, sentences, batch_size):
        """length-sorted batches (longest first, as SentenceTransformer.encode), tokenised one batch AHEAD on a worker
        thread: the tokenizer (Rust, releases the GIL) prepares batch i + 1 while the device runs the forward of
        batch i (the ctypes call releases the GIL as well).  The forward size is the device's business
        (``batch_size`` is accepted and ignored; memory is bounded by ``max_forward_tokens``, see ``_token_slices``): fewer than 512 sentences go in ONE forward (100
        sentences as four forwards of 32 took 5 ms, as one 2.5 ms — small forwards leave most CUs idle; 256 queries as two
        forwards of 128 with the second tokenised beside the first: 6.9-7.2 ms, as one: 6.0-6.7 ms, round 3), 512 and more
        in forwards of 256 (the next is tokenised while one runs); a forward also ends where the sorted texts get shorter than 60 % of its
        first; an embedding does not depend on which sentences share its forward (padding is masked)."""
        n = len(sentences)
        if n == 0:
            return
        order = np.argsort([-len(s) for s in sentences], kind="stable")
        fb = n if n < 512 else 256
        # a forward is padded to its longest sentence: cut where the texts (longest first) fall below 60 % of the
        # forward's first one, but never before 16 sentences — mixed-length notes do not pay for the longest of all
        sels, i = [], 0
        while i < n:
            first = max(len(sentences[order[i]]), 1)
            j = i + 1
            # (short texts are not cut: a forward is padded to a multiple of 32 tokens anyway, and below ~100 tokens a
            # second, smaller forward costs more than the padding it saves — 256 queries of 17-33 words: 8.2 ms cut in
            # two, 6.0-6.7 ms as one)
            while j < n and j - i < fb and (j - i < 16 or first < 512 or len(sentences[order[j]]) * 10 >= first * 6):
                j += 1
            sels.append(order[i:j])
            i = j
        if len(sels) >= 2 and len(sels[-1]) < 16:  # a short tail rides with the forward before it
            sels[-2:] = [np.concatenate(sels[-2:])]
        if len(sels) == 1:
            yield sels[0], self.tokenize([sentences[i] for i in sels[0]])
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="anr-tok")
        nxt = self._pool.submit(self.tokenize, [sentences[i] for i in sels[0]])
        for bi, sel in enumerate(sels):
            cur = nxt.result()
            if bi + 1 < len(sels):
                nxt = self._pool.submit(self.tokenize, [sentences[i] for i in sels[bi + 1]])
            yield sel, cur

    def _token_slices(self, lens):
        """row ranges [a, b) of one tokenised batch (longest first) and the padded width each needs, so that no forward
        holds more than ``max_forward_tokens`` padded tokens: 255 long notes under a long-context model (bge-m3:
        max_seq_length 8192) are 2 M tokens, tens of GB of workspace as ONE forward"""
        n, a = len(lens), 0
        budget = max(int(self.max_forward_tokens), 32)
        while a < n:
            b, L = a, 0
            while b < n:
                L2 = max(L, int(lens[b]))
                if b > a and (b + 1 - a) * ((L2 + 31) // 32 * 32) > budget:
                    break
                L = L2
                b += 1
            yield a, b, L
            a = b

    def encode(self, sentences, batch_size: int = 32, show_progress_bar: bool = False, convert_to_numpy: bool = True,
               normalize_embeddings: bool = False, device=None, **_):
        single = isinstance(sentences, str)
        if single:
            sentences = [sentences]
        n = len(sentences)
        out = np.zeros((n, self._enc.hidden), dtype=np.float32)
        want_norm = bool(normalize_embeddings) or self._info["normalize_module"]
        use_types = self._info["hf"].get("type_vocab_size", 1) > 1
        if 0 < n <= self.combine_max_sentences:
            # the reference's query-time call (one question per call, many worker threads): concurrent calls share forwards
            ids, lens, types = self.tokenize(sentences)
            if ids.shape[0] * ((ids.shape[1] + 31) // 32 * 32) <= 2048:
                res = self._enc.forward(ids, lens, types if use_types else None, normalize=want_norm, shared=True)
                return res[0] if single else res
        for sel, (ids, lens, types) in self._batches(sentences, batch_size):
            for a, b, L in self._token_slices(lens):
                out[sel[a:b]] = self._enc.forward(ids[a:b, :L], lens[a:b], types[a:b, :L] if use_types else None,
                                                  normalize=want_norm)
        return out[0] if single else out

    def encode_device(self, sentences, batch_size: int = 32, normalize_embeddings: bool = False):
        """``encode`` with the [n, hidden] float32 result left on the device (an ``anorag_hip.fusion.DeviceArray``, rows
        in input order): what ``FlatIndex.add_device`` / ``search_device`` consume — no host round trip per batch"""
        from .fusion import DeviceArray
        n = len(sentences)
        out = DeviceArray(n, self._enc.hidden, np.float32, self.device)
        want_norm = bool(normalize_embeddings) or self._info["normalize_module"]
        use_types = self._info["hf"].get("type_vocab_size", 1) > 1
        for sel, (ids, lens, types) in self._batches(sentences, batch_size):
            for a, b, L in self._token_slices(lens):
                self._enc.forward_device(ids[a:b, :L], lens[a:b], types[a:b, :L] if use_types else None, want_norm,
                                         out.ptr, out_rows=sel[a:b])
        return out

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=False)
            self._pool = None
        self._enc.close()
