"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of what the reference gets from ``self.model.encode(...)`` (vector_store/embedding_manager.py:
392-399).  The arithmetic lives in third-party code that is not vendored in the reference and not installed
here: sentence-transformers (>= 2.2.0, requirements.txt:15) on top of transformers / torch.  Restated from
its published pipeline: texts sorted by length (longest first) -> batches -> HF
tokenizer (padding to the longest, truncation to max_seq_length) -> ``AutoModel`` forward (here
``transformers.BertModel`` / ``XLMRobertaModel`` / ``MPNetModel`` in float32 on the CPU, which IS the code sentence-transformers
calls) -> Pooling (masked mean: ``sum(h * mask) / clamp(sum(mask), 1e-9)``, or CLS) -> ``F.normalize(p=2, dim=1)``.

PARITY UNPINNED for real checkpoints: no model weights / vocab exist in the container and the reference has no
test or golden vector for this call; parity is checked on seeded random weights of the
reference's model shapes with a synthetic WordPiece vocabulary.
"""
from __future__ import annotations

import json
import os
from typing import List

import numpy as np


def make_synthetic_model(path: str, *, layers=6, hidden=384, heads=12, intermediate=1536, vocab=2048, max_pos=512,
                         pooling="mean", seed=0, weight_std=0.05, model_type="bert", outlier_dims=0,
                         outlier_gain=30.0) -> str:
    """Write an HF/sentence-transformers style model directory with seeded random weights.
    outlier_dims > 0: that many hidden dimensions carry OUTLIER FEATURES, as trained BERT-family checkpoints do — every
    LayerNorm's gain on them is multiplied by outlier_gain (x 20-40) and the word-embedding columns likewise, so their
    activations reach the tens to hundreds while the rest stay O(1) (the stress case for an f16 residual stream)."""
    import torch
    from transformers import BertConfig, BertModel, MPNetConfig, MPNetModel, XLMRobertaConfig, XLMRobertaModel

    os.makedirs(path, exist_ok=True)
    torch.manual_seed(seed)
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    rng = np.random.default_rng(seed)
    alphabet = "abcdefghijklmnopqrstuvwxyz"
    seen = set(words)
    while len(words) < vocab:
        w = "".join(rng.choice(list(alphabet), size=rng.integers(1, 7)))
        if rng.random() < 0.3:
            w = "##" + w
        if w not in seen:
            seen.add(w)
            words.append(w)
    with open(os.path.join(path, "vocab.txt"), "w") as f:
        f.write("\n".join(words) + "\n")
    if model_type == "bert":
        cfg = BertConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                         intermediate_size=intermediate, max_position_embeddings=max_pos, type_vocab_size=2,
                         initializer_range=weight_std, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        model = BertModel(cfg, add_pooling_layer=False)
    elif model_type == "mpnet":  # all-mpnet-base-v2 family: relative position bias shared by all layers
        cfg = MPNetConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                          intermediate_size=intermediate, max_position_embeddings=max_pos + 2, pad_token_id=1,
                          relative_attention_num_buckets=32, initializer_range=weight_std, hidden_dropout_prob=0.0,
                          attention_probs_dropout_prob=0.0)
        model = MPNetModel(cfg, add_pooling_layer=False)
    else:
        cfg = XLMRobertaConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers,
                               num_attention_heads=heads, intermediate_size=intermediate,
                               max_position_embeddings=max_pos + 2, type_vocab_size=1, pad_token_id=1,
                               initializer_range=weight_std, hidden_dropout_prob=0.0,
                               attention_probs_dropout_prob=0.0)
        model = XLMRobertaModel(cfg, add_pooling_layer=False)
    with torch.no_grad():  # non-trivial LayerNorm parameters and biases
        for n, p in model.named_parameters():
            if "LayerNorm.weight" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.copy_(0.05 * torch.randn_like(p))
        if outlier_dims:
            dims = torch.tensor(np.random.default_rng(seed + 1).choice(hidden, size=int(outlier_dims), replace=False))
            gains = torch.tensor(np.random.default_rng(seed + 2).uniform(0.7, 1.3, int(outlier_dims)) * outlier_gain,
                                 dtype=torch.float32)
            for n, p in model.named_parameters():
                if "LayerNorm.weight" in n:
                    p[dims] *= gains
                elif n.endswith("word_embeddings.weight"):
                    p[:, dims] *= gains
    model.eval()
    model.save_pretrained(path, safe_serialization=True)
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"do_lower_case": True}, f)
    with open(os.path.join(path, "modules.json"), "w") as f:
        json.dump([{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
                   {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
                   {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}], f)
    os.makedirs(os.path.join(path, "1_Pooling"), exist_ok=True)
    with open(os.path.join(path, "1_Pooling", "config.json"), "w") as f:
        json.dump({"word_embedding_dimension": hidden, "pooling_mode_cls_token": pooling == "cls",
                   "pooling_mode_mean_tokens": pooling == "mean", "pooling_mode_max_tokens": False}, f)
    with open(os.path.join(path, "sentence_bert_config.json"), "w") as f:
        json.dump({"max_seq_length": max_pos, "do_lower_case": True}, f)
    return path


def encode(model_dir: str, sentences: List[str], batch_size=32, normalize=True, max_seq_length=None) -> np.ndarray:
    """float32 CPU forward of the model directory through the sentence-transformers pipeline."""
    import torch
    from tokenizers import BertWordPieceTokenizer
    from transformers import AutoModel

    with open(os.path.join(model_dir, "1_Pooling", "config.json")) as f:
        cls_pool = json.load(f).get("pooling_mode_cls_token", False)
    with open(os.path.join(model_dir, "config.json")) as f:
        hf = json.load(f)
    model = AutoModel.from_pretrained(model_dir, add_pooling_layer=False).eval().float()
    tok = BertWordPieceTokenizer(os.path.join(model_dir, "vocab.txt"), lowercase=True)
    max_len = max_seq_length or hf["max_position_embeddings"] - (0 if hf["model_type"] == "bert" else 2)
    tok.enable_truncation(max_length=max_len)
    pad = hf.get("pad_token_id", 0) or 0
    out = np.zeros((len(sentences), hf["hidden_size"]), dtype=np.float32)
    order = np.argsort([-len(s) for s in sentences], kind="stable")
    for s in range(0, len(sentences), batch_size):
        sel = order[s:s + batch_size]
        encs = tok.encode_batch([sentences[i] for i in sel])
        L = max(len(e.ids) for e in encs)
        ids = torch.full((len(encs), L), pad, dtype=torch.long)
        mask = torch.zeros((len(encs), L), dtype=torch.long)
        types = torch.zeros((len(encs), L), dtype=torch.long)
        for i, e in enumerate(encs):
            ids[i, :len(e.ids)] = torch.tensor(e.ids)
            mask[i, :len(e.ids)] = 1
            types[i, :len(e.ids)] = torch.tensor(e.type_ids)
        kw = {"token_type_ids": types} if hf.get("type_vocab_size", 1) > 1 else {}
        with torch.no_grad():
            h = model(input_ids=ids, attention_mask=mask, **kw).last_hidden_state
        if cls_pool:
            emb = h[:, 0]
        else:
            m = mask.unsqueeze(-1).to(h.dtype)
            emb = (h * m).sum(1) / torch.clamp(m.sum(1), min=1e-9)
        if normalize:
            emb = torch.nn.functional.normalize(emb, p=2, dim=1)
        out[sel] = emb.numpy()
    return out


def synthetic_sentences(model_dir: str, n: int, seed=7, min_words=3, max_words=24) -> List[str]:
    with open(os.path.join(model_dir, "vocab.txt")) as f:
        words = [w.strip() for w in f if w.strip() and not w.startswith("[") and not w.startswith("##")]
    rng = np.random.default_rng(seed)
    return [" ".join(rng.choice(words, size=rng.integers(min_words, max_words + 1))) for _ in range(n)]
