#!/usr/bin/env python3
"""Developer tool: ONE query at a time through the drop-in classes — the reference's own calling pattern
(query/query_processor.py encodes and searches per question): EmbeddingManager.encode_queries([text]) ->
VectorIndex.search(embedding, top_k) on a 20 000-note, 768-d corpus, bge-base-en shape with seeded weights.
Beside it the CPU path: the float32 oracle (transformers forward on the host cores) + numpy search."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import encoder as oenc, flat_index as orc
from anorag_hip import compat
from vector_store import EmbeddingManager
from vector_store.vector_index import VectorIndex

md = oenc.make_synthetic_model(os.path.join(tempfile.mkdtemp(), "bge-base-synth"), layers=12, hidden=768, heads=12,
                               intermediate=3072, vocab=30522, max_pos=512, pooling="cls", weight_std=0.03)
cfg = compat.config
cfg.set("embedding.model_path", md); cfg.set("embedding.max_length", 512); cfg.set("vector_store.dimension", 768)
EmbeddingManager._reset_singleton()
em = EmbeddingManager()
N, K = 20_000, 20
x = np.random.default_rng(1).standard_normal((N, 768), dtype=np.float32)
vi = VectorIndex(768); vi.create_index(); vi.add_vectors(x)
qs = oenc.synthetic_sentences(md, 400, seed=9, min_words=6, max_words=18)
def one(q):
    e = em.encode_queries([q])
    return e, vi.search(e, top_k=K)
for q in qs[:300]: one(q)
te = ts = 0.0
for q in qs[300:]:
    t0 = time.perf_counter(); e = em.encode_queries([q]); t1 = time.perf_counter(); r = vi.search(e, top_k=K); t2 = time.perf_counter()
    te += t1 - t0; ts += t2 - t1
n = len(qs) - 300
print(f"device: encode_queries {1e6*te/n:.0f} us + VectorIndex.search(top_k={K}, {N} x 768) {1e6*ts/n:.0f} us = {1e6*(te+ts)/n:.0f} us per query")
# the reference's worker threads (main_musique.py:487-494): 8 threads, one question per call, sharing the model — the
# encoder's combining queue merges the calls that arrive while a forward runs
import threading
def burst(nthreads, work):
    def w(k):
        for q in work[k::nthreads]:
            one(q)
    th = [threading.Thread(target=w, args=(k,)) for k in range(nthreads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0
work = qs * 2
burst(8, work[:160])
t1 = burst(1, work)
t8 = burst(8, work)
f0, s0 = em.model._enc.shared_stats()
print(f"threads: encode_queries + search, {len(work)} questions: 1 thread {1e6*t1/len(work):.0f} us per question, 8 threads "
      f"{1e6*t8/len(work):.0f} us per question ({t1/t8:.1f}x); {s0/max(1,f0):.1f} questions per forward over the run")
xn = orc.preprocess_vectors(x)
prefix = "Represent this sentence for searching relevant passages: "
# the host path with the model loaded ONCE (oracle.encoder.encode reloads it per call): the same float32 pipeline
import torch
torch.set_num_threads(int(os.environ.get("ANORAG_BENCH_THREADS", "16")))  # the box's CPU share; 128 threads were 10x slower
from tokenizers import BertWordPieceTokenizer
from transformers import AutoModel
model = AutoModel.from_pretrained(md, add_pooling_layer=False).eval().float()
tok = BertWordPieceTokenizer(os.path.join(md, "vocab.txt"), lowercase=True)
def host_encode(text):
    e = tok.encode(text)
    ids = torch.tensor([e.ids]); mask = torch.ones_like(ids); types = torch.zeros_like(ids)
    with torch.no_grad():
        h = model(input_ids=ids, attention_mask=mask, token_type_ids=types).last_hidden_state
    return torch.nn.functional.normalize(h[:, 0], p=2, dim=1).numpy()
for q in qs[300:305]: host_encode(prefix + q)
t0 = time.perf_counter()
for q in qs[300:330]: ee = host_encode(prefix + q)
t1 = time.perf_counter()
for q in qs[300:330]:
    s = ee @ xn.T; part = np.argpartition(-s, K - 1, axis=1)[:, :K]; np.argsort(-np.take_along_axis(s, part, 1), axis=1)
t2 = time.perf_counter()
print(f"host  : float32 transformers forward {1e6*(t1-t0)/30:.0f} us + numpy search {1e6*(t2-t1)/30:.0f} us per query "
      f"(torch threads {torch.get_num_threads()}, {os.cpu_count()} logical CPUs)")
got = em.encode_queries([qs[329]])
print(f"cosine of the two embeddings of the last query: {float(np.sum(got[0] * ee[0])):.7f}")
