#!/usr/bin/env python3
"""Developer tool: the combining queue of anr_encoder_forward_shared WITHOUT the interpreter's share — T threads call the
C entry point with pre-tokenised single queries (one ctypes call per query, the interpreter lock released inside it), against
the same queries one after the other through anr_encoder_forward.  bge-base shape, seeded weights."""
import os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import encoder as oenc
from anorag_hip.encoder import SentenceEncoder
md = oenc.make_synthetic_model(os.path.join(tempfile.mkdtemp(), "bge-base-synth"), layers=12, hidden=768, heads=12,
                               intermediate=3072, vocab=30522, max_pos=512, pooling="cls", weight_std=0.03)
enc = SentenceEncoder(md)
qs = oenc.synthetic_sentences(md, 960, seed=9, min_words=6, max_words=18)
toks = [enc.tokenize([q])[:2] for q in qs]
for ids, lens in toks[:50]:
    enc._enc.forward(ids, lens, None, normalize=True)
t0 = time.perf_counter()
serial = [enc._enc.forward(ids, lens, None, normalize=True) for ids, lens in toks]
t_serial = (time.perf_counter() - t0) / len(toks)
print(f"lanes {os.environ.get('ANORAG_ENC_LANES', '2')}: one thread, anr_encoder_forward: {1e3 * t_serial:.3f} ms per query")
for T in (2, 4, 8, 16):
    got = [None] * len(toks)
    def worker(w):
        for j in range(w, len(toks), T):
            got[j] = enc._enc.forward(toks[j][0], toks[j][1], None, normalize=True, shared=True)
    f0, s0 = enc._enc.shared_stats()
    th = [threading.Thread(target=worker, args=(w,)) for w in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = (time.perf_counter() - t0) / len(toks)
    f1, s1 = enc._enc.shared_stats()
    same = all(np.array_equal(a, b) for a, b in zip(got, serial))
    print(f"  {T:2d} threads, anr_encoder_forward_shared: {1e3 * dt:.3f} ms per query ({t_serial / dt:.1f}x), "
          f"{(s1 - s0) / max(1, f1 - f0):.1f} queries per forward, bit-identical: {same}")
enc.close()
