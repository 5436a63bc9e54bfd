#!/usr/bin/env python3
"""Developer tool: time the HIP encoder on the bge-base-en shape (seeded weights)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import encoder as oenc
from anorag_hip.encoder import SentenceEncoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
md = oenc.make_synthetic_model(os.path.join(tempfile.mkdtemp(), "bge-base-synth"), layers=12, hidden=768, heads=12,
                               intermediate=3072, vocab=30522, max_pos=512, pooling="cls", weight_std=0.03)
enc = SentenceEncoder(md)
ids = np.random.default_rng(0).integers(5, 30000, size=(B, L)).astype(np.int32)
lens = np.full((B,), L, dtype=np.int32)
enc._enc.forward(ids, lens, np.zeros_like(ids), normalize=True)
t0 = time.perf_counter()
n = 10
for _ in range(n): enc._enc.forward(ids, lens, np.zeros_like(ids), normalize=True)
dt = (time.perf_counter() - t0) / n
T = B * L
flops = 12 * (2 * T * (4 * 768 * 768 + 2 * 768 * 3072)) + 12 * 4 * B * L * L * 768
print(f"B={B} L={L} tokens={T}: {dt*1e3:.3f} ms/forward  {flops/dt/1e12:.1f} TFLOP/s  {B/dt:.0f} seq/s")
