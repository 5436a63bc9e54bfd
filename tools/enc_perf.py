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
SHAPE = os.environ.get("SHAPE", "bge-base")  # or bge-m3 (XLM-R large: the reference's default model), minilm
LAYERS, H, HEADS, I = {"bge-base": (12, 768, 12, 3072), "bge-m3": (24, 1024, 16, 4096), "minilm": (6, 384, 12, 1536)}[SHAPE]
md = oenc.make_synthetic_model(os.path.join(tempfile.mkdtemp(), SHAPE + "-synth"), layers=LAYERS, hidden=H, heads=HEADS,
                               intermediate=I, vocab=30522, max_pos=512, pooling="cls", weight_std=0.03,
                               model_type="xlm-roberta" if SHAPE == "bge-m3" else "bert")
enc = SentenceEncoder(md)
ids = np.random.default_rng(0).integers(5, 30000, size=(B, L)).astype(np.int32)
lens = np.full((B,), L, dtype=np.int32)
enc._enc.forward(ids, lens, np.zeros_like(ids), normalize=True)
t0 = time.perf_counter()
n = 10
for _ in range(n): enc._enc.forward(ids, lens, np.zeros_like(ids), normalize=True)
dt = (time.perf_counter() - t0) / n
T = B * L
flops = LAYERS * (2 * T * (4 * H * H + 2 * H * I)) + LAYERS * 4 * B * L * L * H
print(f"{SHAPE} B={B} L={L} tokens={T}: {dt*1e3:.3f} ms/forward  {flops/dt/1e12:.1f} TFLOP/s  {B/dt:.0f} seq/s")
