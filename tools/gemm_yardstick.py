#!/usr/bin/env python3
"""Developer tool: what the vendor library (hipBLASLt through torch.matmul, f16 in / f32 accumulate) reaches on the encoder's
GEMM shapes — a yardstick for k_gemm_pp, not part of the product (the product uses no BLAS)."""
import sys, time
import torch
dev = torch.device("cuda", 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
shapes = [("qk", 1536, 768), ("v/out", 768, 768), ("ffn-up", 3072, 768), ("ffn-down", 768, 3072),
          ("m3 qk", 2048, 1024), ("m3 ffn-up", 4096, 1024), ("m3 ffn-down", 1024, 4096)]
# the library's best case on this chip (large square-ish GEMMs): the practical ceiling the 2.5 PFLOP/s figure derates to
shapes = [(n, N, K, T) for n, N, K in shapes] + [("8k cube", 8192, 8192, 8192), ("16k x 8k x 8k", 8192, 8192, 16384),
                                               ("self-join block 100k x 100k x 768", 100_096, 768, 100_096 // 2)]
for name, N, K, T in shapes:
    a = torch.randn((T, K), device=dev, dtype=torch.float16)
    w = torch.randn((N, K), device=dev, dtype=torch.float16)
    for _ in range(10): c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:12s} T={T} N={N} K={K}: {us:7.1f} us  {2*T*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
