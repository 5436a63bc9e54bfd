#!/usr/bin/env python3
"""Developer tool: per-ROLE kernel times of the encoder forward from a rocprofv3 kernel trace (the GEMM template
instantiations are shared by several roles: QK / out-proj / FFN-down are all k_gemm_pp<0,6>): dispatches are taken in
order and labelled by their position in the layer's sequence."""
import csv, glob, os, sys
from collections import defaultdict
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = ["qk", "v", "attn", "out", "ln1", "up", "down", "ln2"]
FUSED = ["qkv+attn", "out", "ln1", "up", "down", "ln2"]  # layers whose projections + attention ran as k_qkv_attn
t = defaultdict(list)
i = 0
names = [r["Kernel_Name"] for r in rows]
k = 0
while k < len(rows):
    n = names[k]
    if "k_embed_ln" in n:
        i = 0
        t["embed"].append(int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"]))
    elif "k_pool" in n:
        t["pool"].append(int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"]))
    elif "k_qkv_attn" in n:
        seq = FUSED
        i = 0
        t[seq[0]].append(int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"]))
        i = 1
    elif any(x in n for x in ("k_gemm", "k_attention", "k_layernorm")):
        t[seq[i % len(seq)]].append(int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"]))
        i += 1
    k += 1
tot = 0.0
for r in seq:
    if not t[r]:
        continue
    v = sorted(t[r]); m = v[len(v) // 2] / 1e3
    tot += m
    print(f"{r:6s} n={len(v):4d} median {m:7.1f} us  mean {sum(v)/len(v)/1e3:7.1f}")
print(f"layer sum of medians {tot:.1f} us")
