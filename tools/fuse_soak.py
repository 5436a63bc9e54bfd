#!/usr/bin/env python3
"""Developer tool: randomised soak of the N-array fusion — for every random configuration (corpus size, row density,
pool, short-list sizes, weights, method, value kinds) the SPARSE form of the bm25 rows must give exactly what the dense
array gives, and (small corpora, non-negative rows) the dense array must give the oracle's finals.  `fuse_soak.py 60`
runs for 60 s and prints the mismatches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd")); sys.path.insert(0, ROOT)
import numpy as np
from anorag_hip.fusion import DeviceArray, SparseRows, fuse_dense
from oracle import fusion as ofu

t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
n_cfg = n_bad = n_oracle = 0
while time.time() < t_end:
    n_cfg += 1
    n = int(rng.choice([1, 2, 7, 65, 300, 4097, 20_000, 200_000, 1_000_000]))
    nq = int(rng.integers(1, 6))
    cap = int(min(n, rng.choice([1, 3, 64, 1000, 4000, 8192])))
    pool = int(rng.choice([1, 5, 50, 80, 300, 1024]))
    m = int(rng.choice([0, 1, 10, 100, 400, 900]))
    method = "rrf" if rng.random() < 0.5 else "linear"
    kinds = [k for k in ("zeros", "neg", "ties", "nan") if rng.random() < 0.35]
    w = {"dense": float(rng.choice([1.0, 0.3, 0.0, -0.5])), "bm25": float(rng.choice([0.5, 1.0, 0.0, 2.5])),
         "graph": float(rng.choice([0.5, 0.0, 1.5])), "path": float(rng.choice([0.1, 0.0, 1.0]))}
    rows = []
    for q in range(nq):
        nnz = int(rng.integers(0, cap + 1))
        ids = rng.choice(n, size=nnz, replace=False)
        val = np.abs(rng.standard_normal(nnz))
        if "ties" in kinds: val = np.round(val * 4) / 4
        if "zeros" in kinds and nnz: val[rng.random(nnz) < 0.1] = 0.0
        if "neg" in kinds and nnz:
            neg = rng.random(nnz) < 0.2; val[neg] = -val[neg]
        if "nan" in kinds and nnz > 2: val[rng.integers(0, nnz)] = np.nan
        rows.append((ids.astype(np.int64), val))
    def short(mm):
        out = []
        for q in range(nq):
            ids, val = rows[q]
            pick = []
            if len(ids): pick += rng.choice(ids, size=min(len(ids), mm // 2), replace=False).tolist()
            pick += rng.integers(0, n, size=mm - len(pick)).tolist() if mm > len(pick) else []
            pick = list(dict.fromkeys(int(x) for x in pick))[:mm]
            sc = np.sort(rng.standard_normal(len(pick)))[::-1].copy()
            out.append((np.asarray(pick, dtype=np.int64), sc))
        return out
    src = {}
    if m and rng.random() < 0.9: src["dense"] = short(m)
    if m and rng.random() < 0.5: src["graph"] = short(max(1, m // 8))
    if m and rng.random() < 0.3: src["path"] = short(max(1, m // 16))
    if pool + 2 * sum(len(v[0][0]) for v in src.values()) + sum(len(v[0][0]) for v in src.values()) > 4096 or \
       sum(max(len(l[0]) for l in v) for v in src.values()) > 1024:
        continue
    dense_arr = np.zeros((nq, n))
    for q, (ids, val) in enumerate(rows): dense_arr[q, ids] = val
    arr = DeviceArray.from_numpy(dense_arr)
    sp = SparseRows.from_numpy(rows, n, cap=max(cap, 1))
    try:
        a = fuse_dense(method, w, 60.0, pool, nq, {**src, "bm25": arr})
        b = fuse_dense(method, w, 60.0, pool, nq, {**src, "bm25": sp})
    except Exception as e:
        print("config", n_cfg, "raised", repr(e)[:200], dict(n=n, nq=nq, cap=cap, pool=pool, m=m, method=method, kinds=kinds, w=w))
        n_bad += 1
        arr.free(); sp.free()
        continue
    same = (np.array_equal(a[3], b[3]) and np.array_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes()
            and np.array_equal(a[2], b[2], equal_nan=True))
    if not same:
        n_bad += 1
        print("MISMATCH sparse vs dense:", dict(n=n, nq=nq, cap=cap, pool=pool, m=m, method=method, kinds=kinds, w=w))
    if n <= 20_000 and "nan" not in kinds:
        n_oracle += 1
        full = np.arange(n, dtype=np.int64)
        for q in range(nq):
            lists = tuple((src[k][q] if k in src else None) if k != "bm25" else (full, dense_arr[q]) for k in ("dense", "bm25", "graph", "path"))
            ids, fin = ofu.fuse_arrays(n, lists, [w["dense"], w["bm25"], w["graph"], w["path"]], method, 60.0, pool)
            cnt = int(a[3][q])
            if cnt != len(fin) or a[1][q, :cnt].tolist() != fin.tolist():
                n_bad += 1
                print("MISMATCH dense vs oracle:", dict(n=n, nq=nq, q=q, cap=cap, pool=pool, m=m, method=method, kinds=kinds, w=w))
                break
    arr.free(); sp.free()
print(f"done: {n_cfg} configurations ({n_oracle} also against the oracle), {n_bad} bad")
