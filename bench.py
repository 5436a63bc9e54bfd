#!/usr/bin/env python3
"""bench.py — headline benchmark of the dense-retrieval hot path (BASELINE.json `metric`).

Workload: exact top-100 of batch-64 queries over a 10M x 768 synthetic corpus (unit-norm Gaussian rows,
cosine == inner product), the configuration BASELINE.json's metric is quoted on.  One "step" is one
batch of 64 queries answered end to end (query prep, scan, select, exact re-score, merge).

  python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1: the whole corpus sits on one MI355X (fp32 rows + blocked f16 image = 46 GB).
N > 1 (launched by torch.distributed.run, one rank per GPU): the corpus is row-sharded, every rank
scans its shard for the same batch, the [64,100] partial top-k (score f32 + global id i64) are
all-gathered over RCCL and merged — each batch only after anr_index_wait() has made it final on its shard
(certificate recovery included); strong scaling (total corpus fixed).

Rank 0 prints ONE JSON line.  `value` = queries/s of the whole job with the corpus resident in HBM.
`roofline` is for the dominant kernel (k_scan): algorithmic bytes = rows x 768 x 2 B (the f16 image
actually streamed) per launch / HIP-event time of that launch.  `cpu_baseline` is the oracle (numpy
sgemm + argpartition restatement of the reference's faiss-flat path) timed on this box's host cores on
a bounded row sample and scaled linearly to the full corpus.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ano-rag_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=10_000_000, help="total corpus rows (all ranks)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--cpu-rows", type=int, default=200_000, help="rows of the CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--recall-queries", type=int, default=-1,
                    help="queries of the last batch checked against the oracle (-1 = the whole batch, 0 = skip)")
    ap.add_argument("--no-facade", action="store_true", help="skip the VectorIndex.search (host in, dicts out) leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsal)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on one GPU: run the N > 1 pipeline (process group, all-gather, merge) with world size 1")
    return ap.parse_args()


def gen_shard(rows, dim, shard, device, chunk=262_144):
    """unit-norm-able Gaussian rows, generated on the device chunk by chunk (torch Philox, seed 1234000 + shard:
    a 30 GB host array from numpy's default_rng([1234, s]) would take minutes; same distribution, other stream)."""
    g = torch.Generator(device=device)
    g.manual_seed(1234 * 1000 + shard)
    done = 0
    while done < rows:
        m = min(chunk, rows - done)
        yield torch.randn((m, dim), generator=g, device=device, dtype=torch.float32)
        done += m


def cpu_baseline(args, world):
    """oracle timed on host cores: bounded sample, scaled linearly in rows."""
    from oracle import flat_index as orc
    try:
        from threadpoolctl import threadpool_info
        thr = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        thr = os.cpu_count() or 1
    n = args.cpu_rows
    x = orc.preprocess_vectors(np.random.default_rng(1234).standard_normal((n, args.dim), dtype=np.float32))
    q = orc.preprocess_vectors(np.random.default_rng(4321).standard_normal((args.batch, args.dim), dtype=np.float32))

    def one():
        s = q @ x.T
        part = np.argpartition(-s, args.k - 1, axis=1)[:, :args.k]
        ps = np.take_along_axis(s, part, axis=1)
        order = np.argsort(-ps, axis=1, kind="stable")
        return np.take_along_axis(part, order, axis=1)

    one()
    t0 = time.perf_counter()
    it = 0
    while True:
        one()
        it += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or it >= 50:
            break
    t_batch = dt / it
    qps_sample = args.batch / t_batch
    qps_full = qps_sample * n / args.rows
    return {
        "value": qps_full,
        "unit": "queries/s",
        "cores": int(thr),
        "kind": "port",
        "sample": (f"numpy fp32 sgemm + argpartition top-{args.k}, batch {args.batch}, {n} x {args.dim} rows: "
                   f"{qps_sample:.1f} q/s measured over {it} batches, scaled x{n / args.rows:.4g} to {args.rows} rows; "
                   f"os.cpu_count()={os.cpu_count()}"),
    }


def oracle_partial_topk(args, shards_host_iter, q_host, row0):
    """float64-arbitrated top-k of the given queries over this rank's rows (global ids): (scores [nq,k], ids [nq,k])"""
    from oracle import flat_index as orc
    top = orc.BlockedTopK(orc.preprocess_vectors(q_host), args.k, "ip")
    base = row0
    for xb in shards_host_iter():
        top.push(orc.preprocess_vectors(xb), base)
        base += xb.shape[0]
    return top.result()


def recall_from_partials(parts, I_gpu, k):
    s = np.concatenate([p[0] for p in parts], axis=1)
    i = np.concatenate([p[1] for p in parts], axis=1)
    o = np.argsort(-s, axis=1, kind="stable")[:, :k]
    ref = np.take_along_axis(i, o, axis=1)
    hits = sum(len(set(ref[r].tolist()) & set(I_gpu[r].tolist())) for r in range(ref.shape[0]))
    return hits / float(ref.shape[0] * k)


def pmc_traffic(rows_per_gpu, dim, notes):
    """HBM bytes per k_scan launch from the committed PMC passes (profiles/*_pmc_traffic_k_scan.json: FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE, separate --pmc runs), scaled by rows when the
    shard differs.  The file records the sha256 of csrc/index_kernels.hpp it was measured on: when the kernel source
    has changed since, the figure is stale and is NOT reported (traffic = null, the reason goes to `traffic_note` and
    stderr) — re-run tools/refresh_profiles.sh."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_k_scan.json")))
    if not files:
        notes.append("no profiles/*_pmc_traffic_k_scan.json")
        return None
    with open(files[-1]) as f:
        p = json.load(f)
    src = os.path.join(ROOT, "ano-rag_amd", "csrc", "index_kernels.hpp")
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()
    if p.get("kernel_source_sha256") != sha:
        msg = (f"{os.path.basename(files[-1])} was measured on another version of index_kernels.hpp "
               f"(recorded {str(p.get('kernel_source_sha256'))[:12]}, current {sha[:12]}): traffic not reported")
        print("bench.py: " + msg, file=sys.stderr)
        notes.append(msg)
        return None
    if p["dim"] != dim:
        notes.append("PMC profile taken at another dim")
        return None
    notes.append(os.path.basename(files[-1]))
    return p["traffic_bytes_per_launch"] * rows_per_gpu / p["rows"]


def facade_leg(idx, args, q_host):
    """The reference-facing call: VectorIndex.search(np.ndarray[B, D], top_k) -> list of per-query lists of dicts
    (reference vector_store/vector_index.py:206-263), timed end to end on the same resident corpus."""
    from vector_store.vector_index import VectorIndex
    vi = VectorIndex(embedding_dim=args.dim)
    vi.index_type, vi.similarity_metric = "Flat", "cosine"
    vi.index, vi.is_trained, vi.total_vectors = idx, True, idx.ntotal  # the corpus already on the device
    for _ in range(3):
        res = vi.search(q_host, top_k=args.k)
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        res = vi.search(q_host, top_k=args.k)
    dt = (time.perf_counter() - t0) / n
    ok = len(res) == args.batch and all(len(r) == args.k for r in res) and set(res[0][0]) == {"index", "score", "rank", "similarity"}
    vi.index = None  # the bench owns the handle
    return {"call": f"VectorIndex.search(np.float32[{args.batch},{args.dim}], top_k={args.k}) -> list[list[dict]]",
            "ms_per_call": dt * 1e3, "value": args.batch / dt, "unit": "queries/s", "calls": n, "shape_ok": bool(ok),
            "note": "host numpy in, Python dicts out, synchronous, one batch at a time (not `value`)"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    dist = None
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from anorag_hip import FlatIndex, METRIC_IP, _lib
    from anorag_hip._lib import OPT_TIMING

    rows_total = args.rows
    per = (rows_total + world - 1) // world
    row0 = min(rank * per, rows_total)
    my_rows = max(0, min(per, rows_total - row0))

    idx = FlatIndex(args.dim, METRIC_IP, normalize=True, device=local_rank)
    idx.reserve(my_rows)
    for xb in gen_shard(my_rows, args.dim, rank, dev):
        torch.cuda.synchronize()
        idx.add_device(xb.data_ptr(), xb.shape[0])
    del xb
    torch.cuda.empty_cache()
    idx.set_option(OPT_TIMING, 1)

    nb = args.steps + args.warmup
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    Q = torch.randn((nb, args.batch, args.dim), generator=gq, device=dev, dtype=torch.float32)
    # steps are issued asynchronously and overlap (the small kernels of neighbouring batches run beside the scan):
    # NSLOT rotating sets of output buffers and streams
    NSLOT = 3
    streams = [torch.cuda.Stream(device=dev) for _ in range(NSLOT)]
    # per slot one packed result buffer [B*k f32 | B*k i64]: the index writes both halves, and for N > 1 the
    # partial top-k lists of all ranks travel in ONE all-gather
    nres = args.batch * args.k
    Pl = [torch.empty(nres * 12, device=dev, dtype=torch.uint8) for _ in range(NSLOT)]
    Dl = [p[: nres * 4].view(torch.float32).view(args.batch, args.k) for p in Pl]
    Il = [p[nres * 4:].view(torch.int64).view(args.batch, args.k) for p in Pl]
    if dist_on:
        from anorag_hip._lib import OPT_ID_OFFSET
        idx.set_option(OPT_ID_OFFSET, row0)  # the shard returns global ids (no -1 padding: every shard holds >= k rows)
        Pg = [torch.empty(world * nres * 12, device=dev, dtype=torch.uint8) for _ in range(NSLOT)]
        Dm = [torch.empty_like(Dl[0]) for _ in range(NSLOT)]
        Im = [torch.empty_like(Il[0]) for _ in range(NSLOT)]
    lib = _lib.load()
    torch.cuda.synchronize()

    # N > 1: the exchange of batch i runs only once batch i is FINAL on its shard (anr_index_wait: certificate
    # recovery done), LAG batches behind the search front so the device never idles on the host
    LAG = NSLOT - 1
    pending = []
    merged = {}

    def exchange(i):
        s = i % NSLOT
        st = streams[s]
        idx.wait(len(pending))  # everything older than the still-pending batches is final
        with torch.cuda.stream(st):
            if args.backend == "nccl":
                dist.all_gather_into_tensor(Pg[s], Pl[s])
            else:  # rehearsal path (gloo has no all_gather_into_tensor for device tensors)
                st.synchronize()
                ph = [torch.empty(nres * 12, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(ph, Pl[s].cpu())
                Pg[s].copy_(torch.cat(ph))
            _lib.check(lib.anr_merge_topk_strided_dev(
                local_rank, C.c_void_p(Pg[s].data_ptr()), C.c_void_p(Pg[s].data_ptr() + nres * 4),
                nres * 3, (nres * 3) // 2, world, args.batch, args.k, 1,
                C.c_void_p(Dm[s].data_ptr()), C.c_void_p(Im[s].data_ptr()),
                C.c_void_p(st.cuda_stream)), "anr_merge_topk_strided_dev")
        merged["last"] = (Dm[s], Im[s])

    def step(i):
        s = i % NSLOT
        idx.search_device_async(Q[i].data_ptr(), args.batch, args.k, Dl[s].data_ptr(), Il[s].data_ptr(),
                                streams[s].cuda_stream)
        if dist_on:
            pending.append(i)
            if len(pending) > LAG:
                exchange(pending.pop(0))
        else:
            merged["last"] = (Dl[s], Il[s])

    def finish():
        idx.sync()              # retires every batch; runs the exact path where a certificate failed
        while pending:
            exchange(pending.pop(0))
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    finish()
    idx.reset_stats()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, nb):
        step(i)
    finish()
    Dres, Ires = merged["last"]
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st_all = idx.last_stats()
    scan_ms, scan_bytes = st_all["scan_ms"], st_all["scan_bytes"]
    n_fallback, n_cand = st_all["n_fallback"], st_all["n_candidates"]
    if dist_on:
        t = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the reference-facing call is timed BEFORE the CPU legs: after the oracle's 10 M-row BLAS passes (recall) the same
    # loop ran 2.3x slower on two boxes (6.4 vs 2.8 ms per call) — host threads left spinning, nothing on the device
    facade = None
    if rank == 0 and world == 1 and not args.no_facade:
        facade = facade_leg(idx, args, Q[nb - 1].cpu().numpy())

    # recall@k of the last batch's first few queries vs the oracle: every rank ranks its own rows on the CPU
    # (float64), rank 0 merges the partial lists and compares with the ids the GPU path returned
    recall = None
    if args.recall_queries != 0:
        nrq = args.batch if args.recall_queries < 0 else min(args.recall_queries, args.batch)
        I_gpu = Ires[:nrq].cpu().numpy()
        qh = Q[nb - 1, :nrq].cpu().numpy()

        def shards():
            for xb in gen_shard(my_rows, args.dim, rank, dev, chunk=524_288):
                yield xb.cpu().numpy()

        part = oracle_partial_topk(args, shards, qh, row0)
        if dist_on:
            parts = [None] * world
            dist.all_gather_object(parts, part)
        else:
            parts = [part]
        if rank == 0:
            recall = recall_from_partials(parts, I_gpu, args.k)

    if rank == 0:
        tnotes = []
        traffic = pmc_traffic(per, args.dim, tnotes)
        qps = args.batch * args.steps / dt
        achieved = (scan_bytes / 1e9) / (scan_ms / 1e3) if scan_ms > 0 else None
        out = {
            "metric": "queries/sec + recall@k vs CPU ref, 10M×768 corpus, batch-64 top-100",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16 scan (MFMA, f32 accumulate) + f32 rows re-scored in f64",
            "data": "synthetic",
            "config": {
                "workload": f"{rows_total} x {args.dim} unit-norm Gaussian corpus, batch-{args.batch} top-{args.k} "
                            f"exact cosine search",
                "rows_total": rows_total,
                "rows_per_gpu": per,
                "dim": args.dim,
                "batch": args.batch,
                "k": args.k,
                "parallelism": f"row-shard x{world}" if world > 1 else "single GPU",
            },
            "recall_at_k": recall,
            "recall_queries": (args.batch if args.recall_queries < 0 else args.recall_queries),
            "exact_fallback_queries": n_fallback,
            "candidates_per_query": n_cand / max(1, args.steps * args.batch),
            "roofline": {
                "bound": "hbm",
                "kernel": "k_scan",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                "traffic": traffic,
                "traffic_note": "; ".join(tnotes),
                "bytes_per_launch": scan_bytes / max(1, args.steps),
                "ms_per_launch": scan_ms / max(1, args.steps),
            },
        }
        if facade is not None:
            out["facade"] = facade
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, world)
        print(json.dumps(out), flush=True)
    idx.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
