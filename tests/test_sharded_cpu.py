"""CPU (gloo, world_size 2) test of the row-sharded search orchestration: partition, local top-k, id offsets,
all-gather of the partials and the host merge reproduce the single-index oracle result.  The per-shard search
is the oracle here (no GPU in this container); on the GPU the same class wraps FlatIndex.search."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from anorag_hip.sharded import ShardedSearcher, merge_topk_host, shard_bounds
from oracle import flat_index as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, d, nq, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = orc.preprocess_vectors(np.random.default_rng(1234).standard_normal((n, d), dtype=np.float32))
    x[n // 2 + 3] = x[5]   # a duplicate row straddling the shard boundary: tie must go to the lower global id
    q = orc.preprocess_vectors(np.random.default_rng(4321).standard_normal((nq, d), dtype=np.float32))
    lo, hi = shard_bounds(n, world, rank)
    shard = x[lo:hi]
    s = ShardedSearcher(lambda qq, kk: orc.flat_search(qq, shard, kk, "ip"), lo, True)
    D, I = s.search(q, k)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), D=D, I=I)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nq,k", [(9, 20), (1, 5), (3, 7)])   # odd nq * k: the id block of the packed exchange buffer
def test_sharded_search_matches_single_index(tmp_path, nq, k):  # must stay 8-byte aligned and the part stride exact
    n, d, world = 5001, 64, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, d, nq, k, str(tmp_path)), nprocs=world, join=True)
    x = orc.preprocess_vectors(np.random.default_rng(1234).standard_normal((n, d), dtype=np.float32))
    x[n // 2 + 3] = x[5]
    q = orc.preprocess_vectors(np.random.default_rng(4321).standard_normal((nq, d), dtype=np.float32))
    Dr, Ir = orc.flat_search(q, x, k, "ip")
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert np.array_equal(got["I"], Ir)
        assert np.array_equal(got["D"], Dr)


def test_packed_layout_alignment():
    from anorag_hip.sharded import packed_layout
    for nres in (1, 5, 7, 21, 100, 6400, 6401):
        id_off, part = packed_layout(nres)
        assert id_off % 8 == 0 and part % 8 == 0 and id_off >= nres * 4 and part == id_off + nres * 8
        assert id_off - nres * 4 < 8


def test_shard_bounds_and_merge_edges():
    assert [shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard_bounds(2, 4, 3) == (2, 2)          # empty trailing shard
    # padding (-1) sorts last, ties by id, L2 order
    Dp = np.array([[[0.9, 0.5, -orc.FLT_MAX]], [[0.9, 0.7, 0.1]]], dtype=np.float32)
    Ip = np.array([[[7, 2, -1]], [[3, 11, 12]]], dtype=np.int64)
    D, I = merge_topk_host(Dp, Ip, 4, True)
    assert I.tolist() == [[3, 7, 11, 2]] and D.tolist()[0][:2] == [np.float32(0.9)] * 2
    D, I = merge_topk_host(Dp[:, :, :2], Ip[:, :, :2], 6, False)
    assert I.tolist() == [[2, 11, 3, 7, -1, -1]] and D[0, 4] == orc.FLT_MAX


@pytest.mark.parametrize("lag,group", [(2, 1), (2, 2), (2, 3), (2, 4), (0, 1), (1, 2), (3, 2)])
def test_exchange_plan_groups_are_contiguous_and_slots_are_never_overwritten_early(lag, group):
    """bench.py's N > 1 loop: every batch is exchanged exactly once, a group's slots are consecutive and start on a
    multiple of the group size (one contiguous send buffer), a slot is handed out again only after the group that last
    used it has been returned, and at most lag + group - 1 batches are owed after any step"""
    from anorag_hip.sharded import ExchangePlan
    rng = np.random.default_rng(lag * 10 + group)
    plan = ExchangePlan(lag, group)
    assert plan.nslot % group == 0 and plan.nslot >= lag + group
    batch = 0
    for run in range(6):
        n = int(rng.integers(0, 40))
        owner = {}            # slot -> batch whose result still sits in it (not yet exchanged)
        exchanged = []
        issued_here = []

        def take(grp):
            assert 1 <= len(grp) <= group
            slots = [s for _, s in grp]
            assert slots == list(range(slots[0], slots[0] + len(grp))) and slots[0] % group == 0
            assert slots[-1] < plan.nslot
            for b, s in grp:
                assert owner.pop(s) == b
                exchanged.append(b)

        for _ in range(n):
            slot, grp = plan.issue(batch)
            assert slot not in owner, "a slot was handed out while its previous batch was still owed"
            owner[slot] = batch
            issued_here.append(batch)
            batch += 1
            if grp is not None:
                assert len(grp) == group
                take(grp)
            assert len(plan.pending) <= lag + group - 1
        for grp in plan.drain():
            take(grp)
        assert exchanged == issued_here and not owner and not plan.pending and plan.issued == 0


def _stream_worker(rank, world, port, n, d, nq, k, n_batches, group, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = orc.preprocess_vectors(np.random.default_rng(1234).standard_normal((n, d), dtype=np.float32))
    x[n // 2 + 3] = x[5]
    lo, hi = shard_bounds(n, world, rank)
    shard = x[lo:hi]
    s = ShardedSearcher(lambda qq, kk: orc.flat_search(qq, shard, kk, "ip"), lo, True)
    stream = s.stream(nq, k, lag=2, group=group)
    qs = np.random.default_rng(99).standard_normal((n_batches, nq, d), dtype=np.float32)
    got, order = {}, []
    for b in range(n_batches):
        for tag, D, I in stream.submit(orc.preprocess_vectors(qs[b]), tag=b):
            got[tag] = (D.copy(), I.copy())
            order.append(tag)
        assert len(stream.plan.pending) <= 2 + group - 1
    for tag, D, I in stream.flush():
        got[tag] = (D.copy(), I.copy())
        order.append(tag)
    assert order == list(range(n_batches))            # every batch once, in submission order
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), D=np.stack([got[b][0] for b in range(n_batches)]),
             I=np.stack([got[b][1] for b in range(n_batches)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_batches,group", [(7, 2), (5, 3), (4, 1), (1, 2)])
def test_sharded_stream_groups_the_exchange_and_matches_the_single_index(tmp_path, n_batches, group):
    """ShardedStream (bench.py's N > 1 loop): `group` batches per all-gather, a short last group, results per batch
    identical to the single-index oracle on both ranks"""
    n, d, world, nq, k = 3001, 48, 2, 5, 9
    port = _free_port()
    mp.spawn(_stream_worker, args=(world, port, n, d, nq, k, n_batches, group, str(tmp_path)), nprocs=world, join=True)
    x = orc.preprocess_vectors(np.random.default_rng(1234).standard_normal((n, d), dtype=np.float32))
    x[n // 2 + 3] = x[5]
    qs = np.random.default_rng(99).standard_normal((n_batches, nq, d), dtype=np.float32)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"s{r}.npz"))
        for b in range(n_batches):
            Dr, Ir = orc.flat_search(orc.preprocess_vectors(qs[b]), x, k, "ip")
            assert np.array_equal(got["I"][b], Ir) and np.array_equal(got["D"][b], Dr)
