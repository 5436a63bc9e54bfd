"""Row-sharded exact search over the GPUs of one node: one process per GPU
(``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm), shard g holds the contiguous rows
``[g*ceil(N/P), ...)``, every rank scans its shard for the same query batch, the ``[B, k]`` partial results
(score f32, global id i64 — ~0.6 MB in total for B=64, k=100, P=8) are all-gathered and merged.

Two merge paths with identical results (score order, ties by ascending global id, -1 padding last):
``merge_topk_host`` (numpy; what BASELINE.json's north_star prescribes, also used by the CPU/gloo tests) and
the device kernel ``anr_merge_topk_dev`` used when the partials already sit on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Tuple

import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """rows [lo, hi) of shard `rank`: contiguous, ceil(N/P) rows each, the last shards may be short/empty"""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def merge_topk_host(Dp: np.ndarray, Ip: np.ndarray, k: int, larger_is_better: bool = True):
    """Dp/Ip: [P, B, k] partial lists (global ids, -1 padded).  Returns (D [B,k], I [B,k])."""
    P, B, kk = Dp.shape
    D = np.full((B, k), -FLT_MAX if larger_is_better else FLT_MAX, dtype=np.float32)
    I = np.full((B, k), -1, dtype=np.int64)
    for b in range(B):
        s = Dp[:, b, :].reshape(-1)
        i = Ip[:, b, :].reshape(-1)
        ok = i >= 0
        s, i = s[ok], i[ok]
        key = -s.astype(np.float64) if larger_is_better else s.astype(np.float64)
        order = np.lexsort((i, key))[:k]
        D[b, :len(order)] = s[order]
        I[b, :len(order)] = i[order]
    return D, I


class ShardedSearcher:
    """Distributed front end.  ``local_search(q, k) -> (D, I_local)`` is the rank's own shard search (numpy in /
    numpy out); ids are offset by the shard's first row before the exchange."""

    def __init__(self, local_search: Callable, row_offset: int, larger_is_better: bool = True, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.local_search = local_search
        self.row_offset = int(row_offset)
        self.larger = bool(larger_is_better)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def search(self, q: np.ndarray, k: int):
        import torch
        D, I = self.local_search(q, k)
        I = np.where(I >= 0, I + self.row_offset, -1).astype(np.int64)
        D = np.ascontiguousarray(D, dtype=np.float32)
        if self.world == 1:
            return D, I
        dt, it = torch.from_numpy(D), torch.from_numpy(I)
        dg = [torch.empty_like(dt) for _ in range(self.world)]
        ig = [torch.empty_like(it) for _ in range(self.world)]
        self.dist.all_gather(dg, dt, group=self.group)
        self.dist.all_gather(ig, it, group=self.group)
        return merge_topk_host(torch.stack(dg).numpy(), torch.stack(ig).numpy(), k, self.larger)


def merge_topk_device(device: int, Dg, Ig, k: int, larger_is_better: bool, D_out, I_out, stream: int = 0) -> None:
    """Dg/Ig: torch CUDA tensors [P,B,k]; D_out/I_out [B,k] (same device)."""
    from . import _lib
    lib = _lib.load()
    P, B, kk = Dg.shape
    _lib.check(lib.anr_merge_topk_dev(int(device), C.c_void_p(Dg.data_ptr()), C.c_void_p(Ig.data_ptr()), int(P),
                                      int(B), int(k), int(bool(larger_is_better)), C.c_void_p(D_out.data_ptr()),
                                      C.c_void_p(I_out.data_ptr()), C.c_void_p(stream)), "anr_merge_topk_dev")
