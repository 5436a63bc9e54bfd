"""Row-sharded exact search over the GPUs of one node.

Shard g holds contiguous rows, every shard scans for the same query batch, and only the ``[B, k]`` partial
results (score f32 + global id i64 — 0.6 MB in total for B=64, k=100, P=8) are exchanged and merged
(score order, ties by ascending global id, -1 padding last).  Two front ends over the same C ABI:

``ShardedFlatIndex``  ONE process drives P handles (one per listed device) — the form the drop-in ``VectorIndex``
    uses when ``anorag_hip.devices`` names several GPUs, because the reference's caller builds one
    ``VectorRetriever`` per process (query/query_processor.py:260-293).  The shard searches run concurrently from a
    thread pool (ctypes releases the GIL; every handle has its own streams), each writes its partial top-k straight
    into pinned host memory, and the partials are merged on the host (``anr_merge_topk_host``) — BASELINE.json
    north_star: "per-shard local top-k merged on the host".

``ShardedSearcher``  one process per GPU under ``torch.distributed`` (bench.py's layout).  With the "nccl" backend
    (RCCL over xGMI) everything stays on the device: the shard writes scores and ids into one packed buffer
    ``[B*k f32 | B*k i64]``, the batch is made final (``FlatIndex.wait`` — certificate recovery included), the
    partials of all ranks travel in ONE ``all_gather_into_tensor`` and ``anr_merge_topk_strided_dev`` merges out of
    the receive buffer.  With "gloo" (CPU rehearsal / tests) the same packed buffer is gathered as a host tensor
    and merged with ``anr_merge_topk_host`` (``merge_topk_host`` is the numpy statement of the same merge).
"""
from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """rows [lo, hi) of shard `rank`: contiguous, ceil(N/P) rows each, the last shards may be short/empty"""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def packed_layout(nres: int) -> Tuple[int, int]:
    """(id_offset, part_bytes) of one rank's packed exchange buffer ``[nres f32 | pad | nres i64]``: the id block
    starts on an 8-byte boundary (nres may be odd: one query with k = 5) and a part is a multiple of 8 bytes, so the
    parts of an all-gather are part_bytes / 4 floats and part_bytes / 8 int64 apart — both exact."""
    id_off = (int(nres) * 4 + 7) // 8 * 8
    return id_off, id_off + int(nres) * 8


class ExchangePlan:
    """Book-keeping of a pipelined search loop whose batches are exchanged (all-gather + merge) in GROUPS, `lag` batches
    behind the search front (bench.py, N > 1): which send-buffer slot a batch writes, and which batches travel together.

    Slots rotate over ``nslot = lag + 2 group`` rounded up to a multiple of `group`; batch number j of a run (counted from
    the last ``drain``) writes slot ``j % nslot``, so every group starts on a multiple of `group` and its slots are
    consecutive: with the slots laid out back to back, a group is ONE contiguous send buffer.  (``lag + group`` slots —
    rounds 2-3 — hand a slot out again ONE submit after its group's collective was launched: the new search then waits for
    an all-gather that only gets a CU when a scan retires, and every other scan started ~60 us late at the 8-GPU shard
    size; a spare group of slots puts a whole group of submits between the two.)  A slot is handed out again
    only after the group that last used it has been returned by ``issue`` / ``drain`` (the caller then orders the new
    write behind that group's collective with an event)."""

    def __init__(self, lag: int = 2, group: int = 2):
        if lag < 0 or group < 1:
            raise ValueError("lag >= 0 and group >= 1")
        self.lag, self.group = int(lag), int(group)
        self.nslot = (self.lag + 2 * self.group + self.group - 1) // self.group * self.group
        self.pending: List[Tuple[int, int]] = []  # (batch, slot) issued and not yet handed back, oldest first
        self.issued = 0

    def issue(self, batch: int) -> Tuple[int, Optional[List[Tuple[int, int]]]]:
        """-> (slot of this batch, the group to exchange now or None)"""
        slot = self.issued % self.nslot
        self.issued += 1
        self.pending.append((batch, slot))
        if len(self.pending) >= self.lag + self.group:
            return slot, [self.pending.pop(0) for _ in range(self.group)]
        return slot, None

    def drain(self):
        """the groups still owed at the end of a run (the last one may be short); the slot counter starts over"""
        while self.pending:
            yield [self.pending.pop(0) for _ in range(min(self.group, len(self.pending)))]
        self.issued = 0


def merge_topk_host(Dp: np.ndarray, Ip: np.ndarray, k: int, larger_is_better: bool = True):
    """numpy statement of the merge.  Dp/Ip: [P, B, k'] partial lists (global ids, -1 padded) -> (D [B,k], I [B,k])."""
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


def merge_topk_host_c(Dp: np.ndarray, Ip: np.ndarray, larger_is_better: bool = True):
    """the library's host merge (anr_merge_topk_host): Dp/Ip [P, B, k] sorted partial lists -> (D [B,k], I [B,k])"""
    from . import _lib
    P, B, k = Dp.shape
    Dp = np.ascontiguousarray(Dp, dtype=np.float32)
    Ip = np.ascontiguousarray(Ip, dtype=np.int64)
    D = np.empty((B, k), dtype=np.float32)
    I = np.empty((B, k), dtype=np.int64)
    _lib.check(_lib.load().anr_merge_topk_host(Dp.ctypes.data, Ip.ctypes.data, int(P),
                                               int(B), int(k), int(bool(larger_is_better)),
                                               D.ctypes.data, I.ctypes.data),
               "anr_merge_topk_host")
    return D, I


class ShardedFlatIndex:
    """``FlatIndex`` surface over P row shards driven by one process (one handle per entry of ``devices``; a device
    may be listed more than once).  Every ``add`` call is cut into P contiguous pieces, piece p appended to shard p,
    and ids stay the sequential ids of the add order (faiss ``IndexFlat.add``, reference vector_index.py:196)."""

    def __init__(self, d: int, metric: int = 0, normalize: bool = False, devices: Sequence[int] = (0,)):
        from .flat_index import FlatIndex
        if not devices:
            raise ValueError("ShardedFlatIndex needs at least one device")
        self.d, self.metric, self.normalize = int(d), int(metric), bool(normalize)
        self.devices = [int(v) for v in devices]
        self.device = self.devices[0]
        self.is_trained = True
        self.shards: List = []
        try:
            for dev in self.devices:
                self.shards.append(FlatIndex(d, metric, normalize, device=dev))
        except Exception:
            self.close()
            raise
        # per shard the segments (local_start, global_start, count) of its rows, in add order
        self._segs: List[List[Tuple[int, int, int]]] = [[] for _ in self.devices]
        self._ntotal = 0
        self._offset_mode = True  # every shard holds one segment: it returns global ids itself (ANR_OPT_ID_OFFSET)
        self._pool = ThreadPoolExecutor(max_workers=len(self.devices), thread_name_prefix="anr-shard")

    # -- bookkeeping ---------------------------------------------------------------------------------
    @property
    def ntotal(self) -> int:
        return self._ntotal

    @property
    def larger_is_better(self) -> bool:
        return self.metric == 0

    def _sync_offsets(self) -> None:
        from ._lib import OPT_ID_OFFSET
        single = all(len(s) <= 1 for s in self._segs)
        for sh, segs in zip(self.shards, self._segs):
            sh.set_option(OPT_ID_OFFSET, segs[0][1] if (single and segs) else 0)
        self._offset_mode = single

    def _to_global(self, p: int, I: np.ndarray) -> np.ndarray:
        segs = self._segs[p]
        if not segs:
            return I
        starts = np.array([s[0] for s in segs], dtype=np.int64)
        shift = np.array([s[1] - s[0] for s in segs], dtype=np.int64)
        j = np.clip(np.searchsorted(starts, I, side="right") - 1, 0, len(segs) - 1)
        return np.where(I >= 0, I + shift[j], -1)

    # -- FlatIndex surface ---------------------------------------------------------------------------
    def reserve(self, n: int) -> None:
        per = (int(n) + len(self.shards) - 1) // len(self.shards)
        for sh in self.shards:
            sh.reserve(max(per, sh.ntotal))

    def add(self, x) -> None:
        x = np.asarray(x)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"add: expected a 2-D array with {self.d} columns, got shape {x.shape}")
        n, P = x.shape[0], len(self.shards)
        if n == 0:
            return
        pieces = []
        for p in range(P):
            lo, hi = shard_bounds(n, P, p)
            if hi > lo:
                pieces.append((p, lo, hi))
        list(self._pool.map(lambda t: self.shards[t[0]].add(x[t[1]:t[2]]), pieces))
        for p, lo, hi in pieces:
            self._segs[p].append((self.shards[p].ntotal - (hi - lo), self._ntotal + lo, hi - lo))
        self._ntotal += n
        self._sync_offsets()

    def search(self, q, k: int):
        q = np.ascontiguousarray(np.asarray(q), dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"search: expected a 2-D array with {self.d} columns, got shape {q.shape}")
        k = int(k)
        live = [p for p, sh in enumerate(self.shards) if sh.ntotal > 0]
        if not live:
            return (np.full((q.shape[0], k), -FLT_MAX if self.larger_is_better else FLT_MAX, dtype=np.float32),
                    np.full((q.shape[0], k), -1, dtype=np.int64))
        parts = list(self._pool.map(lambda p: self.shards[p].search(q, k), live))
        if len(live) == 1 and self._offset_mode:
            return parts[0]
        Dp = np.stack([d for d, _ in parts])
        Ip = np.stack([i if self._offset_mode else self._to_global(p, i) for p, (_, i) in zip(live, parts)])
        return merge_topk_host_c(Dp, Ip, self.larger_is_better)

    def reconstruct_n(self, i0: int, n: int) -> np.ndarray:
        i0, n = int(i0), int(n)
        if i0 < 0 or n < 0 or i0 + n > self._ntotal:
            raise ValueError("row range out of bounds")
        out = np.empty((n, self.d), dtype=np.float32)
        for p, segs in enumerate(self._segs):
            for ls, gs, cnt in segs:
                lo, hi = max(gs, i0), min(gs + cnt, i0 + n)
                if hi > lo:
                    out[lo - i0:hi - i0] = self.shards[p].reconstruct_n(ls + (lo - gs), hi - lo)
        return out

    def score_rows(self, q, ids) -> np.ndarray:
        q = np.ascontiguousarray(np.asarray(q), dtype=np.float32)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.full(ids.shape, np.nan, dtype=np.float32)
        for p, segs in enumerate(self._segs):
            local = np.full(ids.shape, -1, dtype=np.int64)
            for ls, gs, cnt in segs:
                m = (ids >= gs) & (ids < gs + cnt)
                local[m] = ids[m] - gs + ls
            if (local >= 0).any():
                got = self.shards[p].score_rows(q, local)
                out[local >= 0] = got[local >= 0]
        return out

    def self_join(self, *a, **kw):
        raise NotImplementedError("self_join runs on a single-device FlatIndex (graph_scans builds its own)")

    def reset(self) -> None:
        for sh in self.shards:
            sh.reset()
        self._segs = [[] for _ in self.shards]
        self._ntotal = 0
        self._sync_offsets()

    def set_option(self, opt: int, value: int) -> None:
        from ._lib import OPT_ID_OFFSET
        if int(opt) == OPT_ID_OFFSET:
            raise ValueError("the id offsets of a sharded index are managed by the index")
        for sh in self.shards:
            sh.set_option(opt, value)

    def last_stats(self) -> dict:
        tot: dict = {}
        for sh in self.shards:
            for key, v in sh.last_stats().items():
                tot[key] = (max(tot.get(key, 0), v) if key in ("overfetch", "sample_rows", "scan_ms", "total_ms", "n_queries")
                            else tot.get(key, 0) + v)
        return tot

    def close(self) -> None:
        for sh in getattr(self, "shards", []):
            sh.close()
        self.shards = []
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=False)
            self._pool = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass


class ShardedSearcher:
    """One rank of a ``torch.distributed`` job (one process per GPU).  ``index`` is this rank's ``FlatIndex`` holding
    its row shard (first global row = ``row_offset``); alternatively a ``local_search(q, k) -> (D, I_local)``
    callable (numpy in / numpy out) for shard searches that do not come from the library (the CPU test uses the
    oracle).  ``search`` returns the merged global top-k on every rank."""

    def __init__(self, index_or_search, row_offset: int, larger_is_better: Optional[bool] = None, group=None,
                 force_device: bool = False):
        """force_device: take the RCCL device path even with a single rank (a world-size-1 rehearsal of the exchange)"""
        import torch.distributed as dist
        self.dist = dist
        self.row_offset = int(row_offset)
        self.group = group
        self.force_device = bool(force_device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.index = None
        self.local_search: Optional[Callable] = None
        if callable(index_or_search):
            self.local_search = index_or_search
            self.larger = True if larger_is_better is None else bool(larger_is_better)
        else:
            from ._lib import OPT_ID_OFFSET
            self.index = index_or_search
            self.index.set_option(OPT_ID_OFFSET, self.row_offset)  # the shard returns global ids itself
            self.larger = (self.index.metric == 0) if larger_is_better is None else bool(larger_is_better)
        self._bufs = {}

    # -- device path: RCCL ---------------------------------------------------------------------------
    def _search_device(self, q: np.ndarray, k: int):
        import torch
        from . import _lib
        idx = self.index
        dev = torch.device("cuda", idx.device)
        B = q.shape[0]
        nres = B * k
        id_off, part = packed_layout(nres)
        key = (B, k)
        if key not in self._bufs:
            self._bufs[key] = (torch.empty(part, device=dev, dtype=torch.uint8),
                               torch.empty(self.world * part, device=dev, dtype=torch.uint8),
                               torch.empty((B, k), device=dev, dtype=torch.float32),
                               torch.empty((B, k), device=dev, dtype=torch.int64))
        mine, allp, Dm, Im = self._bufs[key]
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev)
            qd = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(dev, non_blocking=False)
            idx.search_device_async(qd.data_ptr(), B, k, mine.data_ptr(), mine.data_ptr() + id_off, st.cuda_stream)
            idx.wait(0)  # final on this shard (certificate recovery done) BEFORE anything leaves it
            self.dist.all_gather_into_tensor(allp, mine, group=self.group)
            _lib.check(_lib.load().anr_merge_topk_strided_dev(
                idx.device, C.c_void_p(allp.data_ptr()), C.c_void_p(allp.data_ptr() + id_off), part // 4,
                part // 8, self.world, B, k, int(self.larger), C.c_void_p(Dm.data_ptr()),
                C.c_void_p(Im.data_ptr()), C.c_void_p(st.cuda_stream)), "anr_merge_topk_strided_dev")
            return Dm.cpu().numpy(), Im.cpu().numpy()

    # -- host path: gloo / single rank -----------------------------------------------------------------
    def _search_host(self, q: np.ndarray, k: int):
        import torch
        if self.index is not None:
            D, I = self.index.search(q, k)  # global ids (ANR_OPT_ID_OFFSET)
        else:
            D, I = self.local_search(q, k)
            I = np.where(I >= 0, I + self.row_offset, -1)
        D = np.ascontiguousarray(D, dtype=np.float32)
        I = np.ascontiguousarray(I, dtype=np.int64)
        if self.world == 1:
            return D, I
        B = q.shape[0]
        nres = B * k
        id_off, part = packed_layout(nres)
        packed = torch.zeros(part, dtype=torch.uint8)  # one exchange: [B*k f32 | pad to 8 | B*k i64]
        packed[: nres * 4] = torch.from_numpy(D.reshape(-1).view(np.uint8))
        packed[id_off:] = torch.from_numpy(I.reshape(-1).view(np.uint8))
        parts = [torch.empty_like(packed) for _ in range(self.world)]
        self.dist.all_gather(parts, packed, group=self.group)
        Dp = np.stack([p[: nres * 4].numpy().view(np.float32).reshape(B, k) for p in parts])
        Ip = np.stack([p[id_off:].numpy().view(np.int64).reshape(B, k) for p in parts])
        return merge_topk_host_c(Dp, Ip, self.larger)

    def stream(self, nq: int, k: int, lag: int = 2, group: int = 2, order_caller: bool = True) -> "ShardedStream":
        """pipelined searches of a stream of [nq, dim] batches, `group` batches per exchange (``ShardedStream``)"""
        return ShardedStream(self, nq, k, lag, group, order_caller)

    def search(self, q: np.ndarray, k: int):
        q = np.asarray(q)
        if self.index is not None and self.backend == "nccl" and (self.world > 1 or self.force_device):
            return self._search_device(q, int(k))
        return self._search_host(q, int(k))


class ShardedStream:
    """The pipelined form of ``ShardedSearcher.search`` for a STREAM of equal-shape query batches (bench.py, N > 1; a
    serving loop): ``submit`` launches this shard's search of one batch and returns at once; the partial lists of
    `group` consecutive batches travel in ONE all-gather, `lag` batches behind the search front and only once they are
    final on this shard; ``submit`` / ``flush`` hand back the merges that completed, as ``(tag, D, I)``.

    Device form (FlatIndex + nccl, or force_device): nothing leaves the GPU — D / I are torch CUDA tensors [nq, k], ordered
    on the caller's current stream (``order_caller``; else complete after ``wait()``) and valid until ``nslot`` further submits; queries are a CUDA tensor (or a numpy array, uploaded).  Host form (gloo, or a
    callable shard search): the shard search is synchronous, the grouped exchange and the merge run on the host, D / I
    are numpy arrays."""

    def __init__(self, searcher: "ShardedSearcher", nq: int, k: int, lag: int = 2, group: int = 2, order_caller: bool = True):
        self.s = searcher
        self.order_caller = bool(order_caller)  # False: read the merged lists after ``wait()`` instead (no marker in the caller's stream)
        self.last_event = None
        self.nq, self.k = int(nq), int(k)
        self.plan = ExchangePlan(lag, group)
        self.nres = self.nq * self.k
        self.id_off, self.part = packed_layout(self.nres)
        s = searcher
        self.device_form = s.index is not None and s.backend == "nccl" and (s.world > 1 or s.force_device)
        G, ns = self.plan.group, self.plan.nslot
        if self.device_form:
            import torch
            self.torch = torch
            dev = torch.device("cuda", s.index.device)
            self.dev = dev
            # ONE stream for every collective and merge.  (Rounds 2-3 gave every slot its own stream and passed it to the search
            # as "the stream the queries come from": the process's ~10 streams share four hardware queues, and a slot
            # stream's wait-for-the-last-collective marker then sat in front of another batch's kernels — with the exchange
            # on, the scans of the 8-GPU shard size ran ~100 us apart instead of back to back.)
            self.xstream = torch.cuda.Stream(device=dev)
            self.qstream = torch.cuda.Stream(device=dev)  # stands in for the null stream (see submit)
            # the slots' send buffers are contiguous, so a group of them is one send buffer
            self.send = torch.empty(ns * self.part, device=dev, dtype=torch.uint8)
            self.recv = [torch.empty(s.world * G * self.part, device=dev, dtype=torch.uint8) for _ in range(ns // G)]
            self.Dm = [torch.empty((self.nq, self.k), device=dev, dtype=torch.float32) for _ in range(ns)]
            self.Im = [torch.empty((self.nq, self.k), device=dev, dtype=torch.int64) for _ in range(ns)]
            self.slot_free = [None] * ns  # event behind the collective that last read a slot
            self._keep = [None] * ns      # uploaded queries stay alive until their slot comes round again
        else:
            self.send_h = np.zeros((ns, self.part), dtype=np.uint8)

    # -- one group: all-gather + merge ---------------------------------------------------------------------------
    def _exchange_device(self, grp):
        from . import _lib
        torch, s = self.torch, self.s
        n, s0 = len(grp), grp[0][1]
        st = self.xstream
        s.index.wait(len(self.plan.pending))  # everything older than the still-pending batches is final on this shard (host wait)
        recv = self.recv[s0 // self.plan.group][: s.world * n * self.part]
        out = []
        with torch.cuda.stream(st):
            s.dist.all_gather_into_tensor(recv, self.send[s0 * self.part:(s0 + n) * self.part], group=s.group)
            for t, (tag, slot) in enumerate(grp):  # rank r's list of batch t sits at r * n * part + t * part
                base = recv.data_ptr() + t * self.part
                _lib.check(_lib.load().anr_merge_topk_strided_dev(
                    s.index.device, C.c_void_p(base), C.c_void_p(base + self.id_off), n * self.part // 4,
                    n * self.part // 8, s.world, self.nq, self.k, int(s.larger), C.c_void_p(self.Dm[slot].data_ptr()),
                    C.c_void_p(self.Im[slot].data_ptr()), C.c_void_p(st.cuda_stream)), "anr_merge_topk_strided_dev")
                out.append((tag, self.Dm[slot], self.Im[slot]))
            ev = torch.cuda.Event()
            ev.record(st)
        for _, slot in grp:
            self.slot_free[slot] = ev
        self.last_event = ev
        if self.order_caller:
            # the caller's stream sees the merged lists complete (a device-side wait: nothing blocks on the host)
            torch.cuda.current_stream(self.dev).wait_event(ev)
        return out

    def _exchange_host(self, grp):
        import torch
        s = self.s
        n, s0 = len(grp), grp[0][1]
        mine = torch.from_numpy(self.send_h[s0:s0 + n].reshape(-1).copy())
        if s.world > 1:
            parts = [torch.empty_like(mine) for _ in range(s.world)]
            s.dist.all_gather(parts, mine, group=s.group)
        else:
            parts = [mine]
        out = []
        for t, (tag, _) in enumerate(grp):
            lo = t * self.part
            Dp = np.stack([p[lo:lo + self.nres * 4].numpy().view(np.float32).reshape(self.nq, self.k) for p in parts])
            Ip = np.stack([p[lo + self.id_off:lo + self.part].numpy().view(np.int64).reshape(self.nq, self.k) for p in parts])
            D, I = merge_topk_host_c(Dp, Ip, s.larger)
            out.append((tag, D, I))
        return out

    # -- the stream --------------------------------------------------------------------------------------------
    def submit(self, q, tag=None):
        """launch this shard's search of one [nq, dim] batch; -> the merges that completed, [(tag, D, I), ...]"""
        s = self.s
        if self.device_form:
            torch = self.torch
            slot, grp = self.plan.issue(tag)
            if self.slot_free[slot] is not None:
                # the collective that last read this slot's send buffer was launched a whole group of submits ago: a host
                # wait that is over before it starts, instead of a marker in a stream
                self.slot_free[slot].synchronize()
                self.slot_free[slot] = None
            if isinstance(q, np.ndarray):
                q = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(self.dev)
            # the stream the queries come from is the CALLER's current stream: the library orders the batch behind it when it
            # is busy (a query tensor whose producer is still queued there is safe to pass — ADVICE r3).  The C-ABI reads a
            # null stream handle as "the index's own stream", and torch's default stream IS handle 0: a busy default stream
            # is therefore handed over as a private stream whose only work is a wait for it.
            cur = torch.cuda.current_stream(self.dev)
            user = cur.cuda_stream
            if user == 0:
                if cur.query():
                    user = 0  # nothing queued before the queries: no ordering to carry
                else:
                    self.qstream.wait_stream(cur)
                    user = self.qstream.cuda_stream
            self._keep[slot] = q  # alive until the slot comes round again (its batch has been retired by then)
            base = self.send.data_ptr() + slot * self.part
            s.index.search_device_async(q.data_ptr(), self.nq, self.k, base, base + self.id_off, user)
            return self._exchange_device(grp) if grp is not None else []
        if hasattr(q, "detach"):
            q = q.detach().cpu().numpy()
        q = np.asarray(q)
        if s.index is not None:
            D, I = s.index.search(q, self.k)  # global ids (ANR_OPT_ID_OFFSET)
        else:
            D, I = s.local_search(q, self.k)
            I = np.where(I >= 0, I + s.row_offset, -1)
        slot, grp = self.plan.issue(tag)
        row = self.send_h[slot]
        row[: self.nres * 4] = np.ascontiguousarray(D, dtype=np.float32).reshape(-1).view(np.uint8)
        row[self.id_off:] = np.ascontiguousarray(I, dtype=np.int64).reshape(-1).view(np.uint8)
        return self._exchange_host(grp) if grp is not None else []

    def flush(self):
        """the merges still owed (device form: call after the index has retired every batch — ``FlatIndex.sync``)"""
        out = []
        for grp in self.plan.drain():
            out += self._exchange_device(grp) if self.device_form else self._exchange_host(grp)
        return out

    def wait(self) -> None:
        """host wait for every merge handed back so far (device form with ``order_caller=False``)"""
        if self.device_form and self.last_event is not None:
            self.last_event.synchronize()


def merge_topk_device(device: int, Dg, Ig, k: int, larger_is_better: bool, D_out, I_out, stream: int = 0) -> None:
    """Dg/Ig: torch CUDA tensors [P,B,k]; D_out/I_out [B,k] (same device)."""
    from . import _lib
    lib = _lib.load()
    P, B, kk = Dg.shape
    _lib.check(lib.anr_merge_topk_dev(int(device), C.c_void_p(Dg.data_ptr()), C.c_void_p(Ig.data_ptr()), int(P),
                                      int(B), int(k), int(bool(larger_is_better)), C.c_void_p(D_out.data_ptr()),
                                      C.c_void_p(I_out.data_ptr()), C.c_void_p(stream)), "anr_merge_topk_dev")
