"""Host binding of the N-array score fusion (``anr_fuse_dense``, include/anorag.h): HybridSearcher.fuse of the
reference (retrieval/hybrid_search.py:34-103) when a source is a full-corpus score vector — what
``bm25_scores`` returns (utils/bm25_search.py:286-340) — rather than a short (id, score) list.

``DeviceArray`` is a plain device buffer obtained through the C ABI (no torch needed); ``fuse_dense`` takes, per
source, a ``DeviceArray`` / a device pointer triple, per-query short lists, or ``None``.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import FuseDenseStats, FuseSource

SOURCES = ("dense", "bm25", "graph", "path")


class _DevicePool:
    """Device buffers of the per-call objects below are kept and handed out again.  ``hipFree`` synchronises the device
    and costs ~0.5-1 ms apiece: the four buffers of a ``SparseRows`` made and freed around every ``fuse_bm25`` call were
    4 ms of a 9-ms call (cProfile, tools/bm25_fuse_perf.py, round 4).  Exact-size reuse; at most ``keep_bytes`` idle per
    device, the oldest idle buffers go first."""

    def __init__(self, keep_bytes: int = 6 << 30):
        import threading
        self.lock = threading.Lock()
        self.idle: Dict[Tuple[int, int], List[int]] = {}
        self.order: List[Tuple[int, int, int]] = []  # (device, nbytes, ptr), oldest first
        self.idle_bytes: Dict[int, int] = {}
        self.keep_bytes = int(keep_bytes)

    def take(self, device: int, nbytes: int) -> int:
        nbytes = max(int(nbytes), 8)
        with self.lock:
            lst = self.idle.get((device, nbytes))
            if lst:
                ptr = lst.pop()
                self.order.remove((device, nbytes, ptr))
                self.idle_bytes[device] -= nbytes
                return ptr
        p = C.c_void_p()
        _lib.check(_lib.load().anr_device_malloc(int(device), nbytes, C.byref(p)), "anr_device_malloc")
        return p.value

    def give(self, device: int, nbytes: int, ptr: int) -> None:
        nbytes = max(int(nbytes), 8)
        drop = []
        with self.lock:
            self.idle.setdefault((device, nbytes), []).append(ptr)
            self.order.append((device, nbytes, ptr))
            self.idle_bytes[device] = self.idle_bytes.get(device, 0) + nbytes
            while self.idle_bytes[device] > self.keep_bytes:
                i = next((k for k, o in enumerate(self.order) if o[0] == device), None)
                if i is None:
                    break
                d, nb, pt = self.order.pop(i)
                self.idle[(d, nb)].remove(pt)
                self.idle_bytes[d] -= nb
                drop.append((d, pt))
        for d, pt in drop:
            _lib.load().anr_device_free(d, C.c_void_p(pt))

    def clear(self) -> None:
        with self.lock:
            order, self.order, self.idle, self.idle_bytes = self.order, [], {}, {}
        for d, _, pt in order:
            _lib.load().anr_device_free(d, C.c_void_p(pt))


_pool = _DevicePool()


class DeviceArray:
    """[nq, n] float64 / float32 array in device memory (row-major).  ``row_max`` (optional): a [nq, 1] float64
    ``DeviceArray`` holding each row's maximum, set by the producers that know it (``DeviceBM25.scores_device``);
    ``fuse_dense`` hands it on and the linear fusion skips its own max pass over the array."""

    def __init__(self, nq: int, n: int, dtype=np.float64, device: int = 0):
        self.row_max = None
        self.nq, self.n, self.device = int(nq), int(n), int(device)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("DeviceArray holds float64 or float32")
        self.nbytes = self.nq * self.n * self.dtype.itemsize
        self.ptr = _pool.take(self.device, self.nbytes)

    @classmethod
    def wrap(cls, ptr: int, nq: int, n: int, dtype=np.float64, device: int = 0, row_max: "DeviceArray" = None) -> "DeviceArray":
        """a view of device memory owned by someone else (a torch tensor, another library's buffer): never freed here"""
        self = object.__new__(cls)
        self.row_max = row_max
        self.nq, self.n, self.device = int(nq), int(n), int(device)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("DeviceArray holds float64 or float32")
        self.nbytes = self.nq * self.n * self.dtype.itemsize
        self.ptr = int(ptr)
        self._borrowed = True
        return self

    @classmethod
    def from_numpy(cls, a: np.ndarray, device: int = 0, with_max: bool = False) -> "DeviceArray":
        """with_max: also upload each row's maximum over its non-NaN entries (``row_max``) — the host has the rows in
        hand anyway, and the linear fusion then skips its max pass"""
        a = np.ascontiguousarray(a)
        if a.ndim != 2:
            raise ValueError("expected [nq, n]")
        if a.dtype not in (np.float64, np.float32):
            a = a.astype(np.float64)
        out = cls(a.shape[0], a.shape[1], a.dtype, device)
        _lib.check(_lib.load().anr_device_copy(device, C.c_void_p(out.ptr), a.ctypes.data, a.nbytes, 0),
                   "anr_device_copy")
        if with_max and a.shape[1] > 0:
            m = np.fmax.reduce(a.astype(np.float64, copy=False), axis=1).reshape(-1, 1)  # fmax skips NaN; all-NaN row -> NaN
            out.row_max = cls.from_numpy(m, device)
        return out

    def numpy(self) -> np.ndarray:
        out = np.empty((self.nq, self.n), dtype=self.dtype)
        _lib.check(_lib.load().anr_device_copy(self.device, out.ctypes.data, C.c_void_p(self.ptr),
                                               self.nbytes, 1), "anr_device_copy")
        return out

    def free(self) -> None:
        if getattr(self, "row_max", None) is not None:
            self.row_max.free()
            self.row_max = None
        if getattr(self, "ptr", None):
            if not getattr(self, "_borrowed", False):
                _pool.give(self.device, self.nbytes, self.ptr)  # (kept for the next array of this size: see _DevicePool)
            self.ptr = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.free()
        except Exception:
            pass


class SparseRows:
    """[nq] rows over ``n`` ids in SPARSE form, in device memory: per row up to ``cap`` explicit (id, value) entries
    (uint32 / float64, ids distinct; any order while ``cap`` <= 8192, ascending ids beyond, up to 65536), every other id
    holds 0.0 — ``anr_fuse_source.sparse_*``.  What
    ``DeviceBM25.scores_sparse_device`` returns; ``fuse_dense`` takes it wherever it takes a ``DeviceArray`` and gives
    the same results without streaming n scores per query."""

    def __init__(self, nq: int, n: int, cap: int, device: int = 0):
        self.nq, self.n, self.cap, self.device = int(nq), int(n), int(cap), int(device)
        self._bufs = []
        self._sizes = (self.nq * self.cap * 4, self.nq * self.cap * 8, self.nq * 4, self.nq * 8)
        for nbytes in self._sizes:
            self._bufs.append(_pool.take(self.device, nbytes))
        self.ids_ptr, self.scores_ptr, self.count_ptr, self.max_ptr = self._bufs
        self.counts: Optional[np.ndarray] = None  # host copy of the row lengths, when the producer returned them

    @classmethod
    def wrap(cls, ids_ptr: int, scores_ptr: int, count_ptr: int, nq: int, n: int, cap: int, device: int = 0) -> "SparseRows":
        """a view of device buffers owned by someone else (uint32 [nq, cap], float64 [nq, cap], int32 [nq]): never freed here"""
        self = object.__new__(cls)
        self.nq, self.n, self.cap, self.device = int(nq), int(n), int(cap), int(device)
        self._bufs = []
        self.ids_ptr, self.scores_ptr, self.count_ptr, self.max_ptr = int(ids_ptr), int(scores_ptr), int(count_ptr), None
        self.counts = None
        return self

    @classmethod
    def from_numpy(cls, rows: Sequence[Tuple[np.ndarray, np.ndarray]], n: int, cap: Optional[int] = None,
                   device: int = 0) -> "SparseRows":
        """rows[i] = (ids, values) of row i (rows of more than 8192 entries are put in id order here: the fusion sorts
        shorter rows itself and asks that of its caller beyond — ``anr_fuse_source.sparse_cap``)"""
        nq = len(rows)
        cap = int(cap or max([len(a) for a, _ in rows] + [1]))
        if cap > 8192:
            srt = []
            for a, b in rows:
                order = np.argsort(np.asarray(a, dtype=np.uint32), kind="stable")
                srt.append((np.asarray(a)[order], np.asarray(b)[order]))
            rows = srt
        ids = np.zeros((nq, cap), dtype=np.uint32)
        val = np.zeros((nq, cap), dtype=np.float64)
        cnt = np.zeros((nq,), dtype=np.int32)
        for i, (a, b) in enumerate(rows):
            if len(a) > cap:
                raise ValueError(f"row {i} holds {len(a)} entries, cap is {cap}")
            ids[i, :len(a)] = np.asarray(a, dtype=np.uint32)
            val[i, :len(a)] = np.asarray(b, dtype=np.float64)
            cnt[i] = len(a)
        out = cls(nq, n, cap, device)
        lib = _lib.load()
        for ptr, arr in ((out.ids_ptr, ids), (out.scores_ptr, val), (out.count_ptr, cnt)):
            if arr.nbytes:
                _lib.check(lib.anr_device_copy(device, C.c_void_p(ptr), arr.ctypes.data, arr.nbytes, 0),
                           "anr_device_copy")
        out.counts = cnt
        return out

    def numpy(self) -> List[Tuple[np.ndarray, np.ndarray]]:
        """the rows back on the host, each sorted by id"""
        lib = _lib.load()
        ids = np.empty((self.nq, self.cap), dtype=np.uint32)
        val = np.empty((self.nq, self.cap), dtype=np.float64)
        cnt = np.empty((self.nq,), dtype=np.int32)
        for ptr, arr in ((self.ids_ptr, ids), (self.scores_ptr, val), (self.count_ptr, cnt)):
            if arr.nbytes:
                _lib.check(lib.anr_device_copy(self.device, arr.ctypes.data, C.c_void_p(ptr), arr.nbytes, 1),
                           "anr_device_copy")
        out = []
        for i in range(self.nq):
            k = max(int(cnt[i]), 0)
            order = np.argsort(ids[i, :k], kind="stable")
            out.append((ids[i, :k][order].astype(np.int64), val[i, :k][order]))
        return out

    def free(self) -> None:
        for b, nbytes in zip(getattr(self, "_bufs", []), getattr(self, "_sizes", ())):
            if b:
                _pool.give(self.device, nbytes, b)
        self._bufs = []

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.free()
        except Exception:
            pass


def fuse_dense(method: str, weights: Dict[str, float], rrf_k: float, pool: int, nq: int,
               sources: Dict[str, Any], device: int = 0, want_stats: bool = False):
    """sources[name] is a ``DeviceArray`` ([nq, N]: every id < N present, NaN = absent), a ``SparseRows`` (the same
    rows given by their non-zero entries; one source at most, the others lists), or a sequence of nq
    ``(ids int64[], scores float64[])`` short lists (the caller's order), or None / missing.
    Returns (ids [nq, pool] int64 (-1 padded), finals [nq, pool] float64, src [nq, pool, 4] float64 (NaN = absent),
    counts [nq] int32[, stats dict])."""
    lib = _lib.load()
    src = (FuseSource * 4)()
    keep = []  # keeps the numpy buffers alive across the call
    for si, name in enumerate(SOURCES):
        v = sources.get(name)
        if v is None:
            continue
        if isinstance(v, SparseRows):
            if v.nq != nq:
                raise ValueError(f"{name}: {v.nq} sparse rows, expected {nq}")
            if v.counts is not None and (np.asarray(v.counts) < 0).any():
                # count -1 = the producer's overflow mark (scores_sparse_device(allow_overflow=True)): such a query has to
                # be fused from its N-vector (HybridSearcher.fuse_bm25 splits the batch); fusing it here would read it as
                # "every BM25 score is 0.0".  Rows whose counts stayed on the device are checked there (ANR_EINVAL).
                bad = np.flatnonzero(np.asarray(v.counts) < 0)
                raise ValueError(f"{name}: sparse rows {bad[:8].tolist()} overflowed in the producer (count -1); "
                                 "route those queries through the N-vector form")
            src[si].array_len = v.n
            src[si].sparse_ids_dev = v.ids_ptr
            src[si].sparse_scores_dev = v.scores_ptr
            src[si].sparse_count_dev = v.count_ptr
            src[si].sparse_cap = v.cap
            continue
        if isinstance(v, DeviceArray):
            if v.nq != nq:
                raise ValueError(f"{name}: array has {v.nq} rows, expected {nq}")
            src[si].array_dev = v.ptr
            src[si].array_len = v.n
            src[si].array_dtype = 0 if v.dtype == np.float64 else 1
            if v.row_max is not None and v.row_max.ptr:
                src[si].array_max_dev = v.row_max.ptr
            continue
        if len(v) != nq:
            raise ValueError(f"{name}: {len(v)} lists for {nq} queries")
        offs = np.zeros(nq + 1, dtype=np.int64)
        for i, (ids, _) in enumerate(v):
            offs[i + 1] = offs[i] + len(ids)
        ids = (np.concatenate([np.asarray(a, dtype=np.int64).reshape(-1) for a, _ in v])
               if offs[-1] else np.zeros(0, dtype=np.int64))
        sc = (np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1) for _, b in v])
              if offs[-1] else np.zeros(0, dtype=np.float64))
        keep += [offs, ids, sc]
        src[si].list_offs = offs.ctypes.data
        src[si].list_ids = ids.ctypes.data
        src[si].list_scores = sc.ctypes.data
    w = np.asarray([float(weights.get(k, 0.0)) for k in SOURCES], dtype=np.float64)
    o_ids = np.empty((nq, pool), dtype=np.int64)
    o_fin = np.empty((nq, pool), dtype=np.float64)
    o_src = np.empty((nq, pool, 4), dtype=np.float64)
    o_cnt = np.empty((nq,), dtype=np.int32)
    st = FuseDenseStats()
    _lib.check(lib.anr_fuse_dense(int(device), 1 if method == "rrf" else 0, int(nq), src, w.ctypes.data,
                                  float(rrf_k), int(pool), o_ids.ctypes.data,
                                  o_fin.ctypes.data, o_src.ctypes.data,
                                  o_cnt.ctypes.data, C.byref(st) if want_stats else None),
               "anr_fuse_dense")
    if want_stats:
        return o_ids, o_fin, o_src, o_cnt, {k: getattr(st, k) for k, _ in st._fields_}
    return o_ids, o_fin, o_src, o_cnt
