"""Thin object wrapper over the anr_index_* C ABI.

``FlatIndex`` is what the drop-in ``vector_store.VectorIndex`` puts where the reference holds a
``faiss.IndexFlatIP`` / ``IndexFlatL2`` object (reference vector_store/vector_index.py:77-80): the same
``add`` / ``search`` / ``reset`` / ``ntotal`` surface, exact results, device-resident corpus.
"""
from __future__ import annotations

import ctypes as C
import gc

import numpy as np

from . import _lib
from ._lib import METRIC_IP, METRIC_L2, AnoragError, SearchStats


def _f32_matrix(a, d, what):
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[1] != d:
        raise ValueError(f"{what}: expected a 2-D array with {d} columns, got shape {a.shape}")
    return np.ascontiguousarray(a, dtype=np.float32)


_heap_settled = False


def _settle_interpreter_heap() -> None:
    """Once per process, when the first index is created: one full collection of the interpreter's heap.  A per-query
    search is ~30 us, and the collector's first generation-2 pass over everything the imports left behind (numpy,
    tokenizers, ...) is ~38 ms in the middle of one of them — the 188th and 317th call of a fresh process in
    tools/first_calls.py, none with the collector off.  After an explicit full collection CPython skips further full
    passes until a quarter more long-lived objects exist."""
    global _heap_settled
    if not _heap_settled:
        _heap_settled = True
        gc.collect()


class FlatIndex:
    """Exact inner-product / squared-L2 index resident on one MI355X."""

    def __init__(self, d: int, metric: int = METRIC_IP, normalize: bool = False, device: int = 0):
        self._lib = _lib.load()
        _settle_interpreter_heap()
        self.d = int(d)
        self.metric = int(metric)
        self.normalize = bool(normalize)
        self.device = int(device)
        self.is_trained = True
        h = C.c_void_p()
        _lib.check(self._lib.anr_index_create(self.d, self.metric, int(self.normalize), self.device, C.byref(h)),
                   "anr_index_create")
        self._h = h

    # -- faiss-like surface -------------------------------------------------------------------
    @property
    def ntotal(self) -> int:
        return int(self._lib.anr_index_ntotal(self._h)) if self._h else 0

    def add(self, x) -> None:
        x = _f32_matrix(x, self.d, "add")
        _lib.check(self._lib.anr_index_add(self._h, x.ctypes.data, x.shape[0]), "anr_index_add")

    def add_npy(self, path: str, chunk_rows: int = 1 << 18) -> int:
        """Stream an [N, d] ``.npy`` file (the reference's ``embeddings.npy``, doc/document_processor.py:164-172, or an
        array saved from ``note_embeddings.npz``) into the index without loading it whole: the file is memory-mapped
        and added chunk by chunk (any float dtype; rows are converted to float32 per chunk).  Returns the row count."""
        arr = np.load(path, mmap_mode="r")
        if arr.ndim != 2 or arr.shape[1] != self.d:
            raise ValueError(f"{path}: expected an [N, {self.d}] array, found shape {arr.shape}")
        if arr.dtype.kind != "f":
            raise ValueError(f"{path}: expected a floating-point array, found {arr.dtype}")
        self.reserve(self.ntotal + arr.shape[0])
        for i in range(0, arr.shape[0], int(chunk_rows)):
            self.add(np.ascontiguousarray(arr[i:i + int(chunk_rows)], dtype=np.float32))
        return int(arr.shape[0])

    def search(self, q, k: int):
        q = _f32_matrix(q, self.d, "search")
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _lib.check(
            self._lib.anr_index_search(self._h, q.ctypes.data, nq, int(k),
                                       D.ctypes.data, I.ctypes.data),
            "anr_index_search",
        )
        return D, I

    def score_rows(self, q, ids):
        """exact score of query i against each stored row ids[i][j] (float32 [nq, per]; NaN for bad ids)"""
        q = _f32_matrix(q, self.d, "score_rows")
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        if ids.ndim != 2 or ids.shape[0] != q.shape[0]:
            raise ValueError("ids must be [nq, per_query]")
        out = np.empty(ids.shape, dtype=np.float32)
        _lib.check(self._lib.anr_index_score_rows(self._h, q.ctypes.data, q.shape[0],
                                                  ids.ctypes.data, ids.shape[1],
                                                  out.ctypes.data), "anr_index_score_rows")
        return out

    def self_join(self, threshold: float, cap_hint: int = 0, sort: bool = True):
        """all pairs i < j of stored rows with inner product >= threshold: (i, j, score) arrays, sorted by (i, j)
        unless sort=False (device order)"""
        cap = int(cap_hint) if cap_hint > 0 else max(4 * self.ntotal, 1 << 16)
        n = C.c_int64(0)
        for _ in range(3):
            I = np.empty((cap,), dtype=np.int64)
            J = np.empty((cap,), dtype=np.int64)
            S = np.empty((cap,), dtype=np.float32)
            _lib.check(self._lib.anr_index_self_join(self._h, C.c_float(float(threshold)), cap,
                                                     I.ctypes.data, J.ctypes.data,
                                                     S.ctypes.data, C.byref(n)), "anr_index_self_join")
            if n.value >= 0:
                break
            cap = -n.value  # the lists were too small: the library reports an upper bound
        else:
            raise _lib.AnoragError("anr_index_self_join: pair lists kept overflowing")
        m = n.value
        if not sort:
            return I[:m], J[:m], S[:m]
        order = np.argsort(I[:m] * np.int64(self.ntotal) + J[:m], kind="stable")
        return I[:m][order], J[:m][order], S[:m][order]

    def reset(self) -> None:
        _lib.check(self._lib.anr_index_reset(self._h), "anr_index_reset")

    def reconstruct_n(self, i0: int, n: int):
        out = np.empty((n, self.d), dtype=np.float32)
        _lib.check(self._lib.anr_index_reconstruct(self._h, int(i0), int(n), out.ctypes.data),
                   "anr_index_reconstruct")
        return out

    def reconstruct_scan_image(self, bits: int, i0: int, n: int):
        """rows [i0, i0 + n) of the image the streaming scan reads (16: the f16 image, 12: the 12-bit one), as float32"""
        out = np.empty((n, self.d), dtype=np.float32)
        _lib.check(self._lib.anr_index_reconstruct_scan_image(self._h, int(bits), int(i0), int(n), out.ctypes.data),
                   "anr_index_reconstruct_scan_image")
        return out

    def scan_image_stats(self) -> dict:
        """{'bits': image batches scan now, 'max_norm', 'max_err16', 'max_err12'}: the statistics the certificate uses"""
        b, a, e16, e12 = C.c_int32(0), C.c_float(0), C.c_float(0), C.c_float(0)
        _lib.check(self._lib.anr_index_scan_image_stats(self._h, C.byref(b), C.byref(a), C.byref(e16), C.byref(e12)),
                   "anr_index_scan_image_stats")
        return {"bits": int(b.value), "max_norm": float(a.value), "max_err16": float(e16.value), "max_err12": float(e12.value)}

    # -- device-pointer surface (torch tensors are only carriers of device memory) --------------
    def reserve(self, n: int) -> None:
        _lib.check(self._lib.anr_index_reserve(self._h, int(n)), "anr_index_reserve")

    def add_device(self, x_ptr: int, n: int, stream: int = 0) -> None:
        _lib.check(self._lib.anr_index_add_dev(self._h, C.c_void_p(x_ptr), int(n), C.c_void_p(stream)),
                   "anr_index_add_dev")

    def search_device_queries(self, q_ptr: int, nq: int, k: int):
        """queries in device memory (float32 [nq, d] at q_ptr, e.g. ``SentenceEncoder.encode_device``), results as
        numpy arrays"""
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _lib.check(self._lib.anr_index_search_devq(self._h, C.c_void_p(q_ptr), int(nq), int(k),
                                                   D.ctypes.data, I.ctypes.data),
                   "anr_index_search_devq")
        return D, I

    def search_device(self, q_ptr: int, nq: int, k: int, d_ptr: int, i_ptr: int, stream: int = 0) -> None:
        _lib.check(
            self._lib.anr_index_search_dev(self._h, C.c_void_p(q_ptr), int(nq), int(k), C.c_void_p(d_ptr),
                                           C.c_void_p(i_ptr), C.c_void_p(stream)),
            "anr_index_search_dev",
        )

    def search_device_async(self, q_ptr: int, nq: int, k: int, d_ptr: int, i_ptr: int, stream: int = 0) -> None:
        """enqueue only; results are final after ``sync()`` (see include/anorag.h)"""
        _lib.check(
            self._lib.anr_index_search_dev_async(self._h, C.c_void_p(q_ptr), int(nq), int(k), C.c_void_p(d_ptr),
                                                 C.c_void_p(i_ptr), C.c_void_p(stream)),
            "anr_index_search_dev_async",
        )

    def sync(self) -> None:
        _lib.check(self._lib.anr_index_sync(self._h), "anr_index_sync")

    def wait(self, keep: int = 0) -> None:
        """retire the oldest in-flight batches until at most ``keep`` remain: their results are final"""
        _lib.check(self._lib.anr_index_wait(self._h, int(keep)), "anr_index_wait")

    def reset_stats(self) -> None:
        _lib.check(self._lib.anr_index_reset_stats(self._h), "anr_index_reset_stats")

    def set_option(self, opt: int, value: int) -> None:
        _lib.check(self._lib.anr_index_set_option(self._h, int(opt), int(value)), "anr_index_set_option")

    def last_stats(self) -> dict:
        st = SearchStats()
        _lib.check(self._lib.anr_index_last_stats(self._h, C.byref(st)), "anr_index_last_stats")
        return st.as_dict()

    BATCH_LOG_NAMES = ("seq", "host_enqueue_ns", "host_enqueued_ns", "host_retired_ns", "dev_prep_end_ns",
                       "dev_sample_end_ns", "dev_ladder_end_ns", "dev_scan_first_wg_ns", "dev_scan_last_wg_ns",
                       "dev_scan_end_ns", "dev_select_end_ns", "dev_post_end_ns", "flags", "host_recovery_ns")

    def batch_log(self, max_batches: int = 512, correlate: bool = True):
        """time line of the most recent pipeline batches (anr_index_batch_log): (records [n, 14] int64, device clock
        minus host clock in ns or None); retires the batches in flight first"""
        out = np.zeros((max(1, int(max_batches)), _lib.BATCH_LOG_FIELDS), dtype=np.int64)
        n = C.c_int32(0)
        off = C.c_int64(0)
        _lib.check(self._lib.anr_index_batch_log(self._h, out.ctypes.data, int(max_batches), C.byref(n),
                                                 C.byref(off) if correlate else None), "anr_index_batch_log")
        return out[: n.value], (int(off.value) if correlate else None)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.anr_index_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass


__all__ = ["FlatIndex", "METRIC_IP", "METRIC_L2", "AnoragError"]
