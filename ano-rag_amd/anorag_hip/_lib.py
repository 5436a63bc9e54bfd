"""ctypes loader for libanorag_hip.so (the C ABI declared in include/anorag.h).

There is no CPU fallback: if the shared library is missing or a HIP call fails the caller gets an
exception (``AnoragError``), never a silently different code path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libanorag_hip.so"
# ANORAG_LIB: load another build of the same C ABI (a deployment's own path, or a developer's A/B variant)
LIB_PATH = os.environ.get("ANORAG_LIB") or os.path.join(_HERE, LIB_NAME)

ANR_OK = 0
METRIC_IP = 0
METRIC_L2 = 1

OPT_FORCE_EXACT = 1
OPT_OVERFETCH = 2
OPT_SAMPLE_ROWS = 3
OPT_CAND_CAP = 4
OPT_TIMING = 5
OPT_ADD_RAW = 6
OPT_STREAMS = 7
OPT_ID_OFFSET = 8
OPT_TINY = 9
OPT_FUSED_POST = 10
OPT_SHADOW = 11
OPT_SCHEDULE = 12
OPT_STREAM_WAIT = 13
OPT_SCAN_BITS = 14
BATCH_LOG_FIELDS = 14


class AnoragError(RuntimeError):
    """A call into libanorag_hip.so failed (message from anr_last_error())."""


class SearchStats(C.Structure):
    _fields_ = [
        ("n_queries", C.c_int64),
        ("n_fallback", C.c_int64),
        ("n_from_lists", C.c_int64),
        ("n_dense_exact", C.c_int64),
        ("n_candidates", C.c_int64),
        ("n_overflow", C.c_int64),
        ("scan_bytes", C.c_int64),
        ("overfetch", C.c_int32),
        ("sample_rows", C.c_int32),
        ("scan_ms", C.c_float),
        ("total_ms", C.c_float),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class FuseSource(C.Structure):
    _fields_ = [("array_dev", C.c_void_p), ("array_len", C.c_int64), ("array_dtype", C.c_int32),
                ("list_ids", C.c_void_p), ("list_scores", C.c_void_p), ("list_offs", C.c_void_p),
                ("array_max_dev", C.c_void_p), ("sparse_ids_dev", C.c_void_p), ("sparse_scores_dev", C.c_void_p),
                ("sparse_count_dev", C.c_void_p), ("sparse_cap", C.c_int64)]


class FuseDenseStats(C.Structure):
    _fields_ = [("n_queries", C.c_int64), ("scan_bytes", C.c_int64), ("n_candidates", C.c_int64),
                ("scan_ms", C.c_float)]


class EncoderConfig(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32), ("hidden", C.c_int32), ("n_heads", C.c_int32), ("intermediate", C.c_int32),
        ("vocab_size", C.c_int32), ("max_positions", C.c_int32), ("type_vocab_size", C.c_int32),
        ("pos_offset", C.c_int32), ("pooling", C.c_int32), ("act", C.c_int32), ("ln_eps", C.c_float),
    ]


_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol include/anorag.h declares must appear here
SIGNATURES = {
    "anr_last_error": (C.c_char_p, []),
    "anr_version": (C.c_char_p, []),
    "anr_device_count": (C.c_int, []),
    "anr_device_malloc": (C.c_int, [C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]),
    "anr_device_free": (C.c_int, [C.c_int32, C.c_void_p]),
    "anr_device_copy": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "anr_index_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "anr_index_destroy": (C.c_int, [C.c_void_p]),
    "anr_index_reserve": (C.c_int, [C.c_void_p, C.c_int64]),
    "anr_index_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "anr_index_add_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "anr_index_ntotal": (C.c_int64, [C.c_void_p]),
    "anr_index_dim": (C.c_int32, [C.c_void_p]),
    "anr_index_reset": (C.c_int, [C.c_void_p]),
    "anr_index_reconstruct": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "anr_index_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "anr_index_search_devq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "anr_index_search_dev": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "anr_index_search_dev_async": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "anr_index_sync": (C.c_int, [C.c_void_p]),
    "anr_index_wait": (C.c_int, [C.c_void_p, C.c_int32]),
    "anr_index_reset_stats": (C.c_int, [C.c_void_p]),
    "anr_index_score_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "anr_similarity_matrix": (C.c_int, [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_void_p]),
    "anr_index_self_join": (
        C.c_int,
        [C.c_void_p, C.c_float, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)],
    ),
    "anr_index_set_option": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    "anr_index_last_stats": (C.c_int, [C.c_void_p, C.POINTER(SearchStats)]),
    "anr_index_reconstruct_scan_image": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    "anr_index_scan_image_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                             C.POINTER(C.c_float)]),
    "anr_index_batch_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "anr_normalize_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32]),
    "anr_merge_topk_dev": (
        C.c_int,
        [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
         C.c_void_p],
    ),
    "anr_merge_topk_strided_dev": (
        C.c_int,
        [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
         C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "anr_merge_topk_host": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p],
    ),
    "anr_fuse_lists": (
        C.c_int,
        [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32,
         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "anr_fuse_dense": (
        C.c_int,
        [C.c_int32, C.c_int32, C.c_int64, C.POINTER(FuseSource), C.c_void_p, C.c_double, C.c_int32, C.c_void_p,
         C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FuseDenseStats)],
    ),
    "anr_fuse_candidates": (
        C.c_int,
        [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p],
    ),
    "anr_fuse_rrf_long": (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "anr_bm25_scores_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                      C.c_void_p]),
    "anr_bm25_combine_fields": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                          C.c_void_p, C.c_void_p]),
    "anr_bm25_sparse_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "anr_bm25_create": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_void_p)]),
    "anr_bm25_destroy": (C.c_int, [C.c_void_p]),
    "anr_bm25_scores": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "anr_bm25_nonzero": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "anr_encoder_create": (C.c_int, [C.POINTER(EncoderConfig), C.c_int32, C.POINTER(C.c_void_p)]),
    "anr_encoder_destroy": (C.c_int, [C.c_void_p]),
    "anr_encoder_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "anr_encoder_finalize": (C.c_int, [C.c_void_p]),
    "anr_encoder_forward": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    ),
    "anr_encoder_forward_dev": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p],
    ),
    "anr_encoder_forward_shared": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    ),
    "anr_encoder_shared_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


def load():
    """Load the shared library once and bind the argument types."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise AnoragError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C ano-rag_amd/csrc`; there is no CPU fallback"
            )
        # PyTorch ships its own libamdhip64.so.7; when both live in one process the HIP runtime must be
        # the same object, so let torch (if it is going to be used at all) load its copy first.
        if os.environ.get("ANORAG_NO_TORCH_PRELOAD", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:  # pragma: no cover - torch absent is fine
                pass
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error() -> str:
    msg = load().anr_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str = "") -> None:
    if rc != ANR_OK:
        raise AnoragError(f"{what or 'libanorag_hip'} failed (code {rc}): {last_error()}")


def device_count() -> int:
    return int(load().anr_device_count())
