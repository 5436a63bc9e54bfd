"""Host-side plumbing the drop-in classes share: the global ``config`` object, the ``logger`` and small
file helpers.  Inside the reference tree the reference's own ``config`` / loguru are used unchanged
(so every ``config.get('a.b.c', default)`` resolves exactly as there); stand-alone (this repository, the
GPU box) a minimal equivalent with the reference's defaults for the hot-path keys takes their place.
"""
from __future__ import annotations

import json
import logging
import os
from copy import deepcopy
from pathlib import Path
from typing import Any, Dict

import numpy as np

# defaults of the keys the hot path reads, as in the reference's config/config_loader.py DEFAULT_CONFIG
# (:10-15 embedding, :87-94 storage, :139-146 retrieval.hybrid, :365-372 vector_store)
_DEFAULTS: Dict[str, Any] = {
    "embedding": {"model_name": "BAAI/bge-m3", "batch_size": 64, "max_length": 512, "normalize": True},
    "vector_store": {"top_k": 20, "similarity_threshold": 0.5, "batch_size": 32, "dimension": 1024,
                     "index_type": "IVFFlat", "similarity_metric": "cosine"},
    "storage": {},
    "retrieval": {"candidate_pool": 50,
                  "hybrid": {"enabled": True, "fusion_method": "linear", "rrf_k": 60,
                             "weights": {"dense": 1.0, "bm25": 0.5, "graph": 0.5, "path": 0.1}}},
    "performance": {"use_gpu": True},
}


class _MiniConfig:
    """``get('a.b.c', default)`` / ``set`` / ``load_config`` with the reference ConfigLoader's semantics
    (config_loader.py:598-630) over an in-memory dict."""

    def __init__(self, data: Dict[str, Any] | None = None):
        self._config = deepcopy(_DEFAULTS)
        if data:
            self.update(data)

    def load_config(self) -> Dict[str, Any]:
        return self._config

    def get(self, key: str, default: Any = None) -> Any:
        value: Any = self._config
        for part in key.split("."):
            if isinstance(value, dict) and part in value:
                value = value[part]
            else:
                return default
        return value

    def set(self, key: str, value: Any) -> None:
        cur = self._config
        parts = key.split(".")
        for k in parts[:-1]:
            cur = cur.setdefault(k, {})
        cur[parts[-1]] = value

    def update(self, data: Dict[str, Any]) -> None:
        def rec(dst, src):
            for k, v in src.items():
                if isinstance(v, dict) and isinstance(dst.get(k), dict):
                    rec(dst[k], v)
                else:
                    dst[k] = v
        rec(self._config, data)

    def reset(self) -> None:
        self._config = deepcopy(_DEFAULTS)


def _find_config():
    try:  # inside the reference tree: its strict-schema loader
        from config import config as ref_config  # type: ignore
        if hasattr(ref_config, "get"):
            return ref_config
    except Exception:
        pass
    return _MiniConfig()


def _find_logger():
    try:
        from loguru import logger as lg  # type: ignore
        return lg
    except Exception:
        lg = logging.getLogger("anorag")
        if not lg.handlers:
            lg.addHandler(logging.NullHandler())
        return lg


config = _find_config()
logger = _find_logger()


def _to_plain(obj):
    if isinstance(obj, np.integer):
        return int(obj)
    if isinstance(obj, np.floating):
        return float(obj)
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, dict):
        return {(_to_plain(k) if isinstance(k, (np.integer, np.floating)) else k): _to_plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_to_plain(v) for v in obj]
    return obj


class FileUtils:
    """the three helpers of the reference's utils/file_utils.py the path uses (:53-87, :140-142)"""

    @staticmethod
    def ensure_dir(directory: str) -> None:
        Path(directory).mkdir(parents=True, exist_ok=True)

    @staticmethod
    def write_json(data, file_path: str) -> None:
        with open(file_path, "w", encoding="utf-8") as f:
            json.dump(_to_plain(data), f, ensure_ascii=False, indent=2)

    @staticmethod
    def read_json(file_path: str):
        with open(file_path, "r", encoding="utf-8") as f:
            return json.load(f)


def hip_available() -> bool:
    """what the reference asks torch (utils/gpu_utils.py:46-55): is there a device to run on"""
    try:
        from . import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


def default_tmp(name: str) -> str:
    import tempfile
    return os.path.join(tempfile.gettempdir(), name)
