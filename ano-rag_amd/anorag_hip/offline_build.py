"""Streamed / sharded index builds for multi-million-note corpora.

The reference builds its index in one shot — ``encode_atomic_notes(all notes)`` -> ``add_vectors`` (vector_store/
retriever.py:140-157; the offline driver vector_store/rebuild_vector_index.py:202-243 calls that or ``add_notes`` per
batch) — which holds every embedding on the host twice.  Here the notes are encoded chunk by chunk, each chunk's
embeddings stay in device memory and go straight into the index (``anr_encoder_forward_dev`` -> ``anr_index_add_dev``),
and the host copy (``embeddings.npy``, doc/document_processor.py:164-172) is written through a memory map when asked
for.  ``sharded_build`` is the one-process-per-GPU form: rank r encodes and indexes the contiguous note range
``shard_bounds(N, world, r)`` — no collective at build time — and hands back a ``ShardedSearcher``.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Sequence

import numpy as np

from .flat_index import METRIC_IP, FlatIndex
from .sharded import ShardedSearcher, shard_bounds


def stream_build(index: FlatIndex, embedding_manager, notes: Sequence[Dict[str, Any]], chunk_notes: int = 65536,
                 embeddings_npy: Optional[str] = None) -> int:
    """Encode ``notes`` in chunks and append them to ``index`` without a host round trip; returns the rows added.
    ``embeddings_npy``: also write the float32 [N, D] embeddings there (np.load-compatible, filled chunk by chunk)."""
    n = len(notes)
    if n == 0:
        return 0
    out = None
    index.reserve(index.ntotal + n)
    for lo in range(0, n, int(chunk_notes)):
        part = notes[lo:lo + int(chunk_notes)]
        dev = embedding_manager.encode_texts_device(embedding_manager._assemble_note_texts(part))
        try:
            if dev.n != index.d or dev.device != index.device:
                raise ValueError(f"encoder output [{dev.nq}, {dev.n}] on device {dev.device} does not fit the index "
                                 f"(dim {index.d}, device {index.device})")
            index.add_device(dev.ptr, dev.nq)
            if embeddings_npy:
                if out is None:
                    out = np.lib.format.open_memmap(embeddings_npy, mode="w+", dtype=np.float32, shape=(n, dev.n))
                out[lo:lo + dev.nq] = dev.numpy()
        finally:
            dev.free()
    if out is not None:
        out.flush()
        del out
    return n


def sharded_build(embedding_manager, notes: Sequence[Dict[str, Any]], world: int, rank: int, device: int = 0,
                  metric: int = METRIC_IP, normalize: bool = True, chunk_notes: int = 65536, group=None):
    """This rank's share of a row-sharded build: notes ``[lo, hi) = shard_bounds(N, world, rank)`` are encoded and
    indexed on ``device``.  Returns ``(searcher, index, (lo, hi))`` — ``searcher.search(q, k)`` gives the merged global
    top-k on every rank (RCCL all-gather of the partial lists when the process group uses the nccl backend)."""
    lo, hi = shard_bounds(len(notes), world, rank)
    index = FlatIndex(int(embedding_manager.embedding_dim), metric, normalize=normalize, device=device)
    stream_build(index, embedding_manager, notes[lo:hi], chunk_notes)
    return ShardedSearcher(index, lo, group=group), index, (lo, hi)
