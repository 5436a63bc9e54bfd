"""The combining queue in front of the encoder forward (anorag_hip.encoder._ForwardCombiner) with a stand-in device
encoder: concurrent single-query calls must each get exactly their own rows, share forwards, never deadlock when the
lead is handed on, and receive the error of a failed forward — the host logic of the reference's query-time pattern
(worker threads sharing one model, main_musique.py:487-494, query/query_processor.py:2761-2766)."""
import threading
import time

import numpy as np
import pytest

from anorag_hip.encoder import _ForwardCombiner


class _FakeEncoder:
    def __init__(self, delay=0.002, fail_on=None):
        self.delay, self.fail_on, self.calls, self.active, self.max_active = delay, fail_on, [], 0, 0
        self.lock = threading.Lock()

    def forward(self, ids, lengths, type_ids=None, normalize=False):
        with self.lock:
            self.active += 1
            self.max_active = max(self.max_active, self.active)
            self.calls.append((ids.shape, bool(normalize), type_ids is not None))
        time.sleep(self.delay)   # the device forward releases the GIL like this
        with self.lock:
            self.active -= 1
        if self.fail_on is not None and (ids == self.fail_on).any():
            raise RuntimeError("forward failed")
        # a row's "embedding": a function of its own tokens and length only
        out = np.zeros((ids.shape[0], 4), np.float32)
        for i in range(ids.shape[0]):
            row = ids[i, :lengths[i]].astype(np.float64)
            out[i] = [row.sum(), (row * np.arange(1, len(row) + 1)).sum(), float(lengths[i]), 2.0 if normalize else 1.0]
        return out


def _req(rng, n_rows, max_len):
    lens = rng.integers(3, max_len + 1, n_rows).astype(np.int32)
    L = int(lens.max())
    ids = np.zeros((n_rows, L), np.int32)
    for i, l in enumerate(lens):
        ids[i, :l] = rng.integers(5, 1000, l)
    return ids, lens


def test_concurrent_calls_share_forwards_and_get_their_own_rows():
    enc = _FakeEncoder()
    comb = _ForwardCombiner(enc, pad_id=0, max_tokens=2048)
    rng = np.random.default_rng(0)
    reqs = [_req(rng, int(rng.integers(1, 3)), 30) for _ in range(160)]
    expect = [_FakeEncoder(0).forward(i, l) for i, l in reqs]
    got = [None] * len(reqs)

    def worker(w):
        for j in range(w, len(reqs), 8):
            got[j] = comb.run(reqs[j][0], reqs[j][1], None, False)

    th = [threading.Thread(target=worker, args=(w,)) for w in range(8)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in th), "deadlock"
    dt = time.perf_counter() - t0
    for g, e in zip(got, expect):
        assert np.array_equal(g, e)
    assert enc.max_active == 1                      # one forward at a time
    assert comb.served == len(reqs) and comb.forwards == len(enc.calls)
    assert comb.forwards <= len(reqs) * 0.6, (comb.forwards, len(reqs))   # merging happened
    assert dt < len(reqs) * enc.delay * 0.7       # and it shows: faster than one forward per request
    assert not comb.running and comb.queue == []


def test_requests_merge_only_when_the_arithmetic_is_unchanged():
    enc = _FakeEncoder(0)
    comb = _ForwardCombiner(enc, pad_id=0, max_tokens=256)
    mk = lambda rows, L: (np.ones((rows, L), np.int32), np.full(rows, L, np.int32))
    batch = [comb._Req(*mk(1, 10), None, False), comb._Req(*mk(2, 31), None, False),      # pad to 32 tokens: together ...
             comb._Req(*mk(1, 40), None, False),                                          # ... with one padded to 64: 4 x 64 = 256
             comb._Req(*mk(1, 12), None, True),                                           # other flag: its own forward
             comb._Req(*mk(1, 9), np.zeros((1, 9), np.int32), False),                     # typed: its own forward
             comb._Req(*mk(6, 32), None, False), comb._Req(*mk(3, 32), None, False)]      # would pass 256 tokens: two forwards
    comb._serve(batch)
    shapes = sorted(c[0] for c in enc.calls)
    assert shapes == sorted([(4, 40), (1, 12), (1, 9), (6, 32), (3, 32)]), shapes
    assert all(r.out is not None and r.out.shape[0] == r.ids.shape[0] for r in batch)
    for r in batch:   # every caller's rows are a function of its own tokens only
        assert np.array_equal(r.out, _FakeEncoder(0).forward(r.ids, r.lens, r.types, r.norm))
    # a single request larger than the budget still runs (alone)
    enc.calls.clear()
    big = comb._Req(*mk(20, 64), None, False)
    comb._serve([big])
    assert enc.calls[0][0] == (20, 64) and big.out.shape == (20, 4)


def test_a_failed_forward_reaches_every_caller_it_served_and_the_queue_keeps_working():
    enc = _FakeEncoder(0.002, fail_on=999_999)
    comb = _ForwardCombiner(enc, pad_id=0)
    rng = np.random.default_rng(1)
    good = [_req(rng, 1, 20) for _ in range(40)]
    bad = (np.array([[7, 999_999, 8]], np.int32), np.array([3], np.int32))
    errors, results = [], []

    def worker(w):
        for j in range(w, len(good), 4):
            try:
                results.append((j, comb.run(good[j][0], good[j][1], None, False)))
            except RuntimeError:
                errors.append(j)          # shared a forward with the bad request

    def bad_worker():
        for _ in range(5):
            with pytest.raises(RuntimeError):
                comb.run(bad[0], bad[1], None, False)

    th = [threading.Thread(target=worker, args=(w,)) for w in range(4)] + [threading.Thread(target=bad_worker)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in th), "deadlock"
    assert len(errors) + len(results) == len(good)
    for j, out in results:
        assert np.array_equal(out, _FakeEncoder(0).forward(*good[j]))
    assert np.array_equal(comb.run(good[0][0], good[0][1], None, False), _FakeEncoder(0).forward(*good[0]))   # still alive
