"""The combining queue behind anr_encoder_forward_shared (csrc/combine.hpp — host-only C++, no HIP in it) driven by a
stand-in forward under g++: concurrent small requests must each get exactly their own rows, share forwards only with
compatible requests (same normalize flag, same use of token types, at most 2048 padded tokens), never run two forwards at
once on one lane (and up to `lanes` side by side), never deadlock when the lead is handed on, and receive the error of a failed forward — the host logic of the
reference's query-time pattern (worker threads sharing one model, main_musique.py:487-494,
query/query_processor.py:2761-2766).  The real forward behind the same queue: tests/test_encoder_gpu.py."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("combine") / "combine_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-I", os.path.join(ROOT, "ano-rag_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "combine_test.cpp"), "-o", exe], check=True)
    return exe


def _run(exe, threads, per_thread, lanes=1):
    out = subprocess.run([exe, str(threads), str(per_thread), str(lanes)], check=True, capture_output=True, text=True,
                         timeout=120).stdout
    return json.loads(out.strip().splitlines()[-1])


@pytest.mark.parametrize("threads,per_thread,lanes", [(8, 200, 1), (32, 100, 1), (3, 150, 1), (8, 200, 2), (32, 100, 2),
                                                       (16, 200, 3), (2, 300, 2)])
def test_concurrent_requests_share_forwards_and_get_their_own_rows(harness, threads, per_thread, lanes):
    r = _run(harness, threads, per_thread, lanes)
    assert r["wrong"] == 0                                    # every caller got exactly its own rows
    assert r["served"] == r["requests"] == threads * per_thread
    assert r["max_concurrent_forwards"] <= lanes and r["lane_clash"] == 0   # one forward at a time per lane
    if threads >= 8:
        assert r["forwards"] < r["served"]                    # requests did share forwards
        assert r["max_concurrent_forwards"] == lanes          # ... and the lanes ran side by side
    assert r["max_tokens"] <= 2048                            # ... within the token budget of a merged forward
    # a poisoned request fails the forward that holds it (and nobody else's); its error text reaches every caller of it
    assert r["failed"] == r["poisoned_ok"] and (r["failed"] >= per_thread // 37 if threads > 1 else True)
    assert r["rc_big"] == 0 and r["big0"] == 120.0            # the queue is idle and usable after the storm


def test_a_single_thread_runs_one_forward_per_request(harness):
    for lanes in (1, 2):
        r = _run(harness, 1, 60, lanes)
        assert r["wrong"] == 0 and r["forwards"] == r["served"] == 60 and r["failed"] == 0
