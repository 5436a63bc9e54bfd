"""pytest configuration: registers the `gpu` marker and puts the product tree and the oracle on sys.path.

`-m "not gpu"` tests run in the CPU-only build container; `-m gpu` tests need one MI355X and call the
HIP path through the C ABI (the oracle under oracle/ is only ever the checker).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ano-rag_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


def _has_gpu():
    try:
        from anorag_hip import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
