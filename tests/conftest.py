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
    """False only when the HIP library LOADS and reports no device (the CPU build container).  A library that fails to
    load is not "no GPU": the gpu tests then run and fail loudly instead of being skipped in silence."""
    try:
        from anorag_hip import _lib
        return _lib.device_count() > 0
    except Exception as e:  # missing / truncated libanorag_hip.so, unresolved symbol, ...
        print(f"conftest: libanorag_hip.so could not be loaded ({e}); gpu tests will run and fail", file=sys.stderr)
        return True


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
