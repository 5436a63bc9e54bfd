"""Host-side bindings of libanorag_hip.so (C ABI: include/anorag.h)."""
from ._lib import AnoragError, METRIC_IP, METRIC_L2, device_count, load  # noqa: F401
from .flat_index import FlatIndex  # noqa: F401
