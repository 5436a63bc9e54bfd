"""Drop-in for the hot-path part of the reference's ``retrieval`` package: only ``hybrid_search`` is provided
(reference retrieval/__init__.py:14-16 exports HybridSearcher from there)."""
from .hybrid_search import HybridSearcher, create_hybrid_searcher  # noqa: F401

__all__ = ["HybridSearcher", "create_hybrid_searcher"]
