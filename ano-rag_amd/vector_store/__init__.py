"""Drop-in for the hot-path classes of the reference's ``vector_store`` package
(reference vector_store/__init__.py:1-3): EmbeddingManager, VectorIndex, VectorRetriever."""
from .embedding_manager import EmbeddingManager
from .vector_index import VectorIndex
from .retriever import VectorRetriever

__all__ = ["EmbeddingManager", "VectorIndex", "VectorRetriever"]
