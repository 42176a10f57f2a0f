"""MI355X-native embedding + cosine k-NN + cosine-cache engine.

Drop-in for the Ollama-embed / OpenSearch-HNSW / Redis-scan leg of the reference's /ask
pipeline (/root/reference/app/main.py:56-180, 250-373).  The arithmetic lives in
``libsqe.so`` (hand-written HIP for gfx950, see ``csrc/``); this package is the ctypes
binding (``_native``), a thin object layer (``engine``) and the host-side mirror of the
reference interface (``retrieval``).  There is no CPU fallback: importing ``_native``
fails loudly when the library has not been built.
"""
from .engine import (EXCHANGE_AUTO, EXCHANGE_COPY, EXCHANGE_RCCL, INDEX_FLAT, INDEX_IVF_FLAT, SCAN_BF16_RESCORE,
                     SCAN_INT8_RESCORE, CacheMatrix, Context, VectorIndex)

__all__ = ["Context", "VectorIndex", "CacheMatrix", "INDEX_FLAT", "INDEX_IVF_FLAT",
           "SCAN_BF16_RESCORE", "SCAN_INT8_RESCORE", "EXCHANGE_AUTO", "EXCHANGE_RCCL", "EXCHANGE_COPY"]
