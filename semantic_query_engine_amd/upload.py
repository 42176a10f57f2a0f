"""The second caller of the hot path: the upload microservice's forms of the same calls
(``app/embedding_gen.py``), backed by libsqe.  They differ from the ``main.py`` forms in
:mod:`retrieval` exactly where the reference differs:

    ollama_embed_text(text, model)                embedding_gen.py:143-166
        blank text -> ``[0.0] * 1024`` (no request); ANY failure -> print + ``[0.0] * 1024``;
        a result whose length is not 1024 is returned as is, with a warning
    embed_texts_in_batches(texts)                 embedding_gen.py:169-190
        ``[]`` -> ``zeros((0, 1024), float32)`` (main.py returns ``np.array([])``); failed rows are zero rows
    init_user_index(user_id)                      embedding_gen.py:83-122
    bulk_index_embeddings(user_id, doc_id, embeddings, chunks)    embedding_gen.py:196-257
        per-user index ``f"{BASE}-{user_id}"`` (:211), ``_id = f"{doc_id}_{i}"`` with i the chunk's index
        INSIDE the document (:219-221; main.py:318,325 uses the global row index), zip() semantics,
        fp32 normalisation x / (||x|| + 1e-9) (:214-215) on the GPU

Zero vectors are legal rows: they normalise to zero (no NaN, main.py:315-316) and score cosine 0.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np

from . import retrieval as RT

BATCH_SIZE = 64
CHUNK_SIZE = 512
EMBED_DIM = 1024
EMBED_MODEL_NAME = os.getenv("OLLAMA_EMBED_MODEL", "mxbai-embed-large:latest")
BASE_OPENSEARCH_INDEX_NAME = os.getenv("OPENSEARCH_INDEX_NAME", "")

os_client: Optional[RT.GpuSearchClient] = None


def configure_client(client: Optional[RT.GpuSearchClient]) -> None:
    """Install the object standing where ``os_client`` stood (embedding_gen.py:70-80)."""
    global os_client
    os_client = client


async def ollama_embed_text(text: str, model: str = EMBED_MODEL_NAME) -> List[float]:
    if not text.strip():
        return [0.0] * EMBED_DIM
    try:
        emb = RT._require_embedder().embed([text])[0].tolist()
        if len(emb) != EMBED_DIM:
            print(f"[WARNING] Mismatch embedding size. Expected {EMBED_DIM}, got {len(emb)}")
        return emb
    except Exception as exc:
        print(f"[ERROR] Ollama embedding error: {exc}")
        return [0.0] * EMBED_DIM


async def embed_texts_in_batches(texts: List[str]) -> np.ndarray:
    """Order-preserving; blank texts and texts of a failing batch come back as zero rows.  The live texts of
    each 64-text batch go through ONE encoder call (the reference makes one HTTP request per text)."""
    if not texts:
        return np.zeros((0, EMBED_DIM), dtype=np.float32)
    rows: List[List[float]] = []
    for i in range(0, len(texts), BATCH_SIZE):
        batch = texts[i:i + BATCH_SIZE]
        live = [j for j, t in enumerate(batch) if t.strip()]
        got = {}
        if live:
            try:
                embs = RT._require_embedder().embed([batch[j] for j in live])
                for j, e in zip(live, embs):
                    if len(e) != EMBED_DIM:
                        print(f"[WARNING] Mismatch embedding size. Expected {EMBED_DIM}, got {len(e)}")
                    got[j] = e.tolist()
            except Exception as exc:
                print(f"[ERROR] Ollama embedding error: {exc}")
        rows.extend(got.get(j, [0.0] * EMBED_DIM) for j in range(len(batch)))
    return np.array(rows, dtype=np.float32)


def init_user_index(user_id: str):
    if not os_client:
        print("[WARNING] No OpenSearch client => skipping index creation.")
        return
    index_name = f"{BASE_OPENSEARCH_INDEX_NAME}-{user_id}"
    if os_client.exists(index_name):
        print(f"[INFO] Index '{index_name}' already exists.")
        return
    try:
        os_client.index(index_name)          # cosine, dim = client.dim: the mapping of embedding_gen.py:96-119
        print(f"[INFO] Created user-specific index '{index_name}'.")
    except Exception as e:
        print(f"[ERROR] Failed creating index '{index_name}': {e}")


def bulk_index_embeddings(user_id: str, doc_id: str, embeddings: np.ndarray, chunks: List[str]):
    if not os_client or embeddings.size == 0:
        print("[ERROR] Missing OpenSearch client or embeddings => cannot index.")
        return
    index_name = f"{BASE_OPENSEARCH_INDEX_NAME}-{user_id}"
    init_user_index(user_id)
    try:
        n = min(len(chunks), embeddings.shape[0])
        docs = [{"doc_id": doc_id, "text": chunks[i]} for i in range(n)]
        done = RT._commit_documents(os_client.index(index_name), embeddings, docs, lambda i, d: f"{doc_id}_{i}")
        print(f"[OpenSearch] Bulk indexed {done} chunk docs for user={user_id}, doc_id={doc_id}")
    except Exception as exc:
        print(f"[OpenSearch] Bulk error (user={user_id} doc_id={doc_id}): {exc}")
