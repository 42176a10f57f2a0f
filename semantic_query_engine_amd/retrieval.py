"""Host-side mirror of the reference's retrieval interface, backed by libsqe (MI355X).

Same names, argument meaning, return shapes and error behaviour as the reference's
``app/main.py`` so ``RAGModel`` / the routes can call them unchanged:

    OpenSearchIndexer(client, index_name)        main.py:291-298
        .has_any_data() -> bool                  main.py:300-307
        .add_embeddings(embeddings, docs)        main.py:309-338
        .search(query_emb, k=3)                  main.py:347-373
    cosine_similarity(a, b) -> float             main.py:59-64
    lfu_cache_get(query_emb) -> Optional[str]    main.py:67-98
    lfu_cache_put(query_emb, response)           main.py:121-128

The OpenSearch ``client`` argument becomes a :class:`GpuSearchClient` (named indexes, each a
libsqe vector index in HBM plus a host docstore of ``_source`` dicts); Redis becomes a
:class:`SemanticLfuCache` (cache matrix resident in HBM, LFU bookkeeping on the host).
Everything numeric runs in the HIP library; nothing here falls back to NumPy.
"""
from __future__ import annotations

import threading
from typing import Dict, List, Optional, Tuple

import numpy as np

from .engine import INDEX_FLAT, CacheMatrix, Context, VectorIndex

# Constants of the reference (main.py:35-44), names kept.
BATCH_SIZE = 64
CHUNK_SIZE = 512
EMBED_DIM = 1024
REDIS_MAX_ITEMS = 1000
CACHE_SIM_THRESHOLD = 0.96

_default_ctx: Optional[Context] = None
_ctx_lock = threading.Lock()


def default_context(device: Optional[int] = None, devices=None) -> Context:
    """Process-wide context.  ``devices=[0, 1, ...]``: this one process drives all of them (what the
    reference's single uvicorn process needs); otherwise one device -- ``device``, or LOCAL_RANK in the
    one-process-per-GPU form."""
    global _default_ctx
    with _ctx_lock:
        if _default_ctx is None:
            import os
            if devices is not None:
                _default_ctx = Context(devices=devices)
            else:
                dev = device if device is not None else int(os.environ.get("LOCAL_RANK", "0"))
                _default_ctx = Context(dev)
        return _default_ctx


# ------------------------------------------------------------------------------ index
class _GpuNamedIndex:
    """One OpenSearch index: vectors in HBM, ``_source`` documents on the host."""

    def __init__(self, ctx: Context, dim: int, kind: int, nlist: int):
        self.vectors = VectorIndex(ctx, dim, kind, nlist)
        self.sources: List[Dict[str, str]] = []       # row -> {"doc_id", "text"}
        self.row_of_id: Dict[str, int] = {}           # OpenSearch _id -> row
        self.lock = threading.Lock()


class GpuSearchClient:
    """Stands where the ``OpenSearch`` client object stood (main.py:250-259): a registry of
    named cosine indexes (the per-user ``<base>-<user_id>`` indexes of
    embedding_gen.py:211 are just more names)."""

    def __init__(self, ctx: Optional[Context] = None, dim: int = EMBED_DIM, kind: int = INDEX_FLAT,
                 nlist: int = 0, devices=None):
        """``devices=[0, ..., 7]``: every index of this client is sharded over those GPUs of the node, driven
        from this one process (ignored when a context is passed)."""
        self.ctx = ctx or default_context(devices=devices)
        self.dim, self.kind, self.nlist = dim, kind, nlist
        self._indexes: Dict[str, _GpuNamedIndex] = {}
        self._lock = threading.Lock()

    def index(self, name: str) -> _GpuNamedIndex:
        with self._lock:
            if name not in self._indexes:
                self._indexes[name] = _GpuNamedIndex(self.ctx, self.dim, self.kind, self.nlist)
            return self._indexes[name]

    def exists(self, name: str) -> bool:
        with self._lock:
            return name in self._indexes

    def count(self, index: str) -> Dict[str, int]:
        """Shape of ``client.count(index=...)`` (main.py:304-305)."""
        return {"count": len(self.index(index).vectors)}

    # ---- persistence: OpenSearch kept the index across restarts, so ``has_any_data()`` could skip the
    # rebuild (main.py:422-424).  Here one named index = <dir>/<name>.sqeidx (vectors, sqe_index_save) +
    # <dir>/<name>.docs.jsonl (row order: {"_id", "doc_id", "text"}).
    def save_index(self, name: str, directory: str) -> None:
        import json
        import os
        idx = self.index(name)
        os.makedirs(directory, exist_ok=True)
        with idx.lock:
            idx.vectors.save(os.path.join(directory, name + ".sqeidx"))
            id_of_row = {row: os_id for os_id, row in idx.row_of_id.items()}
            with open(os.path.join(directory, name + ".docs.jsonl"), "w", encoding="utf-8") as f:
                for row, src in enumerate(idx.sources):
                    f.write(json.dumps({"_id": id_of_row[row], "doc_id": src["doc_id"], "text": src["text"]}) + "\n")

    def load_index(self, name: str, directory: str) -> bool:
        """Load a saved index under ``name``; False when there is nothing to load."""
        import json
        import os
        vp, dp = os.path.join(directory, name + ".sqeidx"), os.path.join(directory, name + ".docs.jsonl")
        if not (os.path.exists(vp) and os.path.exists(dp)):
            return False
        named = _GpuNamedIndex.__new__(_GpuNamedIndex)
        named.vectors = VectorIndex.load(self.ctx, vp)
        named.sources, named.row_of_id, named.lock = [], {}, threading.Lock()
        with open(dp, "r", encoding="utf-8") as f:
            for row, line in enumerate(f):
                d = json.loads(line)
                named.sources.append({"doc_id": d["doc_id"], "text": d["text"]})
                named.row_of_id[d["_id"]] = row
        if len(named.sources) != len(named.vectors):
            raise ValueError(f"{name}: {len(named.sources)} documents for {len(named.vectors)} vectors")
        with self._lock:
            self._indexes[name] = named
        return True


def _commit_documents(idx: "_GpuNamedIndex", embeddings: np.ndarray, docs: List[Dict[str, str]], id_of) -> int:
    """The "index" op of the bulk call (main.py:318-338): insert, or overwrite an existing ``_id``.

    Order matters: (1) validate and convert the vectors, (2) plan against the docstore WITHOUT touching it,
    (3) run the device add / update, (4) only then commit ``sources`` / ``row_of_id``.  A failing device call
    (wrong dimension, allocation failure) therefore leaves host docstore and vector rows in step: every later
    add still lands at vector row == sources row."""
    vecs = np.ascontiguousarray(embeddings, dtype=np.float32)
    if vecs.ndim != 2 or vecs.shape[1] != idx.vectors.dim:
        raise ValueError(f"embeddings must be [n, {idx.vectors.dim}], got {vecs.shape}")
    n = min(len(docs), vecs.shape[0])                 # zip() semantics of main.py:318
    with idx.lock:
        base = len(idx.sources)
        new_src: List[Dict[str, str]] = []
        new_ids: Dict[str, int] = {}                  # _id -> position in this call's insert list
        new_from: List[int] = []                      # embedding row of each insert (the LAST writer of its _id)
        upd: Dict[int, tuple] = {}                    # stored row -> (source, embedding row); last writer wins
        for i in range(n):
            os_id = id_of(i, docs[i])
            src = {"doc_id": docs[i]["doc_id"], "text": docs[i]["text"]}
            row = idx.row_of_id.get(os_id)
            if row is not None:
                upd[row] = (src, i)
            elif os_id in new_ids:                    # same _id twice in one call: the later document replaces the earlier
                new_src[new_ids[os_id]], new_from[new_ids[os_id]] = src, i
            else:
                new_ids[os_id] = len(new_src)
                new_src.append(src)
                new_from.append(i)
        # normalisation x / (||x|| + 1e-9) (main.py:315-316) happens on the GPU
        if new_from:
            contiguous = new_from == list(range(new_from[0], new_from[0] + len(new_from)))
            idx.vectors.add(vecs[new_from[0]:new_from[0] + len(new_from)] if contiguous else vecs[new_from])
        try:
            if upd:
                rows = sorted(upd)
                idx.vectors.update(np.array(rows, np.int64), vecs[[upd[r][1] for r in rows]])
        finally:
            # the appended vector rows exist whatever the update did: their documents must exist too
            for os_id, pos in new_ids.items():
                idx.row_of_id[os_id] = base + pos
            idx.sources.extend(new_src)
        for row, (src, _i) in upd.items():
            idx.sources[row] = src
    return n


class OpenSearchIndexer:
    """Drop-in for the reference class of the same name (main.py:291-373)."""

    def __init__(self, client: GpuSearchClient, index_name: str):
        self.client = client
        self.index_name = index_name

    def has_any_data(self) -> bool:
        if not self.client:
            return False
        try:
            resp = self.client.count(index=self.index_name)
            return resp["count"] > 0
        except Exception:
            return False

    def add_embeddings(self, embeddings: np.ndarray, docs: List[Dict[str, str]]):
        if not self.client or embeddings.size == 0:
            print("[OpenSearchIndexer] No embeddings or no OpenSearch client.")
            return
        try:
            idx = self.client.index(self.index_name)
            n = _commit_documents(idx, embeddings, docs, lambda i, d: f"{d['doc_id']}_{i}")   # _id rule of main.py:325
            print(f"[OpenSearchIndexer] Inserted {n} docs, errors=[]")
        except Exception as e:
            print(f"[OpenSearchIndexer] Bulk indexing error: {e}")

    def search(self, query_emb: np.ndarray, k: int = 3) -> List[Tuple[Dict[str, str], float]]:
        if not self.client or query_emb.size == 0:
            return []
        try:
            idx = self.client.index(self.index_name)
            q = np.ascontiguousarray(query_emb, dtype=np.float32)
            cos, ids = idx.vectors.search(q[0:1], k)          # row 0 only (main.py:355)
            rows = [int(r) for r in ids[0] if r >= 0]
            embs = idx.vectors.get_rows(rows) if rows else np.zeros((0, idx.vectors.dim), np.float32)
            results = []
            for j, row in enumerate(rows):
                src = dict(idx.sources[row])
                src["embedding"] = embs[j].tolist()           # _source carries the stored vector
                # nmslib cosinesimil _score = 1 / (1 + (1 - cos))
                results.append((src, float(1.0 / (2.0 - float(cos[0, j])))))
            print(f"[OpenSearchIndexer] Found {len(results)} relevant results.")
            return results
        except Exception as e:
            print(f"[OpenSearchIndexer] Search error: {e}")
            return []

    # batched form of the same call (the GPU path's throughput is in B > 1)
    def search_batch(self, query_embs: np.ndarray, k: int = 3) -> Tuple[np.ndarray, np.ndarray]:
        idx = self.client.index(self.index_name)
        return idx.vectors.search(np.ascontiguousarray(query_embs, dtype=np.float32), k)


# ------------------------------------------------------------------------------ cache
class SemanticLfuCache:
    """The Redis LIST ``query_cache_lfu`` of main.py:56-128 with the scan on the GPU.

    List position 0 is the newest entry (``lpush``, main.py:128).  The embeddings live in
    slots of a resident cache matrix; ``_order[i]`` is the slot of list position ``i``."""

    def __init__(self, ctx: Optional[Context] = None, max_items: int = REDIS_MAX_ITEMS,
                 threshold: float = CACHE_SIM_THRESHOLD, dim: int = EMBED_DIM):
        self.ctx = ctx or default_context()
        self.max_items, self.threshold, self.dim = max_items, threshold, dim
        self.matrix = CacheMatrix(self.ctx, max_items, dim)
        self._order: List[int] = []                    # list position -> slot
        self._entries: Dict[int, Dict] = {}            # slot -> {"response", "freq"}
        self._free = list(range(max_items - 1, -1, -1))
        self._lock = threading.Lock()
        self.last_index, self.last_sim = -1, -1.0

    def __len__(self) -> int:
        return len(self._order)

    def get(self, query_emb: np.ndarray) -> Optional[str]:
        """lfu_cache_get (main.py:67-98)."""
        with self._lock:
            self.last_index, self.last_sim = -1, -1.0
            if not self._order:
                return None
            best_sim, best_index = self.matrix.best(self._order, query_emb[0])
            self.last_index, self.last_sim = best_index, best_sim
            if best_sim < self.threshold or best_index < 0:
                return None
            entry = self._entries[self._order[best_index]]
            entry["freq"] = entry.get("freq", 1) + 1
            return entry["response"]

    def _remove_least_frequent_item(self) -> None:
        """main.py:101-118: first strict minimum of freq in list order."""
        if not self._order:
            return
        min_freq, min_index = float("inf"), -1
        for i, slot in enumerate(self._order):
            freq = self._entries[slot].get("freq", 1)
            if freq < min_freq:
                min_freq, min_index = freq, i
        if min_index >= 0:
            slot = self._order.pop(min_index)
            del self._entries[slot]
            self._free.append(slot)

    def put(self, query_emb: np.ndarray, response: str) -> None:
        """lfu_cache_put (main.py:121-128)."""
        with self._lock:
            if len(self._order) >= self.max_items:
                self._remove_least_frequent_item()
            slot = self._free.pop()
            self.matrix.set_slot(slot, np.asarray(query_emb, dtype=np.float32)[0])
            self._entries[slot] = {"response": response, "freq": 1}
            self._order.insert(0, slot)

    def freqs(self) -> List[int]:
        return [self._entries[s].get("freq", 1) for s in self._order]

    def responses(self) -> List[str]:
        return [self._entries[s]["response"] for s in self._order]


_default_cache: Optional[SemanticLfuCache] = None


def _cache() -> SemanticLfuCache:
    global _default_cache
    if _default_cache is None:
        _default_cache = SemanticLfuCache()
    return _default_cache


def cosine_similarity(a: np.ndarray, b: np.ndarray) -> float:
    """main.py:59-64 on the GPU (fp32, zero-norm rule); returns a Python float."""
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(1, -1)
    return float(default_context().cosine_all(a, np.asarray(b, dtype=np.float32))[0])


def lfu_cache_get(query_emb: np.ndarray) -> Optional[str]:
    return _cache().get(query_emb)


def lfu_cache_put(query_emb: np.ndarray, response: str) -> None:
    _cache().put(query_emb, response)


# ------------------------------------------------------------------------------ embeddings
class Embedder:
    """Tokenizer + encoder pair standing where Ollama stood (main.py:134-145): text -> 1024 floats.
    No prefix or instruction is added to queries or passages (the reference adds none, main.py:139)."""

    def __init__(self, encoder, tokenizer, max_len: int = 512):
        self.encoder, self.tokenizer, self.max_len = encoder, tokenizer, max_len
        self._lock = threading.Lock()

    def _tokenize(self, texts: List[str]):
        ids, lens = self.tokenizer.encode_batch(texts, self.max_len)
        s = int(min(self.max_len, max(16, (int(lens.max()) + 15) // 16 * 16)))
        return ids[:, :s], lens

    def embed(self, texts: List[str]) -> np.ndarray:
        if not texts:
            return np.zeros((0, self.encoder.cfg["hidden"]), np.float32)
        ids, lens = self._tokenize(texts)
        with self._lock:
            return self.encoder.encode_ids(ids, lens)

    def embed_batches(self, texts: List[str], batch_size: int = 64) -> np.ndarray:
        """Order-preserving bulk form: the WordPiece tokenisation of batch i + 1 (host, GIL released inside
        the C++ tokenizer) runs on a worker thread while the GPU encodes batch i."""
        if not texts:
            return np.zeros((0, self.encoder.cfg["hidden"]), np.float32)
        from concurrent.futures import ThreadPoolExecutor
        chunks = [texts[i:i + batch_size] for i in range(0, len(texts), batch_size)]
        out = []
        with ThreadPoolExecutor(max_workers=1) as pool:
            nxt = pool.submit(self._tokenize, chunks[0])
            for i in range(len(chunks)):
                ids, lens = nxt.result()
                if i + 1 < len(chunks):
                    nxt = pool.submit(self._tokenize, chunks[i + 1])
                with self._lock:
                    out.append(self.encoder.encode_ids(ids, lens))
        return np.concatenate(out, axis=0)


_embedder: Optional[Embedder] = None


def configure_embedder(embedder: Embedder) -> None:
    """Install the process-wide embedder used by the reference-named functions below."""
    global _embedder
    _embedder = embedder


def _require_embedder() -> Embedder:
    if _embedder is None:
        raise RuntimeError("no embedder configured: call configure_embedder(Embedder(encoder, tokenizer))")
    return _embedder


async def ollama_embed_text(text: str, model: str = "mxbai-embed-large:latest") -> List[float]:
    """main.py:134-145: one text -> its embedding as a list of floats (``model`` kept for call compatibility)."""
    return _require_embedder().embed([text])[0].tolist()


async def embed_texts_in_batches(texts: List[str], batch_size: int = 64) -> np.ndarray:
    """main.py:148-169: order-preserving embedding of ``texts`` -> float32 [n, 1024]; [] -> np.array([])."""
    if not texts:
        return np.array([])
    return _require_embedder().embed_batches(texts, batch_size).astype(np.float32)


async def embed_query(query: str) -> np.ndarray:
    """main.py:172-180: one query -> float32 [1, 1024]; blank -> size-0 array."""
    if not query.strip():
        return np.array([])
    return _require_embedder().embed([query]).astype(np.float32)
