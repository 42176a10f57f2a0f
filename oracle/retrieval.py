"""NumPy restatement of the retrieval arithmetic of the reference hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines (``/root/reference/app/main.py``) whose behaviour it restates.
"""
from __future__ import annotations

import json
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# Constants of the reference (main.py:35-44).
BATCH_SIZE = 64
CHUNK_SIZE = 512
EMBED_DIM = 1024
REDIS_MAX_ITEMS = 1000
CACHE_SIM_THRESHOLD = 0.96


# --------------------------------------------------------------------------- cosine
def cosine_similarity(a: np.ndarray, b: np.ndarray) -> float:
    """main.py:59-64.  fp32 norms and dot, zero norm -> 0.0, result widened to float."""
    norm_a = np.linalg.norm(a)
    norm_b = np.linalg.norm(b)
    if norm_a == 0.0 or norm_b == 0.0:
        return 0.0
    return float(np.dot(a, b) / (norm_a * norm_b))


def cosine_best(mat: np.ndarray, q: np.ndarray) -> Tuple[float, int]:
    """The scan loop of ``lfu_cache_get`` (main.py:73-87): first strict maximum of
    ``cosine_similarity(q, mat[i])`` starting from best_sim=-1.0, best_index=-1.
    A NaN similarity never wins (``nan > x`` is False)."""
    best_sim, best_index = -1.0, -1
    for i in range(mat.shape[0]):
        sim = cosine_similarity(q, mat[i])
        if sim > best_sim:
            best_sim, best_index = sim, i
    return best_sim, best_index


def cosine_all(mat: np.ndarray, q: np.ndarray) -> np.ndarray:
    """Vectorised float64 cosine of q against every row (zero norm -> 0.0), used to
    bound the fp32 result of the GPU cache scan."""
    m = mat.astype(np.float64)
    v = q.astype(np.float64)
    nm = np.sqrt((m * m).sum(1))
    nv = np.sqrt((v * v).sum())
    with np.errstate(invalid="ignore", divide="ignore"):
        s = (m @ v) / (nm * nv)
    s[(nm == 0) | (nv == 0)] = 0.0
    return s


# ------------------------------------------------------------------------ normalise
def normalize_rows(e: np.ndarray) -> np.ndarray:
    """main.py:315-316 (index side) and main.py:353-354 (query side):
    ``e / (||e||_2 + 1e-9)`` evaluated in float32 (NumPy keeps float32 because the
    Python scalar 1e-9 is weakly typed).  All-zero rows stay zero, no NaN."""
    e = np.asarray(e, dtype=np.float32)
    norms = np.linalg.norm(e, axis=1, keepdims=True)
    return (e / (norms + np.float32(1e-9))).astype(np.float32)


# --------------------------------------------------------------------------- top-k
def exact_topk(xn: np.ndarray, qn: np.ndarray, k: int, block: int = 65536
               ) -> Tuple[np.ndarray, np.ndarray]:
    """Exact cosine top-k of already-normalised float32 rows, scored in float64.

    Restates what OpenSearch k-NN ``cosinesimil`` returns for a ``knn`` query
    (main.py:356-367) under the exactness assumption of SURVEY 8c: best first,
    ties broken by the lowest row id (stable argsort).  Returns (cos [B,k] float64,
    ids [B,k] int64); ids are -1 and cos -inf past the number of rows."""
    xn = np.asarray(xn, dtype=np.float32)
    qn = np.asarray(qn, dtype=np.float32)
    n, b = xn.shape[0], qn.shape[0]
    kk = min(k, n)
    q64 = qn.astype(np.float64)
    best_s = np.full((b, 0), -np.inf)
    best_i = np.zeros((b, 0), dtype=np.int64)
    for lo in range(0, n, block):
        hi = min(n, lo + block)
        s = q64 @ xn[lo:hi].astype(np.float64).T                       # [B, blk]
        ids = np.broadcast_to(np.arange(lo, hi, dtype=np.int64), s.shape)
        s = np.concatenate([best_s, s], axis=1)
        ids = np.concatenate([best_i, ids], axis=1)
        # candidates are in ascending-id order within each part and the carried part
        # has lower ids than the new block, so a stable sort keeps lowest id first.
        order = np.argsort(-s, axis=1, kind="stable")[:, :kk]
        best_s = np.take_along_axis(s, order, axis=1)
        best_i = np.take_along_axis(ids, order, axis=1)
    cos = np.full((b, k), -np.inf)
    idx = np.full((b, k), -1, dtype=np.int64)
    cos[:, :kk] = best_s
    idx[:, :kk] = best_i
    return cos, idx


def knn_search(x_raw: np.ndarray, q_raw: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """add_embeddings + search composed (main.py:309-373): normalise both sides in
    float32 exactly as the reference does, then exact top-k."""
    return exact_topk(normalize_rows(x_raw), normalize_rows(q_raw), k)


def os_score_from_cosine(cos):
    """OpenSearch k-NN ``_score`` for nmslib ``cosinesimil``: 1/(1+d), d = 1-cos
    ([EXT]); monotone in cosine so ordering is unchanged."""
    return 1.0 / (2.0 - np.asarray(cos, dtype=np.float64))


def recall_at_k(ids: np.ndarray, ref_ids: np.ndarray) -> float:
    """|ids ∩ ref| / k averaged over queries (padding -1 ignored on the ref side)."""
    hits, total = 0, 0
    for a, r in zip(ids, ref_ids):
        r = r[r >= 0]
        hits += len(set(a.tolist()) & set(r.tolist()))
        total += len(r)
    return hits / max(total, 1)


# ---------------------------------------------------------------------------- IVF
def ivf_search(xn: np.ndarray, qn: np.ndarray, centroids: np.ndarray, assign: np.ndarray,
               k: int, nprobe: int) -> Tuple[np.ndarray, np.ndarray]:
    """IVF-flat semantics: probe the ``nprobe`` centroids with the largest inner
    product (ties -> lowest list id), exact top-k over the rows assigned to them."""
    b = qn.shape[0]
    cs = qn.astype(np.float64) @ centroids.astype(np.float64).T
    probes = np.argsort(-cs, axis=1, kind="stable")[:, :nprobe]
    cos = np.full((b, k), -np.inf)
    idx = np.full((b, k), -1, dtype=np.int64)
    for i in range(b):
        rows = np.nonzero(np.isin(assign, probes[i]))[0]
        if rows.size == 0:
            continue
        s = xn[rows].astype(np.float64) @ qn[i].astype(np.float64)
        order = np.argsort(-s, kind="stable")[:k]
        cos[i, :order.size] = s[order]
        idx[i, :order.size] = rows[order]
    return cos, idx


# ---------------------------------------------------------------------- LFU cache
class LfuCacheOracle:
    """main.py:67-128 with the Redis LIST replaced by a Python list of JSON strings
    (index 0 = newest, because ``lpush`` inserts at the head, main.py:128)."""

    def __init__(self, max_items: int = REDIS_MAX_ITEMS, threshold: float = CACHE_SIM_THRESHOLD):
        self.items: List[str] = []
        self.max_items = max_items
        self.threshold = threshold
        self.last_index = -1
        self.last_sim = -1.0

    def get(self, query_emb: np.ndarray) -> Optional[str]:
        """main.py:67-98."""
        self.last_index, self.last_sim = -1, -1.0
        if not self.items:
            return None
        query_vec = query_emb[0]
        best_sim, best_index, best_entry = -1.0, -1, None
        for i, item in enumerate(self.items):
            entry = json.loads(item)
            cached = np.array(entry["embedding"], dtype=np.float32)
            sim = cosine_similarity(query_vec, cached)
            if sim > best_sim:
                best_sim, best_index, best_entry = sim, i, entry
        self.last_index, self.last_sim = best_index, best_sim
        if best_sim < self.threshold:
            return None
        if best_entry:
            best_entry["freq"] = best_entry.get("freq", 1) + 1
            self.items[best_index] = json.dumps(best_entry)
            return best_entry["response"]
        return None

    def _remove_least_frequent_item(self) -> None:
        """main.py:101-118: first strict minimum of freq; LREM count=1 removes the
        first list element equal to that JSON string (head to tail)."""
        if not self.items:
            return
        min_freq, min_index = float("inf"), -1
        for i, item in enumerate(self.items):
            freq = json.loads(item).get("freq", 1)
            if freq < min_freq:
                min_freq, min_index = freq, i
        if min_index >= 0:
            self.items.remove(self.items[min_index])

    def put(self, query_emb: np.ndarray, response: str) -> None:
        """main.py:121-128."""
        entry = {"embedding": query_emb.tolist()[0], "response": response, "freq": 1}
        if len(self.items) >= self.max_items:
            self._remove_least_frequent_item()
        self.items.insert(0, json.dumps(entry))

    def freqs(self) -> List[int]:
        return [json.loads(it).get("freq", 1) for it in self.items]

    def responses(self) -> List[str]:
        return [json.loads(it)["response"] for it in self.items]


# ------------------------------------------------------------------------ chunker
def basic_cleaning(text: str) -> str:
    """main.py:379-380."""
    return text.replace("\n", " ").strip()


def chunk_text(text: str, chunk_size: int = CHUNK_SIZE) -> List[str]:
    """main.py:383-393 (dup embedding_gen.py:128-137): whitespace split, windows of
    ``chunk_size`` words joined by one space."""
    words = text.split()
    return [" ".join(words[i:i + chunk_size]).strip() for i in range(0, len(words), chunk_size)]


def corpus_docs(pmc_dir: str, files: Optional[Sequence[str]] = None) -> List[Dict[str, str]]:
    """main.py:427-443: files named PMC*.txt, utf-8 with latin-1 fallback, cleaned,
    chunked, one ``{"doc_id": fname, "text": chunk}`` per chunk.  The reference walks
    ``os.listdir`` order (unspecified); the oracle sorts."""
    import os
    names = sorted(os.listdir(pmc_dir)) if files is None else list(files)
    docs: List[Dict[str, str]] = []
    for fname in names:
        if fname.startswith("PMC") and fname.endswith(".txt"):
            path = os.path.join(pmc_dir, fname)
            try:
                with open(path, "r", encoding="utf-8") as f:
                    text = f.read()
            except UnicodeDecodeError:
                with open(path, "r", encoding="latin-1") as f:
                    text = f.read()
            for c in chunk_text(basic_cleaning(text), CHUNK_SIZE):
                docs.append({"doc_id": fname, "text": c})
    return docs
