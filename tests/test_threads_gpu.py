"""Thread safety of the C ABI: the reference calls `add_embeddings` from a thread-pool thread
(main.py:454-455) while `search` runs on the event-loop thread (main.py:499), and ctypes drops the GIL.
GPU only."""
import threading

import numpy as np
import pytest

from oracle import retrieval as R

pytestmark = pytest.mark.gpu


def test_concurrent_add_and_search_and_cache():
    from semantic_query_engine_amd import Context
    from semantic_query_engine_amd.retrieval import GpuSearchClient, OpenSearchIndexer, SemanticLfuCache
    ctx = Context(0)
    dim = 256
    client = GpuSearchClient(ctx, dim=dim)
    ix = OpenSearchIndexer(client, "idx")
    rng = np.random.default_rng(0)
    base = rng.standard_normal((2000, dim)).astype(np.float32)
    ix.add_embeddings(base, [{"doc_id": f"d{i}", "text": f"t{i}"} for i in range(2000)])
    extra = rng.standard_normal((40, 500, dim)).astype(np.float32)
    errors, found = [], []

    def adder():
        try:
            for b in range(extra.shape[0]):
                docs = [{"doc_id": f"x{b}_{i}", "text": "n"} for i in range(500)]
                ix.add_embeddings(extra[b], docs)
        except Exception as e:          # pragma: no cover
            errors.append(e)

    def searcher():
        try:
            for it in range(150):
                j = (it * 37) % 2000
                hits = ix.search(base[j:j + 1], k=3)
                found.append(hits[0][0]["text"] == f"t{j}" and abs(hits[0][1] - 1.0) < 1e-3)
        except Exception as e:          # pragma: no cover
            errors.append(e)

    def cacher():
        try:
            cache = SemanticLfuCache(ctx, max_items=50, dim=dim)
            for it in range(200):
                v = base[it:it + 1]
                cache.put(v, f"r{it}")
                assert cache.get(v) == f"r{it}"
        except Exception as e:          # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=f) for f in (adder, searcher, searcher, cacher)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert all(found) and len(found) == 300
    # final state equals a sequential build
    assert client.count("idx")["count"] == 2000 + 40 * 500
    allx = np.concatenate([base, extra.reshape(-1, dim)], 0)
    q = rng.standard_normal((8, dim)).astype(np.float32)
    cos, ids = ix.search_batch(q, 10)
    ec, ei = R.exact_topk(R.normalize_rows(allx), R.normalize_rows(q), 10)
    assert np.array_equal(ids, ei)
    assert np.abs(cos - ec).max() < 1e-3
