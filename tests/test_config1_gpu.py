"""BASELINE config 1 at the reference's real scale: 32,717 rows (the bundled corpus' chunk count) and
100 queries, top-10 ids + scores against the exact oracle -- with synthetic vectors, because neither
the corpus nor model weights can travel to the GPU box.  Also a two-thread stress of the C ABI (the
reference calls add_embeddings from a thread-pool thread while search runs on the event loop,
main.py:454-455 vs :499).  GPU only."""
import threading

import numpy as np
import pytest

from oracle import retrieval as R
from tests.gpu_util import assert_topk_matches, exact_topk_fast

pytestmark = pytest.mark.gpu


def test_corpus_scale_top10_matches_oracle():
    from semantic_query_engine_amd.retrieval import GpuSearchClient, OpenSearchIndexer
    rng = np.random.default_rng(32717)
    n, nq = 32717, 100
    # clustered like text embeddings: neighbours are genuinely close
    cen = rng.standard_normal((300, 1024)).astype(np.float32)
    x = (cen[rng.integers(0, 300, n)] + 0.5 * rng.standard_normal((n, 1024))).astype(np.float32)
    q = (x[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, 1024))).astype(np.float32)
    client = GpuSearchClient(dim=1024)
    ix = OpenSearchIndexer(client, "pmc")
    docs = [{"doc_id": f"PMC{i // 11}.txt", "text": f"chunk {i}"} for i in range(n)]
    ix.add_embeddings(x, docs)
    cos, ids = ix.search_batch(q, k=10)
    ref_cos, ref_ids = exact_topk_fast(x, q, 10)
    assert R.recall_at_k(ids, ref_ids) == 1.0
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
    assert np.abs(cos - ref_cos).max() < 1e-5
    hits = ix.search(q[7:8], k=10)                    # the reference's one-query call shape
    assert [h[0]["text"] for h in hits] == [f"chunk {i}" for i in ref_ids[7]]


def test_two_threads_add_while_searching():
    from semantic_query_engine_amd import Context, VectorIndex
    rng = np.random.default_rng(1)
    ctx = Context(0)
    idx = VectorIndex(ctx, 256)
    base = rng.standard_normal((4000, 256)).astype(np.float32)
    extra = rng.standard_normal((6000, 256)).astype(np.float32)
    q = base[:16] * 2.0
    idx.add(base)
    errors = []

    def adder():
        try:
            for i in range(0, 6000, 500):
                idx.add(extra[i:i + 500])
        except Exception as e:      # pragma: no cover
            errors.append(e)

    def searcher():
        try:
            for _ in range(30):
                cos, ids = idx.search(q, 5)
                assert np.array_equal(ids[:, 0], np.arange(16))      # rows 0..15 are their own best match
                assert np.all(cos[:, 0] > 0.999999)
        except Exception as e:      # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=adder), threading.Thread(target=searcher), threading.Thread(target=searcher)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert len(idx) == 10000
    allx = np.concatenate([base, extra])
    cos, ids = idx.search(q, 5)
    ref_cos, ref_ids = R.knn_search(allx, q, 5)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(allx), R.normalize_rows(q))
