"""The HTTP shim over the real GPU index (libsqe through GpuSearchClient).  GPU only."""
import json

import numpy as np
import pytest
from fastapi.testclient import TestClient

from oracle import retrieval as R

pytestmark = pytest.mark.gpu


def test_bulk_and_search_over_libsqe():
    from semantic_query_engine_amd import Context, shim
    from semantic_query_engine_amd.retrieval import GpuSearchClient
    ctx = Context(0)
    dim = 1024
    client = GpuSearchClient(ctx, dim=dim)
    c = TestClient(shim.create_app(client, None, dim))
    rng = np.random.default_rng(3)
    n = 200
    x = rng.standard_normal((n, dim)).astype(np.float32)
    xn = x / (np.linalg.norm(x, axis=1, keepdims=True) + 1e-9)
    assert c.put("/medical-search-index", json={"mappings": {"properties": {"embedding": {
        "type": "knn_vector", "dimension": dim, "method": {"space_type": "cosinesimil"}}}}}).status_code == 200
    for lo in range(0, n, 64):
        lines = []
        for i in range(lo, min(n, lo + 64)):
            lines.append(json.dumps({"index": {"_index": "medical-search-index", "_id": f"PMC{i // 5}.txt_{i}"}}))
            lines.append(json.dumps({"doc_id": f"PMC{i // 5}.txt", "text": f"chunk {i}", "embedding": xn[i].tolist()}))
        j = c.post("/_bulk", content=("\n".join(lines) + "\n").encode()).json()
        assert j["errors"] is False
    assert c.get("/medical-search-index/_count").json()["count"] == n
    q = x[77] + 0.1 * rng.standard_normal(dim).astype(np.float32)
    qn = q / (np.linalg.norm(q) + 1e-9)
    r = c.post("/medical-search-index/_search", json={"size": 10, "query": {"knn": {"embedding": {"vector": qn.tolist(), "k": 10}}}})
    hits = r.json()["hits"]["hits"]
    cos, want = R.exact_topk(R.normalize_rows(xn), R.normalize_rows(qn[None]), 10)
    assert [h["_id"] for h in hits] == [f"PMC{i // 5}.txt_{i}" for i in want[0]]
    for h, cv in zip(hits, cos[0]):
        assert abs(h["_score"] - 1.0 / (2.0 - float(cv))) < 1e-3          # cosine within 1e-3 (north_star)
    assert np.allclose(hits[0]["_source"]["embedding"], xn[77], atol=1e-6)
    assert c.post("/api/embeddings", json={"model": "m", "prompt": "x"}).status_code == 500   # no model loaded
