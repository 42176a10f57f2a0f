"""Wire layer of the Ollama / OpenSearch shim (semantic_query_engine_amd/shim.py) on CPU: the app is
driven with oracle-backed stand-ins for the GPU index and the embedder, with the byte streams
opensearch-py and the reference's own `ollama_embed_text` produce (NDJSON bulk bodies, gzip, k-NN query
bodies).  Where /root/reference is present the reference's `ollama_embed_text` is lifted from source
and run UNMODIFIED against a live uvicorn server."""
import ast
import asyncio
import gzip
import json
import os
import socket
import threading
import time

import numpy as np
import pytest
from fastapi.testclient import TestClient

from oracle import retrieval as R
from semantic_query_engine_amd import shim

DIM = 32
REF_MAIN = "/root/reference/app/main.py"


class OracleVectors:
    """VectorIndex stand-in answered by the oracle (normalise on add, exact top-k, ties -> lowest row)."""

    def __init__(self, dim):
        self.dim, self.xn = dim, np.zeros((0, dim), np.float32)

    def __len__(self):
        return self.xn.shape[0]

    def add(self, x):
        self.xn = np.concatenate([self.xn, R.normalize_rows(np.asarray(x, np.float32))], 0)

    def update(self, rows, x):
        self.xn[np.asarray(rows)] = R.normalize_rows(np.asarray(x, np.float32))

    def get_rows(self, rows):
        return self.xn[np.asarray(rows, dtype=np.int64)]

    def search(self, q, k, nprobe=0):
        cos, ids = R.exact_topk(self.xn, R.normalize_rows(np.asarray(q, np.float32)), k)
        return cos.astype(np.float32), ids


class OracleNamed:
    def __init__(self, dim):
        self.vectors, self.sources, self.row_of_id, self.lock = OracleVectors(dim), [], {}, threading.Lock()


class OracleClient:
    def __init__(self, dim):
        self.dim, self._ix = dim, {}

    def index(self, name):
        return self._ix.setdefault(name, OracleNamed(self.dim))

    def exists(self, name):
        return name in self._ix

    def count(self, index):
        return {"count": len(self.index(index).vectors)}


class HashEmbedder:
    """Deterministic text -> vector; records the batches it was called with."""

    def __init__(self, dim):
        self.dim, self.calls = dim, []

    def embed(self, texts):
        self.calls.append(list(texts))
        out = np.zeros((len(texts), self.dim), np.float32)
        for i, t in enumerate(texts):
            rng = np.random.default_rng(abs(hash(t)) % (2 ** 32))
            out[i] = rng.standard_normal(self.dim)
        return out


def _index_body(dim):
    return {"settings": {"index": {"knn": True}},
            "mappings": {"properties": {"doc_id": {"type": "keyword"}, "text": {"type": "text"},
                                        "embedding": {"type": "knn_vector", "dimension": dim,
                                                      "method": {"name": "hnsw", "engine": "nmslib", "space_type": "cosinesimil",
                                                                 "parameters": {"m": 64, "ef_construction": 500}}}}}}


def _bulk_body(index, ids, docs, embs):
    lines = []
    for _id, d, e in zip(ids, docs, embs):
        lines.append(json.dumps({"index": {"_index": index, "_id": _id}}))
        lines.append(json.dumps({"doc_id": d["doc_id"], "text": d["text"], "embedding": [float(x) for x in e]}))
    return ("\n".join(lines) + "\n").encode()


@pytest.fixture()
def app_client():
    oc, emb = OracleClient(DIM), HashEmbedder(DIM)
    return TestClient(shim.create_app(oc, emb, DIM)), oc, emb


def test_index_lifecycle_and_count(app_client):
    c, oc, _ = app_client
    assert c.get("/").json()["version"]["distribution"] == "opensearch"
    assert c.head("/medical-search-index").status_code == 404
    r = c.put("/medical-search-index", json=_index_body(DIM))
    assert r.status_code == 200 and r.json()["acknowledged"] is True
    assert c.head("/medical-search-index").status_code == 200
    assert c.put("/medical-search-index", json=_index_body(DIM)).status_code == 400      # already exists
    assert c.get("/medical-search-index/_count").json()["count"] == 0
    assert c.post("/medical-search-index/_count", json={}).json()["count"] == 0
    assert c.get("/nope/_count").status_code == 404
    assert c.put("/other", json=_index_body(DIM + 1)).status_code == 400                 # wrong dimension


def test_bulk_then_knn_search_matches_oracle(app_client):
    c, oc, _ = app_client
    rng = np.random.default_rng(0)
    n = 150                                                # > 64: the reference flushes every 64 (main.py:333)
    x = rng.standard_normal((n, DIM)).astype(np.float32)
    xn = x / (np.linalg.norm(x, axis=1, keepdims=True) + 1e-9)          # the reference normalises before sending
    docs = [{"doc_id": f"PMC{i // 3}.txt", "text": f"chunk {i}"} for i in range(n)]
    ids = [f"{d['doc_id']}_{i}" for i, d in enumerate(docs)]
    c.put("/idx", json=_index_body(DIM))
    for lo in range(0, n, 64):
        body = _bulk_body("idx", ids[lo:lo + 64], docs[lo:lo + 64], xn[lo:lo + 64])
        if lo == 64:                                       # http_compress=True (main.py:254): gzip request bodies
            r = c.post("/_bulk", content=gzip.compress(body), headers={"content-encoding": "gzip", "content-type": "application/json"})
        else:
            r = c.post("/_bulk", content=body, headers={"content-type": "application/x-ndjson"})
        j = r.json()
        assert r.status_code == 200 and j["errors"] is False
        assert [it["index"]["status"] for it in j["items"]] == [201] * min(64, n - lo)
        assert [it["index"]["_id"] for it in j["items"]] == ids[lo:lo + 64]
    assert c.get("/idx/_count").json()["count"] == n
    q = x[17] + 0.05 * rng.standard_normal(DIM).astype(np.float32)
    qn = q / (np.linalg.norm(q) + 1e-9)
    r = c.post("/idx/_search", json={"size": 3, "query": {"knn": {"embedding": {"vector": [float(v) for v in qn], "k": 3}}}})
    hits = r.json()["hits"]["hits"]
    cos, want = R.exact_topk(R.normalize_rows(xn), R.normalize_rows(qn[None]), 3)
    assert [h["_id"] for h in hits] == [ids[i] for i in want[0]]
    assert hits[0]["_source"]["doc_id"] == docs[17]["doc_id"] and hits[0]["_source"]["text"] == "chunk 17"
    assert len(hits[0]["_source"]["embedding"]) == DIM
    for h, cv in zip(hits, cos[0]):
        assert abs(h["_score"] - 1.0 / (2.0 - float(cv))) < 1e-6
    assert r.json()["hits"]["max_score"] == hits[0]["_score"]
    # same _id again overwrites (op "index"), "create" conflicts
    body = _bulk_body("idx", ids[:2], [{"doc_id": "X", "text": "new"}] * 2, xn[40:42])
    j = c.post("/_bulk", content=body).json()
    assert [it["index"]["status"] for it in j["items"]] == [200, 200] and c.get("/idx/_count").json()["count"] == n
    lines = body.decode().replace('"index"', '"create"', 1)
    j = c.post("/_bulk", content=lines.encode()).json()
    assert j["errors"] is True and j["items"][0]["create"]["status"] == 409


def test_bad_requests(app_client):
    c, _, _ = app_client
    c.put("/idx", json=_index_body(DIM))
    assert c.post("/idx/_search", json={"query": {"match_all": {}}}).status_code == 400
    assert c.post("/idx/_search", json={"size": 1, "query": {"knn": {"embedding": {"vector": [0.0] * 3, "k": 1}}}}).status_code == 400
    assert c.post("/nope/_search", json={}).status_code == 404
    j = c.post("/_bulk", content=b'{"index":{"_index":"idx","_id":"a"}}\n{"doc_id":"d","text":"t","embedding":[1,2]}\n').json()
    assert j["errors"] is True and j["items"][0]["index"]["status"] == 400
    assert c.post("/_bulk", content=b'{"index":{"_index":"idx","_id":"a"}}\n').status_code == 400
    assert c.post("/api/embeddings", json={"prompt": "x"}).status_code == 400             # model missing


def test_ollama_embeddings_and_microbatching(app_client):
    c, _, emb = app_client
    r = c.post("/api/embeddings", json={"model": "mxbai-embed-large:latest", "prompt": "heart failure", "stream": False})
    v = r.json()["embedding"]
    assert r.status_code == 200 and len(v) == DIM
    assert np.allclose(v, HashEmbedder(DIM).embed(["heart failure"])[0])
    assert c.post("/api/embeddings", json={"model": "m", "prompt": ""}).json() == {"embedding": []}

    # concurrent requests share encoder calls
    async def many():
        app = shim.create_app(OracleClient(DIM), emb2 := HashEmbedder(DIM), DIM)
        import httpx
        transport = httpx.ASGITransport(app=app)
        async with httpx.AsyncClient(transport=transport, base_url="http://shim") as ac:
            rs = await asyncio.gather(*[ac.post("/api/embeddings", json={"model": "m", "prompt": f"text {i}"}) for i in range(40)])
        return rs, emb2
    rs, emb2 = asyncio.run(many())
    assert all(r.status_code == 200 for r in rs)
    for i, r in enumerate(rs):
        assert np.allclose(r.json()["embedding"], HashEmbedder(DIM).embed([f"text {i}"])[0])
    assert sum(len(b) for b in emb2.calls) == 40 and len(emb2.calls) < 40 and max(len(b) for b in emb2.calls) <= 64


def test_concurrent_searches_share_one_batched_scan():
    """32 concurrent `_search` requests (each B = 1 on the wire, as main.py:499 sends them)
    reach the index as ONE batched call, and every request still gets exactly its own oracle answer."""
    class CountingVectors(OracleVectors):
        def __init__(self, dim):
            super().__init__(dim)
            self.search_batches = []

        def search(self, q, k, nprobe=0):
            self.search_batches.append(np.asarray(q).shape[0])
            return super().search(q, k, nprobe)

    rng = np.random.default_rng(3)
    n, nq = 400, 32
    x = rng.standard_normal((n, DIM)).astype(np.float32)
    oc = OracleClient(DIM)
    named = oc.index("idx")
    named.vectors = CountingVectors(DIM)
    named.vectors.add(x)
    named.sources = [{"doc_id": f"PMC{i // 4}.txt", "text": f"chunk {i}"} for i in range(n)]
    named.row_of_id = {f"PMC{i // 4}.txt_{i}": i for i in range(n)}
    qs = (x[rng.integers(0, n, nq)] + 0.1 * rng.standard_normal((nq, DIM))).astype(np.float32)
    ks = [3 if i % 2 else 10 for i in range(nq)]                       # mixed k in one batch

    async def many():
        app = shim.create_app(oc, None, DIM)
        # the product's window is 1 ms; parsing 32 bodies of 1024 floats on a loaded test box can take longer than
        # that, and what is tested here is the batching, not the host's speed
        app.state.search_batcher.max_wait = 0.5
        import httpx
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://shim") as ac:
            rs = await asyncio.gather(*[ac.post("/idx/_search", json={"size": ks[i], "query": {"knn": {"embedding": {
                "vector": [float(v) for v in qs[i]], "k": ks[i]}}}}) for i in range(nq)])
        return rs, app.state.search_batcher
    rs, sb = asyncio.run(many())
    assert all(r.status_code == 200 for r in rs)
    assert named.vectors.search_batches == [nq] and sb.batches == 1     # one device call for the 32 requests
    xn = R.normalize_rows(x)
    for i, r in enumerate(rs):
        hits = r.json()["hits"]["hits"]
        cos, want = R.exact_topk(xn, R.normalize_rows(qs[i:i + 1]), ks[i])
        assert [h["_id"] for h in hits] == [f"PMC{j // 4}.txt_{j}" for j in want[0]]
        assert all(abs(h["_score"] - 1.0 / (2.0 - float(c))) < 1e-6 for h, c in zip(hits, cos[0]))
        assert hits[0]["_source"]["text"] == f"chunk {want[0][0]}"


def _small_oracle_index(n=200, seed=5, vectors_cls=None):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, DIM)).astype(np.float32)
    oc = OracleClient(DIM)
    named = oc.index("idx")
    if vectors_cls is not None:
        named.vectors = vectors_cls(DIM)
    named.vectors.add(x)
    named.sources = [{"doc_id": f"PMC{i}.txt", "text": f"chunk {i}"} for i in range(n)]
    named.row_of_id = {f"PMC{i}.txt_{i}": i for i in range(n)}
    return oc, named, x


def test_malformed_search_requests_fail_alone():
    """r02 advisor finding: a request is validated before it joins a batch -- a nested vector, a wrong length, a
    non-finite value or k outside [1, 256] answers 400 by itself, and the good requests sent at the same moment
    still get their oracle answers from one batched call."""
    oc, named, x = _small_oracle_index()
    good = [float(v) for v in x[7]]
    bad_bodies = [
        {"size": 3, "query": {"knn": {"embedding": {"vector": [good, good], "k": 3}}}},           # (2, dim): nested
        {"size": 3, "query": {"knn": {"embedding": {"vector": [[v] for v in good], "k": 3}}}},    # (dim, 1): passed r02's shape check
        {"size": 3, "query": {"knn": {"embedding": {"vector": good[:-1], "k": 3}}}},
        {"size": 1000, "query": {"knn": {"embedding": {"vector": good, "k": 1000}}}},             # k > 256
        {"size": 3, "query": {"knn": {"embedding": {"vector": 1.5, "k": 3}}}},
    ]

    async def many():
        app = shim.create_app(oc, None, DIM)
        app.state.search_batcher.max_wait = 0.3
        import httpx
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://shim") as ac:
            reqs = [ac.post("/idx/_search", json=b) for b in bad_bodies]
            reqs += [ac.post("/idx/_search", json={"size": 3, "query": {"knn": {"embedding": {"vector": [float(v) for v in x[i]], "k": 3}}}})
                     for i in (7, 11, 13)]
            # NaN cannot travel as JSON through httpx's encoder: send the body by hand
            nan_body = json.dumps({"size": 3, "query": {"knn": {"embedding": {"vector": good, "k": 3}}}}).replace(repr(good[0]), "NaN", 1)
            reqs.append(ac.post("/idx/_search", content=nan_body, headers={"content-type": "application/json"}))
            return await asyncio.gather(*reqs), app.state.search_batcher
    rs, sb = asyncio.run(many())
    for r in rs[:len(bad_bodies)]:
        assert r.status_code == 400, r.text
    assert rs[-1].status_code == 400
    for r, i in zip(rs[len(bad_bodies):-1], (7, 11, 13)):
        assert r.status_code == 200
        assert r.json()["hits"]["hits"][0]["_id"] == f"PMC{i}.txt_{i}"
    assert sb.batches == 1 and sb.batch_sizes == [3]                     # the bad requests never reached the batcher


def test_device_failure_in_a_batch_fails_only_the_offending_request():
    """If a batched call does fail, the group's requests are re-run one by one: only the request that cannot be
    served answers 500 (r02: every co-batched client saw the failure)."""
    class Picky(OracleVectors):
        def search(self, q, k, nprobe=0):
            q = np.asarray(q)
            if np.any(np.abs(q[:, 0] - 12345.0) < 0.5):                  # the poisoned query fails whatever batch it is in
                raise RuntimeError("sqe_index_search: hip error")
            return super().search(q, k, nprobe)

    oc, named, x = _small_oracle_index(vectors_cls=Picky)
    poisoned = x[3].copy()
    poisoned[0] = 12345.0

    async def many():
        app = shim.create_app(oc, None, DIM)
        app.state.search_batcher.max_wait = 0.3
        import httpx
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://shim") as ac:
            vecs = [x[5], poisoned, x[9]]
            return await asyncio.gather(*[ac.post("/idx/_search", json={"size": 2, "query": {"knn": {"embedding": {
                "vector": [float(v) for v in q], "k": 2}}}}) for q in vecs]), app.state.search_batcher
    rs, sb = asyncio.run(many())
    assert [r.status_code for r in rs] == [200, 500, 200]
    assert rs[0].json()["hits"]["hits"][0]["_id"] == "PMC5.txt_5" and rs[2].json()["hits"]["hits"][0]["_id"] == "PMC9.txt_9"
    assert "hip error" in rs[1].json()["error"]["reason"]


def test_failed_device_add_keeps_docstore_and_vectors_in_step():
    """A bulk request whose device add fails must not leave documents behind: later adds still land at
    vector row == docstore row, and hits map to the right documents."""
    class FailingOnce(OracleVectors):
        fail_next = True

        def add(self, x):
            if self.fail_next:
                self.fail_next = False
                raise RuntimeError("hipMalloc: out of memory")
            super().add(x)

    oc = OracleClient(DIM)
    named = oc.index("idx")
    named.vectors = FailingOnce(DIM)
    c = TestClient(shim.create_app(oc, None, DIM))
    rng = np.random.default_rng(1)
    x = rng.standard_normal((6, DIM)).astype(np.float32)
    docs = [{"doc_id": f"D{i}", "text": f"t{i}"} for i in range(6)]
    ids = [f"D{i}_{i}" for i in range(6)]
    j = c.post("/_bulk", content=_bulk_body("idx", ids[:3], docs[:3], x[:3])).json()
    assert j["errors"] is True and [it["index"]["status"] for it in j["items"]] == [500] * 3
    assert named.sources == [] and named.row_of_id == {} and len(named.vectors) == 0
    # a malformed vector (a string inside the list) is a per-document 400, the rest of the request goes through
    body = _bulk_body("idx", ids[3:], docs[3:], x[3:]).decode().split("\n")
    bad = json.loads(body[1]); bad["embedding"][0] = "oops"; body[1] = json.dumps(bad)
    j = c.post("/_bulk", content="\n".join(body).encode()).json()
    assert [it["index"]["status"] for it in j["items"]] == [400, 201, 201]
    assert len(named.sources) == len(named.vectors) == 2
    r = c.post("/idx/_search", json={"size": 1, "query": {"knn": {"embedding": {"vector": [float(v) for v in x[5]], "k": 1}}}})
    assert r.json()["hits"]["hits"][0]["_id"] == ids[5] and r.json()["hits"]["hits"][0]["_source"]["text"] == "t5"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference source not present (GPU box)")
def test_reference_ollama_embed_text_runs_unmodified_against_the_shim():
    """main.py:134-145 lifted as source text (ast), executed against a live server: same call, same return type."""
    import httpx
    import uvicorn
    tree = ast.parse(open(REF_MAIN).read())
    fn = next(n for n in tree.body if isinstance(n, ast.AsyncFunctionDef) and n.name == "ollama_embed_text")
    port = _free_port()
    ns = {"httpx": httpx, "List": list, "EMBED_MODEL_NAME": "mxbai-embed-large:latest",
          "OLLAMA_API_URL": f"http://127.0.0.1:{port}/api"}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "<lifted>", "exec"), ns)
    emb = HashEmbedder(DIM)
    server = uvicorn.Server(uvicorn.Config(shim.create_app(OracleClient(DIM), emb, DIM), host="127.0.0.1", port=port, log_level="error"))
    t = threading.Thread(target=server.run, daemon=True)
    t.start()
    try:
        for _ in range(100):
            if server.started:
                break
            time.sleep(0.05)
        out = asyncio.run(ns["ollama_embed_text"]("chest pain differential"))
        assert isinstance(out, list) and len(out) == DIM and all(isinstance(v, float) for v in out)
        assert np.allclose(out, HashEmbedder(DIM).embed(["chest pain differential"])[0])
    finally:
        server.should_exit = True
        t.join(timeout=5)
