"""Wire-compatible HTTP front for the GPU path: the subset of the Ollama and
OpenSearch REST APIs that the reference uses, so the UNMODIFIED ``app/main.py`` can be pointed at it
with environment variables only (``OLLAMA_API_URL=http://HOST:PORT/api``, ``OPENSEARCH_HOST=HOST``,
``OPENSEARCH_PORT=PORT``).

    reference call (app/main.py)                       endpoint here
    ------------------------------------------------   ------------------------------------------
    :141  POST {OLLAMA_API_URL}/embeddings              POST /api/embeddings      {"embedding": [...]}
    :258  os_client.info()                              GET  /
    :261  os_client.indices.exists(name)                HEAD /{index}
    :282  os_client.indices.create(index, body)         PUT  /{index}
    :304  client.count(index=...)                       GET|POST /{index}/_count  {"count": n}
    :342  helpers.bulk(client, actions)                 POST /_bulk (NDJSON, gzip accepted)
    :361  client.search(index, body={"size","query":{"knn":{"embedding":{"vector","k"}}}})
                                                        GET|POST /{index}/_search

Scores are the k-NN plugin's nmslib ``cosinesimil`` score ``1 / (2 - cos)``; ``_source`` carries
``doc_id``, ``text`` and the stored vector, as it did in OpenSearch.  Concurrent requests are
micro-batched because that is where the GPU path's throughput is: embedding requests into one encoder
call per <= 64 texts or 2 ms, ``_search`` requests into one batched scan per index per <= 64 queries or
1 ms (the reference issues every query as B = 1, main.py:499, :684; a B = 64 scan of 10 M rows costs what
a B = 1 scan costs -- both read the index once).

The app is built around two duck-typed objects so that the wire layer can be tested without a GPU:
``client`` (retrieval.GpuSearchClient: ``index(name)``, ``exists(name)``, ``count(index=)``) and
``embedder`` (retrieval.Embedder: ``embed(texts) -> float32 [n, dim]``).

    python -m semantic_query_engine_amd.shim --model /models/mxbai-embed-large-v1 --port 9200
"""
from __future__ import annotations

import asyncio
import gzip
import json
import time
from typing import Any, Dict, List, Optional

import numpy as np
from fastapi import FastAPI, Request, Response
from fastapi.responses import JSONResponse

_SHARDS = {"total": 1, "successful": 1, "skipped": 0, "failed": 0}
MAX_SEARCH_K = 256                                       # sqe_index_search: 1 <= k <= 256 (include/sqe.h)


class _EmbedBatcher:
    """Collects concurrent single-text requests into one ``embedder.embed`` call."""

    def __init__(self, embedder, max_batch: int = 64, max_wait_ms: float = 2.0):
        self.embedder, self.max_batch, self.max_wait = embedder, max_batch, max_wait_ms / 1e3
        self.queue: "asyncio.Queue" = asyncio.Queue()
        self.task: Optional[asyncio.Task] = None
        self.batches = 0

    async def embed(self, text: str) -> np.ndarray:
        if self.task is None or self.task.done():
            self.task = asyncio.get_running_loop().create_task(self._run())
        fut = asyncio.get_running_loop().create_future()
        await self.queue.put((text, fut))
        return await fut

    async def _run(self):
        loop = asyncio.get_running_loop()
        while True:
            items = [await self.queue.get()]
            deadline = loop.time() + self.max_wait
            while len(items) < self.max_batch:
                left = deadline - loop.time()
                if left <= 0:
                    break
                try:
                    items.append(await asyncio.wait_for(self.queue.get(), left))
                except asyncio.TimeoutError:
                    break
            texts = [t for t, _ in items]
            try:
                out = await loop.run_in_executor(None, self.embedder.embed, texts)
                self.batches += 1
                for (_, fut), row in zip(items, out):
                    if not fut.done():
                        fut.set_result(row)
            except Exception as e:                       # every waiter sees the failure
                for _, fut in items:
                    if not fut.done():
                        fut.set_exception(e)


class _SearchBatcher:
    """Collects concurrent k-NN requests into one batched scan per index."""

    def __init__(self, client, max_batch: int = 64, max_wait_ms: float = 1.0):
        self.client, self.max_batch, self.max_wait = client, max_batch, max_wait_ms / 1e3
        self.queue: "asyncio.Queue" = asyncio.Queue()
        self.task: Optional[asyncio.Task] = None
        self.batches = 0                                  # device calls made
        self.batch_sizes: List[int] = []

    async def search(self, index: str, vector: np.ndarray, k: int, field: str):
        if self.task is None or self.task.done():
            self.task = asyncio.get_running_loop().create_task(self._run())
        fut = asyncio.get_running_loop().create_future()
        await self.queue.put((index, vector, k, field, fut))
        return await fut

    async def _run(self):
        loop = asyncio.get_running_loop()
        while True:
            items = [await self.queue.get()]
            deadline = loop.time() + self.max_wait
            while len(items) < self.max_batch:
                left = deadline - loop.time()
                if left <= 0:
                    break
                try:
                    items.append(await asyncio.wait_for(self.queue.get(), left))
                except asyncio.TimeoutError:
                    break
            groups: Dict[str, List] = {}
            for it in items:
                groups.setdefault(it[0], []).append(it)
            for name, group in groups.items():
                try:
                    vectors = np.concatenate([g[1] for g in group], axis=0)
                    hits = await loop.run_in_executor(None, _search_hits_batch, self.client, name, vectors,
                                                      [g[2] for g in group], [g[3] for g in group])
                    self.batches += 1
                    self.batch_sizes.append(len(group))
                    for g, h in zip(group, hits):
                        if not g[4].done():
                            g[4].set_result(h)
                except Exception as e:
                    # A failing batch must not turn every co-batched client's search into a 500 (requests are
                    # validated before they are queued, so this is a device error or a request the validation
                    # missed): run the group's requests one by one, only the offending ones fail.
                    if len(group) == 1:
                        if not group[0][4].done():
                            group[0][4].set_exception(e)
                        continue
                    for g in group:
                        if g[4].done():
                            continue
                        try:
                            h = await loop.run_in_executor(None, _search_hits_batch, self.client, name, g[1], [g[2]], [g[3]])
                            self.batches += 1
                            self.batch_sizes.append(1)
                            g[4].set_result(h[0])
                        except Exception as e1:
                            g[4].set_exception(e1)


def _os_error(status: int, etype: str, reason: str, **extra) -> JSONResponse:
    err = {"type": etype, "reason": reason}
    err.update(extra)
    return JSONResponse({"error": {"root_cause": [err], **err}, "status": status}, status_code=status)


async def _body(request: Request) -> bytes:
    raw = await request.body()
    if request.headers.get("content-encoding", "").lower() == "gzip" and raw:
        raw = gzip.decompress(raw)                       # opensearch-py with http_compress=True (main.py:254)
    return raw


def create_app(client, embedder=None, embed_dim: int = 1024) -> FastAPI:
    app = FastAPI(title="semantic-query-engine GPU shim")
    batcher = _EmbedBatcher(embedder) if embedder is not None else None
    searcher = _SearchBatcher(client)
    mappings: Dict[str, Any] = {}
    app.state.batcher = batcher
    app.state.search_batcher = searcher

    # ------------------------------------------------------------------ Ollama
    @app.post("/api/embeddings")
    async def ollama_embeddings(request: Request):
        if batcher is None:
            return JSONResponse({"error": "no embedding model loaded"}, status_code=500)
        try:
            payload = json.loads(await _body(request) or b"{}")
        except ValueError:
            return JSONResponse({"error": "invalid JSON"}, status_code=400)
        if "model" not in payload:
            return JSONResponse({"error": "model is required"}, status_code=400)
        prompt = payload.get("prompt", "")
        if not isinstance(prompt, str) or prompt == "":
            return JSONResponse({"embedding": []})       # Ollama answers an empty prompt with an empty list
        vec = await batcher.embed(prompt)
        return JSONResponse({"embedding": [float(x) for x in vec]})

    # ------------------------------------------------------------------ OpenSearch
    @app.get("/")
    async def info():
        return {"name": "sqe-gpu", "cluster_name": "sqe", "cluster_uuid": "sqe",
                "version": {"distribution": "opensearch", "number": "2.11.0", "build_type": "sqe-shim",
                            "lucene_version": "n/a", "minimum_wire_compatibility_version": "7.10.0",
                            "minimum_index_compatibility_version": "7.0.0"},
                "tagline": "The OpenSearch Project: https://opensearch.org/"}

    @app.post("/_bulk")
    @app.put("/_bulk")
    async def bulk_root(request: Request):
        return await _bulk(request, None)

    @app.post("/{index}/_bulk")
    @app.put("/{index}/_bulk")
    async def bulk_index(index: str, request: Request):
        return await _bulk(request, index)

    async def _bulk(request: Request, default_index: Optional[str]):
        t0 = time.perf_counter()
        lines = [ln for ln in (await _body(request)).split(b"\n") if ln.strip()]
        # group consecutive documents of one index into one add_embeddings-style device call
        items: List[Dict[str, Any]] = []
        pending: Dict[str, List] = {}
        order: List[tuple] = []
        i = 0
        try:
            while i < len(lines):
                action = json.loads(lines[i])
                (op, meta), = action.items()
                i += 1
                if op == "delete":
                    items.append({"delete": {"_index": meta.get("_index", default_index), "_id": meta.get("_id"),
                                             "status": 400, "error": {"type": "illegal_argument_exception",
                                                                      "reason": "delete is not supported by this shim"}}})
                    continue
                if op not in ("index", "create", "update"):
                    return _os_error(400, "illegal_argument_exception", f"Malformed action/metadata line, unknown action [{op}]")
                if i >= len(lines):
                    return _os_error(400, "illegal_argument_exception", "The bulk request must be terminated by a newline [\\n]")
                src = json.loads(lines[i])
                i += 1
                name = meta.get("_index", default_index)
                slot = len(items)
                items.append(None)
                pending.setdefault(name, []).append((slot, op, meta.get("_id"), src))
                order.append(name)
        except ValueError as e:
            return _os_error(400, "parse_exception", f"malformed bulk body: {e}")
        errors = False
        for name, docs in pending.items():
            res = await asyncio.get_running_loop().run_in_executor(None, _index_docs, client, name, docs, embed_dim)
            for (slot, op, _id, _src), r in zip(docs, res):
                items[slot] = {op: r}
                errors = errors or r["status"] >= 300
        return {"took": int((time.perf_counter() - t0) * 1e3), "errors": errors, "items": items}

    @app.head("/{index}")
    async def index_exists(index: str):
        return Response(status_code=200 if client.exists(index) else 404)

    @app.put("/{index}")
    async def index_create(index: str, request: Request):
        if client.exists(index):
            return _os_error(400, "resource_already_exists_exception", f"index [{index}] already exists", index=index)
        raw = await _body(request)
        body = json.loads(raw) if raw.strip() else {}
        for field, spec in body.get("mappings", {}).get("properties", {}).items():
            if spec.get("type") == "knn_vector":
                dim = int(spec.get("dimension", embed_dim))
                space = spec.get("method", {}).get("space_type", "cosinesimil")
                if dim != client.dim:
                    return _os_error(400, "mapper_parsing_exception", f"knn_vector dimension {dim} != {client.dim} of this server")
                if space != "cosinesimil":
                    return _os_error(400, "mapper_parsing_exception", f"space_type [{space}] is not served; only cosinesimil")
        mappings[index] = body
        client.index(index)
        return {"acknowledged": True, "shards_acknowledged": True, "index": index}

    @app.get("/{index}/_count")
    @app.post("/{index}/_count")
    async def count(index: str):
        if not client.exists(index):
            return _os_error(404, "index_not_found_exception", f"no such index [{index}]", index=index)
        return {"count": client.count(index=index)["count"], "_shards": _SHARDS}

    @app.get("/{index}/_search")
    @app.post("/{index}/_search")
    async def search(index: str, request: Request):
        t0 = time.perf_counter()
        if not client.exists(index):
            return _os_error(404, "index_not_found_exception", f"no such index [{index}]", index=index)
        raw = await _body(request)
        try:
            body = json.loads(raw) if raw.strip() else {}
            knn = body["query"]["knn"]
            (field, spec), = knn.items()
            vector = np.asarray(spec["vector"], dtype=np.float32)
            k = int(body.get("size", spec.get("k", 10)))
            k = max(1, min(k, int(spec.get("k", k)))) if "k" in spec else k
        except (KeyError, ValueError, TypeError) as e:
            return _os_error(400, "parsing_exception", f"only {{'query': {{'knn': {{field: {{'vector', 'k'}}}}}}}} is served: {e}")
        # every request is validated BEFORE it joins a batch: one malformed request must fail alone
        if vector.ndim != 1 or vector.shape[0] != client.dim:
            got = "x".join(str(d) for d in vector.shape) or "a scalar"
            return _os_error(400, "illegal_argument_exception", f"query vector must be a flat list of {client.dim} numbers, got {got}")
        if not np.all(np.isfinite(vector)):
            return _os_error(400, "illegal_argument_exception", "query vector holds a NaN or an infinity")
        if not 1 <= k <= MAX_SEARCH_K:
            return _os_error(400, "illegal_argument_exception", f"size / k must be in [1, {MAX_SEARCH_K}] (sqe_index_search), got {k}")
        vector = vector[None, :]
        try:
            hits = await searcher.search(index, vector, k, field)
        except Exception as e:
            return _os_error(500, "sqe_device_exception", str(e))
        total = client.count(index=index)["count"]
        return {"took": int((time.perf_counter() - t0) * 1e3), "timed_out": False, "_shards": _SHARDS,
                "hits": {"total": {"value": min(total, len(hits)), "relation": "eq"},
                         "max_score": hits[0]["_score"] if hits else None, "hits": hits}}

    return app


def _index_docs(client, name: str, docs, embed_dim: int):
    """docs: [(slot, op, _id, _source)] of one index -> per-document bulk item bodies, in order.

    Vectors are validated and converted first, the device add / update runs next, and the host docstore
    (``sources`` / ``row_of_id``) is committed only after it succeeded: a failing device call reports every
    planned document as failed (500) and leaves docstore and vector rows in step."""
    idx = client.index(name)
    out: List[Optional[Dict[str, Any]]] = [None] * len(docs)
    shards = {"total": 1, "successful": 1, "failed": 0}
    with idx.lock:
        base_rows = len(idx.sources)
        new_vecs: List[np.ndarray] = []
        new_recs: List[Dict[str, Any]] = []
        new_ids: Dict[str, int] = {}                          # _id -> position among this call's inserts
        upd: Dict[int, tuple] = {}                            # stored row -> (record, vector): last writer wins
        planned: List[tuple] = []                             # (position in docs, item body on success)
        for pos, (_slot, op, _id, src) in enumerate(docs):
            base = {"_index": name, "_id": _id, "_shards": shards, "_primary_term": 1}
            emb = src.get("embedding") if isinstance(src, dict) else None
            vec = None
            if _id is not None and isinstance(emb, list) and len(emb) == client.dim:
                try:
                    vec = np.asarray(emb, dtype=np.float32)  # None / strings inside the list fail here, per document
                except (TypeError, ValueError):
                    vec = None
            if vec is None or vec.shape != (client.dim,):
                out[pos] = {**base, "status": 400, "error": {"type": "mapper_parsing_exception",
                                                             "reason": f"_id and an 'embedding' of {client.dim} floats are required"}}
                continue
            rec = {"doc_id": src.get("doc_id"), "text": src.get("text")}
            row = idx.row_of_id.get(_id)
            if row is None and _id not in new_ids:
                new_ids[_id] = len(new_vecs)
                new_vecs.append(vec)
                new_recs.append(rec)
                planned.append((pos, {**base, "_version": 1, "result": "created", "_seq_no": base_rows + len(new_vecs) - 1, "status": 201}))
            elif op == "create":
                out[pos] = {**base, "status": 409, "error": {"type": "version_conflict_engine_exception",
                                                             "reason": f"[{_id}]: version conflict, document already exists"}}
            elif row is None:                                 # second write to an _id inserted earlier in this request
                p0 = new_ids[_id]
                new_vecs[p0], new_recs[p0] = vec, rec
                planned.append((pos, {**base, "_version": 2, "result": "updated", "_seq_no": base_rows + p0, "status": 200}))
            else:
                upd[row] = (rec, vec)
                planned.append((pos, {**base, "_version": 2, "result": "updated", "_seq_no": row, "status": 200}))
        try:
            if new_vecs:
                idx.vectors.add(np.stack(new_vecs))
        except Exception as e:                                # nothing was appended: docstore untouched
            for pos, body in planned:
                out[pos] = {k: v for k, v in body.items() if k in ("_index", "_id")}
                out[pos].update({"status": 500, "error": {"type": "sqe_device_exception", "reason": str(e)}})
            return out
        for _id, p0 in new_ids.items():                       # the vector rows exist: so do their documents
            idx.row_of_id[_id] = base_rows + p0
        idx.sources.extend(new_recs)
        upd_failed = None
        try:
            if upd:
                rows = sorted(upd)
                idx.vectors.update(np.asarray(rows, np.int64), np.stack([upd[r][1] for r in rows]))
                for r in rows:
                    idx.sources[r] = upd[r][0]
        except Exception as e:
            upd_failed = str(e)
        for pos, body in planned:
            if upd_failed is not None and body["result"] == "updated" and body["_seq_no"] < base_rows:
                out[pos] = {"_index": name, "_id": body["_id"], "status": 500,
                            "error": {"type": "sqe_device_exception", "reason": upd_failed}}
            else:
                out[pos] = body
    return out


def _search_hits_batch(client, name: str, vectors: np.ndarray, ks: List[int], fields: List[str]):
    """One batched scan for the concurrent requests of one index: row b of ``vectors`` is request b's query
    (the reference sends row 0 only, main.py:355).  Exact cosine order, ``_score = 1 / (2 - cos)`` (what
    OpenSearchIndexer.search returns, with _id); request b gets its own first ``ks[b]`` hits."""
    idx = client.index(name)
    kmax = max(ks)
    with idx.lock:
        cos, ids = idx.vectors.search(np.ascontiguousarray(vectors, dtype=np.float32), kmax)
        rows_of = [[int(r) for r in ids[b][:ks[b]] if r >= 0] for b in range(len(ks))]
        flat = [r for rows in rows_of for r in rows]
        embs = idx.vectors.get_rows(flat) if flat else np.zeros((0, client.dim), np.float32)
        rev = getattr(idx, "_id_of_row", None)
        if rev is None or len(rev) != len(idx.row_of_id):
            rev = {row: os_id for os_id, row in idx.row_of_id.items()}
            idx._id_of_row = rev
        out, at = [], 0
        for b, rows in enumerate(rows_of):
            hits = []
            for j, row in enumerate(rows):
                src = idx.sources[row]
                hits.append({"_index": name, "_id": rev.get(row), "_score": float(1.0 / (2.0 - float(cos[b, j]))),
                             "_source": {"doc_id": src["doc_id"], "text": src["text"], fields[b]: [float(x) for x in embs[at + j]]}})
            at += len(rows)
            out.append(hits)
    return out


def _search_hits(client, name: str, vector: np.ndarray, k: int, field: str):
    """The un-batched form (one request): row 0 only."""
    return _search_hits_batch(client, name, vector[0:1], [k], [field])[0]


def main(argv=None) -> None:
    import argparse

    import uvicorn

    from .retrieval import GpuSearchClient, default_context
    from .weights import embedder_from_local

    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--model", help="local Hugging Face directory or GGUF file of the embedding model (never a model name)")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=9200)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--load", help="directory with indexes written by GpuSearchClient.save_index")
    args = ap.parse_args(argv)
    ctx = default_context()
    client = GpuSearchClient(ctx, dim=args.dim)
    if args.load:
        import glob
        import os
        for p in glob.glob(os.path.join(args.load, "*.sqeidx")):
            client.load_index(os.path.basename(p)[:-len(".sqeidx")], args.load)
    embedder = embedder_from_local(ctx, args.model) if args.model else None
    uvicorn.run(create_app(client, embedder, args.dim), host=args.host, port=args.port, log_level="warning")


if __name__ == "__main__":
    main()
