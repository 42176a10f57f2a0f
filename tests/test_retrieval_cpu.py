"""Host logic of the reference-named mirrors (retrieval.py = main.py forms, upload.py = embedding_gen.py forms)
with oracle-backed stand-ins for the GPU objects.  CPU only."""
import asyncio
import threading

import numpy as np
import pytest

from oracle import retrieval as R
from semantic_query_engine_amd import retrieval as RT
from semantic_query_engine_amd import upload as UP

DIM = 1024


class OracleVectors:
    def __init__(self, dim=DIM):
        self.dim, self.xn = dim, np.zeros((0, dim), np.float32)

    def __len__(self):
        return self.xn.shape[0]

    def add(self, x):
        x = np.asarray(x, np.float32)
        if x.ndim != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected [n, {self.dim}] array, got {x.shape}")
        self.xn = np.concatenate([self.xn, R.normalize_rows(x)], 0)

    def update(self, rows, x):
        self.xn[np.asarray(rows)] = R.normalize_rows(np.asarray(x, np.float32))

    def get_rows(self, rows):
        return self.xn[np.asarray(rows, dtype=np.int64)]

    def search(self, q, k, nprobe=0):
        cos, ids = R.exact_topk(self.xn, R.normalize_rows(np.asarray(q, np.float32)), k)
        return cos.astype(np.float32), ids


class Named:
    def __init__(self):
        self.vectors, self.sources, self.row_of_id, self.lock = OracleVectors(), [], {}, threading.Lock()


class Client:
    dim = DIM

    def __init__(self):
        self._ix = {}

    def index(self, name):
        return self._ix.setdefault(name, Named())

    def exists(self, name):
        return name in self._ix

    def count(self, index):
        return {"count": len(self.index(index).vectors)}


class FakeEmbedder:
    def __init__(self, fail_on=None, dim=DIM):
        self.fail_on, self.dim, self.calls = fail_on, dim, []

    def embed(self, texts):
        self.calls.append(list(texts))
        if self.fail_on is not None and any(self.fail_on in t for t in texts):
            raise RuntimeError("encoder failed")
        out = np.zeros((len(texts), self.dim), np.float32)
        for i, t in enumerate(texts):
            out[i] = np.random.default_rng(abs(hash(t)) % (2 ** 32)).standard_normal(self.dim)
        return out


def test_wrong_dim_batch_then_good_batch_keeps_hits_on_the_right_documents(capsys):
    """a failing device add must not leave the docstore ahead of the vector rows."""
    ix = RT.OpenSearchIndexer(Client(), "medical-search-index")
    rng = np.random.default_rng(0)
    bad = rng.standard_normal((3, 512)).astype(np.float32)
    ix.add_embeddings(bad, [{"doc_id": f"B{i}", "text": f"bad {i}"} for i in range(3)])
    assert "Bulk indexing error" in capsys.readouterr().out                 # swallowed with a print, as main.py:344-345
    named = ix.client.index("medical-search-index")
    assert named.sources == [] and named.row_of_id == {} and not ix.has_any_data()
    good = rng.standard_normal((5, DIM)).astype(np.float32)
    docs = [{"doc_id": f"G{i}", "text": f"good {i}"} for i in range(5)]
    ix.add_embeddings(good, docs)
    assert len(named.sources) == len(named.vectors) == 5
    for i in range(5):
        (src, score), = ix.search(good[i:i + 1] * 2.0, k=1)
        assert src["doc_id"] == f"G{i}" and src["text"] == f"good {i}" and abs(score - 1.0) < 1e-6
    # same _id again (global row index i, main.py:325) overwrites in place
    ix.add_embeddings(good[::-1].copy(), [{"doc_id": f"G{i}", "text": f"new {i}"} for i in range(5)])
    assert len(named.vectors) == 5
    (src, _), = ix.search(good[4:5], k=1)
    assert src["text"] == "new 0"


def test_search_groups_by_doc_id_like_ragmodel_ask():
    """main.py:500-507: hits are grouped by doc_id, texts joined with a newline, scores dropped."""
    ix = RT.OpenSearchIndexer(Client(), "i")
    rng = np.random.default_rng(1)
    base = rng.standard_normal((2, DIM)).astype(np.float32)
    embs = np.stack([base[0], base[0] + 0.01 * rng.standard_normal(DIM), base[1], base[0] + 0.02 * rng.standard_normal(DIM)]).astype(np.float32)
    docs = [{"doc_id": "PMC1.txt", "text": "a"}, {"doc_id": "PMC2.txt", "text": "b"}, {"doc_id": "PMC3.txt", "text": "c"},
            {"doc_id": "PMC1.txt", "text": "d"}]
    ix.add_embeddings(embs, docs)
    results = ix.search(base[0:1], k=3)
    doc_map = {}
    for doc_dict, _score in results:                                        # the loop of main.py:501-506
        doc_map.setdefault(doc_dict["doc_id"], []).append(doc_dict["text"])
    assert doc_map == {"PMC1.txt": ["a", "d"], "PMC2.txt": ["b"]}
    assert [len(r[0]["embedding"]) for r in results] == [DIM] * 3


def test_upload_forms_of_embedding_gen(capsys):
    RT.configure_embedder(emb := FakeEmbedder(fail_on="BOOM"))
    assert asyncio.run(UP.ollama_embed_text("   ")) == [0.0] * 1024          # embedding_gen.py:147-148
    assert emb.calls == []
    v = asyncio.run(UP.ollama_embed_text("heart failure"))
    assert len(v) == 1024 and v == emb.embed(["heart failure"])[0].tolist()
    assert asyncio.run(UP.ollama_embed_text("BOOM")) == [0.0] * 1024         # :164-166
    assert "[ERROR] Ollama embedding error" in capsys.readouterr().out
    e = asyncio.run(UP.embed_texts_in_batches([]))
    assert e.shape == (0, 1024) and e.dtype == np.float32                    # :173-174
    assert asyncio.run(RT.embed_texts_in_batches([])).shape == (0,)           # main.py:152-153 differs
    texts = [f"t{i}" for i in range(70)]
    texts[3], texts[66] = "", "BOOM"
    e = asyncio.run(UP.embed_texts_in_batches(texts))
    assert e.shape == (70, 1024) and e.dtype == np.float32
    assert not e[3].any() and np.array_equal(e[5], emb.embed(["t5"])[0])
    assert not e[64:].any()                                                  # the failing batch: zero rows
    # wrong model dimension: returned as is with the reference's warning
    RT.configure_embedder(FakeEmbedder(dim=8))
    assert len(asyncio.run(UP.ollama_embed_text("x"))) == 8
    assert "Mismatch embedding size. Expected 1024, got 8" in capsys.readouterr().out


def test_bulk_index_embeddings_per_user_index_and_per_document_ids(capsys):
    UP.configure_client(None)
    UP.bulk_index_embeddings("u1", "doc", np.ones((1, DIM), np.float32), ["c"])
    assert "cannot index" in capsys.readouterr().out
    client = Client()
    UP.configure_client(client)
    rng = np.random.default_rng(2)
    e1, e2 = rng.standard_normal((3, DIM)).astype(np.float32), rng.standard_normal((2, DIM)).astype(np.float32)
    e1[1] = 0.0                                                              # a failed embed: zero row, indexed as is
    UP.bulk_index_embeddings("u1", "notes_1", e1, ["a0", "a1", "a2"])
    UP.bulk_index_embeddings("u1", "labs_2", e2, ["b0", "b1", "extra chunk without embedding"])
    UP.bulk_index_embeddings("u2", "notes_1", e2, ["z0", "z1"])
    base = UP.BASE_OPENSEARCH_INDEX_NAME
    n1 = client.index(f"{base}-u1")
    assert sorted(n1.row_of_id) == ["labs_2_0", "labs_2_1", "notes_1_0", "notes_1_1", "notes_1_2"]   # :221: per-document i
    assert len(n1.vectors) == 5 and len(client.index(f"{base}-u2").vectors) == 2
    assert not n1.vectors.xn[1].any()                                        # zero row stays zero: no NaN
    # re-uploading the same doc_id overwrites its chunks (same _id), other documents untouched
    UP.bulk_index_embeddings("u1", "notes_1", e1[::-1].copy(), ["n0", "n1", "n2"])
    assert len(n1.vectors) == 5 and [s["text"] for s in n1.sources] == ["n0", "n1", "n2", "b0", "b1"]
    (src, _), = RT.OpenSearchIndexer(client, f"{base}-u1").search(e2[1:2], k=1)
    assert src == {"doc_id": "labs_2", "text": "b1", "embedding": src["embedding"]}
