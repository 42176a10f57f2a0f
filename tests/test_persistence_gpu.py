"""Index persistence (sqe_index_save / sqe_index_load).  GPU only.
A loaded index holds the same fp32 rows bit for bit, so every search returns identical ids AND
identical cosines; `has_any_data()` is true after a load, which is what lets the reference skip its
rebuild (main.py:422-424)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from semantic_query_engine_amd import Context
    return Context(0)


def test_flat_roundtrip_bit_identical(ctx, tmp_path):
    from semantic_query_engine_amd import VectorIndex
    rng = np.random.default_rng(5)
    x = rng.standard_normal((5000, 256)).astype(np.float32)
    x[17] = 0.0                                            # an all-zero row stays zero
    q = rng.standard_normal((33, 256)).astype(np.float32)
    a = VectorIndex(ctx, 256)
    a.add(x[:3000])
    a.add(x[3000:])
    p = str(tmp_path / "flat.sqeidx")
    a.save(p)
    assert os.path.getsize(p) == 64 + 5000 * 256 * 4
    b = VectorIndex.load(ctx, p)
    assert len(b) == 5000 and b.dim == 256
    rows = np.array([0, 17, 4999, 1234])
    assert np.array_equal(a.get_rows(rows), b.get_rows(rows))
    ca, ia = a.search(q, 10)
    cb, ib = b.search(q, 10)
    assert np.array_equal(ia, ib) and np.array_equal(ca, cb)
    # the loaded index keeps growing like any other
    b.add(x[:10])
    assert len(b) == 5010


def test_ivf_roundtrip(ctx, tmp_path):
    from semantic_query_engine_amd import INDEX_IVF_FLAT, VectorIndex
    rng = np.random.default_rng(6)
    centres = rng.standard_normal((32, 128)).astype(np.float32)
    x = (centres[rng.integers(0, 32, 6000)] + 0.3 * rng.standard_normal((6000, 128))).astype(np.float32)
    q = (centres[rng.integers(0, 32, 20)] + 0.3 * rng.standard_normal((20, 128))).astype(np.float32)
    a = VectorIndex(ctx, 128, INDEX_IVF_FLAT, 32)
    a.add(x)
    a.train(x[:4000], iters=8, seed=1)
    p = str(tmp_path / "ivf.sqeidx")
    a.save(p)
    b = VectorIndex.load(ctx, p)
    cen_a, asg_a = a.ivf_export(32)
    cen_b, asg_b = b.ivf_export(32)
    assert np.array_equal(cen_a, cen_b) and np.array_equal(asg_a, asg_b)
    ca, ia = a.search(q, 5, nprobe=4)
    cb, ib = b.search(q, 5, nprobe=4)
    assert np.array_equal(ia, ib) and np.array_equal(ca, cb)


def test_bad_files_are_refused(ctx, tmp_path):
    from semantic_query_engine_amd import VectorIndex
    from semantic_query_engine_amd._native import SqeError
    p = tmp_path / "junk.sqeidx"
    p.write_bytes(b"not an index" * 10)
    with pytest.raises(SqeError):
        VectorIndex.load(ctx, str(p))
    a = VectorIndex(ctx, 64)
    a.add(np.ones((300, 64), np.float32))
    good = tmp_path / "good.sqeidx"
    a.save(str(good))
    cut = tmp_path / "cut.sqeidx"
    cut.write_bytes(good.read_bytes()[:-100])
    with pytest.raises(SqeError, match="truncated"):
        VectorIndex.load(ctx, str(cut))


def test_indexer_skips_rebuild_after_load(ctx, tmp_path):
    """OpenSearchIndexer mirror: save, new client, load -> has_any_data() and the same hits."""
    from semantic_query_engine_amd.retrieval import GpuSearchClient, OpenSearchIndexer
    rng = np.random.default_rng(7)
    emb = rng.standard_normal((40, 1024)).astype(np.float32)
    docs = [{"doc_id": f"PMC{i // 4}.txt", "text": f"chunk {i}"} for i in range(40)]
    c1 = GpuSearchClient(ctx, dim=1024)
    ix1 = OpenSearchIndexer(c1, "medical-search-index")
    ix1.add_embeddings(emb, docs)
    hits1 = ix1.search(emb[7:8], k=3)
    c1.save_index("medical-search-index", str(tmp_path))
    c2 = GpuSearchClient(ctx, dim=1024)
    ix2 = OpenSearchIndexer(c2, "medical-search-index")
    assert not ix2.has_any_data()
    assert c2.load_index("medical-search-index", str(tmp_path))
    assert ix2.has_any_data()
    hits2 = ix2.search(emb[7:8], k=3)
    assert [(h[0]["doc_id"], h[0]["text"], h[1]) for h in hits1] == [(h[0]["doc_id"], h[0]["text"], h[1]) for h in hits2]
    assert hits2[0][0]["text"] == "chunk 7"
    # re-adding the same documents overwrites by _id instead of duplicating (main.py:325)
    ix2.add_embeddings(emb, docs)
    assert c2.count("medical-search-index")["count"] == 40
    assert not c2.load_index("no-such-index", str(tmp_path))
