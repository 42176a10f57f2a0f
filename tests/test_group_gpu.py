"""One host process, several shards behind ONE context (sqe_create with n_dev > 1, group.hip): the form the
reference's single uvicorn process needs (main.py:738-739).  The test box has one GPU, so the group is
rehearsed with P LOGICAL shards on device 0 -- the same code path (round-robin placement g % P, query
broadcast, per-shard pipeline on its own stream, one exchange step, merge with local -> global ids), with
the peer-copy exchange; the in-library RCCL leg (dlopen'ed librccl, ncclCommInitAll + grouped
ncclAllGather) is run on a one-shard group.  Unmeasured at N > 1 devices (no multi-GPU box in this
pipeline): DESIGN.md section 5 says so.  GPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import retrieval as R
from tests.gpu_util import assert_topk_matches

pytestmark = pytest.mark.gpu


def _check(idx, x, q, k):
    cos, ids = idx.search(q, k)
    ref_cos, ref_ids = R.knn_search(x, q, k)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
    return cos, ids


@pytest.mark.parametrize("P", [2, 3, 8])
def test_logical_shards_on_one_device_match_global_oracle(P, tmp_path):
    from semantic_query_engine_amd import EXCHANGE_COPY, Context, VectorIndex
    ctx = Context(devices=[0] * P, exchange=EXCHANGE_COPY)
    info = ctx.group_info()
    assert info == {"shards": P, "exchange": "copy", "devices": [0] * P}
    rng = np.random.default_rng(100 + P)
    dim, k = 256, 10
    x = rng.standard_normal((5003, dim)).astype(np.float32)
    q = rng.standard_normal((70, dim)).astype(np.float32)
    q[0] = x[41] * 3.0
    x[4000], x[1999] = x[41], x[41]                     # equal cosines on different shards: lowest GLOBAL id first
    idx = VectorIndex(ctx, dim)
    assert len(idx) == 0
    cos0, ids0 = idx.search(q[:3], 4)                   # empty index: (-inf, -1) padding
    assert np.all(ids0 == -1) and np.all(np.isneginf(cos0))
    # ragged appends: every call starts at a different shard
    for lo, hi in ((0, 1), (1, 8), (8, 1500), (1500, 1501), (1501, 5003)):
        idx.add(x[lo:hi])
    assert len(idx) == 5003
    cos, ids = _check(idx, x, q, k)
    assert ids[0, :3].tolist() == [41, 1999, 4000]
    # stored rows come back by GLOBAL row id, normalised as the reference stores them (main.py:315-316)
    rows = np.array([0, 1, P, 41, 4000, 5002])
    assert np.allclose(idx.get_rows(rows), R.normalize_rows(x[rows]), atol=1e-6)
    # device entry points: memory of the leader device, context stream, no host hop
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(rng.standard_normal((777, dim)).astype(np.float32)).to(dev)
    qd = torch.from_numpy(q).to(dev)
    cd = torch.empty((70, k), dtype=torch.float32, device=dev)
    jd = torch.empty((70, k), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    idx.add_device(xd.data_ptr(), 777)
    idx.search_device(qd.data_ptr(), 70, k, cd.data_ptr(), jd.data_ptr())
    ctx.synchronize()
    x2 = np.concatenate([x, xd.cpu().numpy()])
    ref_cos, ref_ids = R.knn_search(x2, q, k)
    assert_topk_matches(cd.cpu().numpy(), jd.cpu().numpy(), ref_cos, ref_ids, R.normalize_rows(x2), R.normalize_rows(q))
    # overwrite rows that live on different shards
    upd = np.array([3, 4, 5, 5779])
    x2[upd] = q[10:14] * 2.0
    idx.update(upd, x2[upd])
    cos, ids = _check(idx, x2, q, k)
    assert ids[10:14, 0].tolist() == upd.tolist() and np.all(cos[10:14, 0] > 0.999999)
    # id_base of a sharded index is added after the local -> global map
    idx.set_option("id_base", 1000)
    assert np.array_equal(idx.search(q[:5], 3)[1], ids[:5, :3] + 1000)
    idx.set_option("id_base", 0)
    # persistence: rows are written in global order, so the file loads on a different number of shards
    path = os.path.join(tmp_path, "g.sqeidx")
    idx.save(path)
    one = Context(0)
    same = VectorIndex.load(one, path)
    assert len(same) == len(idx)
    c1, i1 = same.search(q, k)
    assert np.array_equal(i1, ids) and np.array_equal(c1, cos)       # bit-identical results after a load
    back = VectorIndex.load(ctx, path)
    c2, i2 = back.search(q, k)
    assert np.array_equal(i2, ids) and np.array_equal(c2, cos)


@pytest.mark.parametrize("P", [2, 3, 8])
def test_ivf_behind_the_group_matches_the_global_oracle(P, tmp_path):
    """SURVEY 8(e) as written: "IVF: replicate centroids, shard lists by row ownership -- same [B,k] exchange".
    sqe_index_train on a group trains once on the leader and replicates the centroids; every shard assigns and scans
    the rows it owns.  The answers equal oracle.retrieval.ivf_search run on the exported GLOBAL structure (centroids
    + the list of every row in global row order) at several nprobe, rows added before and after training and through
    the device entry points included; a file saved from P shards loads on one device with bit-identical results, and
    back on P shards."""
    from semantic_query_engine_amd import EXCHANGE_COPY, INDEX_IVF_FLAT, Context, VectorIndex
    ctx = Context(devices=[0] * P, exchange=EXCHANGE_COPY)
    rng = np.random.default_rng(300 + P)
    n, d, k, nlist = 24000, 128, 10, 128
    cen = rng.standard_normal((150, d)).astype(np.float32)
    x = (cen[rng.integers(0, 150, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    q = (x[rng.integers(0, n, 40)] + 0.2 * rng.standard_normal((40, d))).astype(np.float32)
    idx = VectorIndex(ctx, d, INDEX_IVF_FLAT, nlist)
    with pytest.raises(Exception):
        idx.search(q, k)                                  # not trained yet
    idx.add(x[:7001])                                     # rows added before training: assigned by train(), on every shard
    idx.train(x[rng.permutation(n)[:12000]], iters=8, seed=3)
    idx.add(x[7001:20000])
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x[20000:]).to(dev)
    torch.cuda.synchronize()
    idx.add_device(xd.data_ptr(), n - 20000)              # leader-device block, dealt to the shards over peer reads
    ctx.synchronize()
    assert len(idx) == n
    centroids, assign = idx.ivf_export(nlist)
    assert assign.shape == (n,) and assign.min() >= 0 and assign.max() < nlist
    xn, qn = R.normalize_rows(x), R.normalize_rows(q)
    best = xn.astype(np.float64) @ centroids.astype(np.float64).T
    assert np.all(best.max(1) - best[np.arange(n), assign] < 2e-6)      # every row sits in the list of its best centroid
    results = {}
    for nprobe in (1, 8, nlist):
        cos, ids = idx.search(q, k, nprobe=nprobe)
        ref_cos, ref_ids = R.ivf_search(xn, qn, centroids, assign, k, nprobe)
        assert_topk_matches(cos, ids, ref_cos, ref_ids, xn, qn)
        results[nprobe] = (cos, ids)
    exact_cos, exact_ids = R.exact_topk(xn, qn, k)
    assert R.recall_at_k(results[nlist][1], exact_ids) == 1.0            # nprobe = nlist is the exact search
    assert R.recall_at_k(results[8][1], exact_ids) >= 0.95
    # overwriting rows on different shards re-assigns them
    upd = np.array([5, 6, 7, 20001])
    x[upd] = q[:4] * 3.0
    idx.update(upd, x[upd])
    cos, ids = idx.search(q[:4], 1, nprobe=4)
    assert ids[:, 0].tolist() == upd.tolist() and np.all(np.abs(cos[:, 0] - 1.0) < 1e-5)
    centroids2, assign2 = idx.ivf_export(nlist)
    assert np.array_equal(centroids2, centroids)
    # persistence across shard counts: same file format as a single-device IVF index
    path = os.path.join(tmp_path, "givf.sqeidx")
    idx.save(path)
    cos8, ids8 = idx.search(q, k, nprobe=8)
    one = VectorIndex.load(Context(0), path)
    c1, i1 = one.search(q, k, nprobe=8)
    assert np.array_equal(i1, ids8) and np.array_equal(c1, cos8)
    cen1, asg1 = one.ivf_export(nlist)
    assert np.array_equal(cen1, centroids2) and np.array_equal(asg1, assign2)
    back = VectorIndex.load(ctx, path)
    c2, i2 = back.search(q, k, nprobe=8)
    assert np.array_equal(i2, ids8) and np.array_equal(c2, cos8)
    # a single-device IVF file loads on the group as well
    path1 = os.path.join(tmp_path, "one.sqeidx")
    one.save(path1)
    again = VectorIndex.load(ctx, path1)
    c3, i3 = again.search(q, k, nprobe=8)
    assert np.array_equal(i3, ids8) and np.array_equal(c3, cos8)


def test_group_ivf_client_is_the_reference_named_indexer():
    """`GpuSearchClient(devices=[...], kind=IVF)` (r02 verdict: "cannot exist today") behind OpenSearchIndexer."""
    from semantic_query_engine_amd import INDEX_IVF_FLAT, Context
    from semantic_query_engine_amd.retrieval import GpuSearchClient, OpenSearchIndexer
    rng = np.random.default_rng(6)
    x = rng.standard_normal((3000, 1024)).astype(np.float32)
    docs = [{"doc_id": f"PMC{i // 7}.txt", "text": f"chunk {i}"} for i in range(3000)]
    client = GpuSearchClient(Context(devices=[0, 0, 0]), kind=INDEX_IVF_FLAT, nlist=16)
    ix = OpenSearchIndexer(client, "i")
    client.index("i").vectors.train(x[:2000], iters=4, seed=1)
    ix.add_embeddings(x[:1500], docs[:1500])
    ix.add_embeddings(x[1500:], docs[1500:])
    for j in (0, 1499, 1500, 2999):
        hits = ix.search(x[j:j + 1] * 1.5, k=3)
        assert hits[0][0]["text"] == f"chunk {j}" and abs(hits[0][1] - 1.0) < 1e-5


def test_in_library_rccl_all_gather_leg():
    """A one-shard group with the RCCL exchange forced: ncclCommInitAll(1 device) + ncclGroupStart / AllGather /
    GroupEnd on the library's stream, then the merge -- the leg the 8-GPU form takes, on the one GPU there is."""
    from semantic_query_engine_amd import EXCHANGE_RCCL, Context, VectorIndex
    ctx = Context(devices=[0], exchange=EXCHANGE_RCCL)
    assert ctx.group_info() == {"shards": 1, "exchange": "rccl", "devices": [0]}
    rng = np.random.default_rng(9)
    x = rng.standard_normal((3000, 128)).astype(np.float32)
    q = rng.standard_normal((33, 128)).astype(np.float32)
    idx = VectorIndex(ctx, 128)
    idx.add(x)
    for _ in range(3):                                   # repeated steps reuse the gather buffers
        _check(idx, x, q, 10)
    with pytest.raises(Exception):
        Context(devices=[0, 0], exchange=EXCHANGE_RCCL)  # RCCL needs distinct devices


def test_group_client_is_a_drop_in_for_the_single_device_one():
    """`GpuSearchClient(devices=[...])` behind the reference-named OpenSearchIndexer: same hits, same _source."""
    from semantic_query_engine_amd import Context
    from semantic_query_engine_amd.retrieval import GpuSearchClient, OpenSearchIndexer
    rng = np.random.default_rng(4)
    x = rng.standard_normal((900, 1024)).astype(np.float32)
    docs = [{"doc_id": f"PMC{i // 7}.txt", "text": f"chunk {i}"} for i in range(900)]
    a = OpenSearchIndexer(GpuSearchClient(Context(0)), "i")
    b = OpenSearchIndexer(GpuSearchClient(Context(devices=[0, 0, 0, 0])), "i")
    for ix in (a, b):
        ix.add_embeddings(x[:500], docs[:500])
        ix.add_embeddings(x[500:], docs[500:])
    for j in (0, 499, 500, 899):
        ha, hb = a.search(x[j:j + 1] * 1.5, k=3), b.search(x[j:j + 1] * 1.5, k=3)
        assert [h[0]["text"] for h in ha] == [h[0]["text"] for h in hb] and ha[0][0]["text"] == f"chunk {j}"
        assert np.allclose([h[1] for h in ha], [h[1] for h in hb], atol=1e-6)
        assert np.allclose(ha[0][0]["embedding"], hb[0][0]["embedding"], atol=1e-7)


def test_objects_do_not_share_a_lock():
    """An index, the cache and a second index run from different threads at once; each has its own mutex and
    stream (internal.h), so none of them waits for another's host call, and every answer stays the oracle's."""
    import threading
    from semantic_query_engine_amd import Context, VectorIndex
    from semantic_query_engine_amd.retrieval import SemanticLfuCache
    ctx = Context(0)
    rng = np.random.default_rng(2)
    xa, xb = rng.standard_normal((20000, 256)).astype(np.float32), rng.standard_normal((3000, 128)).astype(np.float32)
    ia, ib = VectorIndex(ctx, 256), VectorIndex(ctx, 128)
    ia.add(xa)
    ib.add(xb[:1000])
    errors = []

    def search_a():
        try:
            for it in range(40):
                cos, ids = ia.search(xa[it:it + 8] * 2.0, 3)
                assert np.array_equal(ids[:, 0], np.arange(it, it + 8))
        except Exception as e:      # pragma: no cover
            errors.append(e)

    def add_b():
        try:
            for lo in range(1000, 3000, 100):
                ib.add(xb[lo:lo + 100])
                cos, ids = ib.search(xb[lo:lo + 1], 1)
                assert ids[0, 0] == lo
        except Exception as e:      # pragma: no cover
            errors.append(e)

    def cache():
        try:
            c = SemanticLfuCache(ctx, max_items=64, dim=256)
            for it in range(150):
                c.put(xa[it:it + 1], f"r{it}")
                assert c.get(xa[it:it + 1]) == f"r{it}"
        except Exception as e:      # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=f) for f in (search_a, search_a, add_b, cache)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert len(ib) == 3000
    _check(ib, xb, xb[:16] + 0.1, 5)


def test_two_indexes_of_one_group_searched_from_two_threads():
    """Two indexes of ONE multi-shard context searched at once: their per-index locks do not exclude each other, and the
    group's enqueue workers have one task slot each, so a fan-out holds the group's fan-out lock from its first post to its
    last wait (group.hip: Group::fan_mu; without it a second post could overwrite a task -- a shard never searched, a stale
    gather slot merged -- and leave a closure pointing into a finished call's stack frame)."""
    import threading
    from semantic_query_engine_amd import EXCHANGE_COPY, Context, VectorIndex
    ctx = Context(devices=[0, 0], exchange=EXCHANGE_COPY)
    rng = np.random.default_rng(31)
    xa, xb = rng.standard_normal((6001, 256)).astype(np.float32), rng.standard_normal((4003, 128)).astype(np.float32)
    ia, ib = VectorIndex(ctx, 256), VectorIndex(ctx, 128)
    ia.add(xa)
    ib.add(xb)
    errors = []

    def run(idx, x):
        try:
            for it in range(60):
                lo = (it * 37) % (len(x) - 16)
                cos, ids = idx.search(x[lo:lo + 16] * 1.5, 3)
                assert np.array_equal(ids[:, 0], np.arange(lo, lo + 16)), (it, ids[:, 0])
                assert np.all(cos[:, 0] > 0.999)
        except Exception as e:      # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=run, args=a) for a in ((ia, xa), (ib, xb), (ia, xa), (ib, xb))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    _check(ia, xa, xa[:8] + 0.05, 5)
    _check(ib, xb, xb[:8] + 0.05, 5)
