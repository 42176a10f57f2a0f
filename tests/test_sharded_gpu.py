"""Merge kernel + sharded searcher on one GPU: P shards held as separate indexes."""
import numpy as np
import pytest
import torch

from oracle import retrieval as R

pytestmark = pytest.mark.gpu


def test_merge_of_row_shards_matches_global_oracle():
    from semantic_query_engine_amd import Context, VectorIndex
    from semantic_query_engine_amd.sharded import ShardedSearcher, packed_part_bytes, shard_rows
    rng = np.random.default_rng(5)
    n, dim, b, k, P = 5000, 128, 33, 10, 3
    x = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((b, dim)).astype(np.float32)
    q[0] = x[100]
    x[4000] = x[100]; x[2000] = x[100]                 # the same vector in all three shards
    ctx = Context(0)
    dev = torch.device("cuda", 0)
    part = packed_part_bytes(b, k)
    gathered = torch.zeros(part * P, dtype=torch.uint8, device=dev)
    qd = torch.from_numpy(q).to(dev)
    idxs = []
    for p in range(P):
        lo, hi = shard_rows(n, P, p)
        idx = VectorIndex(ctx, dim)
        idx.add(x[lo:hi])
        idx.set_option("id_base", lo)
        base = gathered.data_ptr() + p * part
        idx.search_device(qd.data_ptr(), b, k, base + b * k * 8, base)
        idxs.append(idx)
    cos = torch.empty((b, k), dtype=torch.float32, device=dev)
    ids = torch.empty((b, k), dtype=torch.int64, device=dev)
    g = gathered.data_ptr()
    ctx.merge_topk_device(g + b * k * 8, g, part, P, b, k, cos.data_ptr(), ids.data_ptr())
    ctx.synchronize()
    ref_cos, ref_ids = R.knn_search(x, q, k)
    assert np.array_equal(ids.cpu().numpy(), ref_ids)
    assert np.abs(cos.cpu().numpy() - ref_cos).max() < 1e-5
    assert ids[0, :3].tolist() == [100, 2000, 4000]
    # world == 1 searcher: plain local search on its own torch stream
    s = ShardedSearcher(ctx, idxs[0], id_base=0, world=1, device=dev)
    c1, i1 = s.search(qd, k)
    s.synchronize()
    lo, hi = shard_rows(n, P, 0)
    rc, ri = R.knn_search(x[lo:hi], q, k)
    assert np.array_equal(i1.cpu().numpy(), ri)
    ctx.set_stream(0)
