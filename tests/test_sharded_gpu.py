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


_RCCL_REHEARSAL = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SQE_ROOT"])
from oracle import retrieval as R
from semantic_query_engine_amd import Context, VectorIndex
from semantic_query_engine_amd.sharded import ShardedSearcher
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
rng = np.random.default_rng(11)
n, dim, b, k, base = 3000, 256, 70, 10, 5000
x = rng.standard_normal((n, dim)).astype(np.float32)
q = rng.standard_normal((b, dim)).astype(np.float32)
ctx = Context(0)
idx = VectorIndex(ctx, dim)
idx.add(x)
s = ShardedSearcher(ctx, idx, id_base=base, dist=dist, world=1, device=dev, force_collective=True)
assert s.collective
qd = torch.from_numpy(q).to(dev)
torch.cuda.synchronize()
for _ in range(3):                     # repeated steps reuse the packed buffers
    cos, ids = s.search(qd, k)
s.synchronize()
dist.barrier()
rc, ri = R.knn_search(x, q, k)
assert np.array_equal(ids.cpu().numpy(), ri + base), "ids differ after all-gather + merge"
assert np.abs(cos.cpu().numpy() - rc).max() < 1e-5
dist.destroy_process_group()
print("rccl rehearsal ok")
"""


def test_rccl_all_gather_path_one_rank(tmp_path):
    """The RCCL leg of the N > 1 search (packed [B,k] -> all_gather_into_tensor on the searcher's stream ->
    merge kernel) on the one GPU a test box has: a one-rank nccl group with the collective path forced.
    Runs in a child process so the process group never outlives the test."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "rehearsal.py"
    script.write_text(_RCCL_REHEARSAL)
    env = dict(os.environ, SQE_ROOT=root, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rccl rehearsal ok" in r.stdout
