"""N > 1 path on CPU: two gloo ranks drive the real ShardedSearcher (packing, all-gather,
merge call, global ids) with oracle-backed stand-ins for the two GPU calls.  CPU only."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import retrieval as R
from semantic_query_engine_amd.sharded import ShardedSearcher, packed_part_bytes, shard_rows

DIM = 64


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    ct = {np.float32: ctypes.c_float, np.int64: ctypes.c_int64}[dtype]
    return np.ctypeslib.as_array((ct * n).from_address(ptr)).reshape(shape)


class OracleShardIndex:
    """search_device over host pointers, answered by the oracle (test double)."""

    def __init__(self, x_shard):
        self.xn = R.normalize_rows(x_shard)
        self.id_base = 0

    def set_option(self, key, value):
        assert key == "id_base"
        self.id_base = int(value)

    def search_device(self, q_ptr, b, k, cos_ptr, id_ptr, nprobe=0):
        q = _view(q_ptr, (b, DIM), np.float32)
        cos, ids = R.exact_topk(self.xn, R.normalize_rows(q), k)
        _view(cos_ptr, (b, k), np.float32)[:] = cos.astype(np.float32)
        _view(id_ptr, (b, k), np.int64)[:] = np.where(ids >= 0, ids + self.id_base, -1)


class OracleCtx:
    device = 0

    def set_stream(self, s):
        pass

    def merge_topk_device(self, cos_ptr, id_ptr, stride, P, B, k, cos_out, id_out):
        out_c, out_i = _view(cos_out, (B, k), np.float32), _view(id_out, (B, k), np.int64)
        for q in range(B):
            ent = []
            for p in range(P):
                c = _view(cos_ptr + p * stride, (B, k), np.float32)[q]
                i = _view(id_ptr + p * stride, (B, k), np.int64)[q]
                ent += [(-float(cv), int(iv)) for cv, iv in zip(c, i) if iv >= 0]
            ent.sort()
            ent = ent[:k]
            out_c[q] = -np.inf
            out_i[q] = -1
            for j, (nc, iv) in enumerate(ent):
                out_c[q, j], out_i[q, j] = -nc, iv


def _data():
    rng = np.random.default_rng(21)
    x = rng.standard_normal((301, DIM)).astype(np.float32)
    q = rng.standard_normal((5, DIM)).astype(np.float32)
    q[0] = x[7] * 2.0
    x[250] = x[7]            # exact duplicate living in the other shard: lowest global id first
    x[151] = x[7]
    return x, q


def _worker(rank, world, port, k, out, force=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, q = _data()
    lo, hi = shard_rows(x.shape[0], world, rank)
    s = ShardedSearcher(OracleCtx(), OracleShardIndex(x[lo:hi]), id_base=lo, dist=dist, world=world,
                        device=torch.device("cpu"), force_collective=force)
    assert s.collective == (world > 1 or force)
    cos, ids = s.search(torch.from_numpy(q), k)
    s.synchronize()
    if rank == 0:
        out.put((cos.numpy().copy(), ids.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,k", [(2, 10), (2, 200), (3, 4)])
def test_sharded_search_matches_global_oracle(world, k):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, out)) for r in range(world)]
    for p in procs:
        p.start()
    cos, ids = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, q = _data()
    ref_cos, ref_ids = R.knn_search(x, q, k)
    assert np.array_equal(ids, ref_ids)
    valid = ref_ids >= 0
    assert np.allclose(cos[valid], ref_cos[valid], atol=1e-6) and np.all(np.isneginf(cos[~valid]))
    assert ids[0, :3].tolist() == [7, 151, 250]          # equal cosines across shards: lowest id first


def test_one_rank_forced_collective_path():
    """The rehearsal form used on a one-GPU box (bench.py --force-collective): a one-rank group still goes
    through pack -> all-gather -> merge and returns the global answer."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), 10, out, True))
    p.start()
    cos, ids = out.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    x, q = _data()
    ref_cos, ref_ids = R.knn_search(x, q, 10)
    assert np.array_equal(ids, ref_ids) and np.allclose(cos, ref_cos, atol=1e-6)


def test_shard_rows_and_message_layout():
    assert [shard_rows(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard_rows(2, 4, 3) == (2, 2)
    assert sum(hi - lo for lo, hi in (shard_rows(10_000_000, 8, r) for r in range(8))) == 10_000_000
    assert packed_part_bytes(1024, 10) == 122880 and packed_part_bytes(3, 1) % 16 == 0
