"""Row-sharded search: one process per GPU, per-shard top-k, ONE exchange step.

Every rank holds rows [id_base, id_base + len(index)) of the global index and the full
query batch.  A search is: local scan + rescore -> [B,k] (cosine, GLOBAL id) -> one
all-gather of the packed per-shard results over RCCL/xGMI (120 KB per rank at B=1024,
k=10: latency-bound, so a single collective) -> merge kernel (ties to the lowest global
id).  With world == 1 there is no collective.

On a GPU the library context is switched onto a dedicated torch stream so the scan, the
collective and the merge are ordered on the device without host synchronisation.  `search` orders that
stream after the caller's current stream (which produced `q`) and the caller's current stream after the
work, so the returned tensors can be used like any torch result; they are buffers REUSED by the next
`search` with the same (B, k).  `close()` hands the context back its own stream.
"""
from __future__ import annotations

import contextlib
from typing import Optional, Tuple

import torch


def packed_part_bytes(b: int, k: int) -> int:
    """Per-rank message: ids int64 [B,k] then cosines fp32 [B,k], padded to 16 bytes."""
    return (b * k * 12 + 15) // 16 * 16


def shard_rows(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of the global index owned by `rank` (contiguous, ceil-divided)."""
    per = (n_total + world - 1) // world
    return min(n_total, rank * per), min(n_total, (rank + 1) * per)


class ShardedSearcher:
    """`ctx` / `index` are the engine objects (Context / VectorIndex); anything exposing the same
    ``search_device`` / ``merge_topk_device`` / ``set_option`` / ``set_stream`` calls works,
    which is how the CPU/gloo protocol tests drive this class without a GPU."""

    def __init__(self, ctx, index, id_base: int = 0, dist=None, world: int = 1,
                 device: Optional[torch.device] = None, group=None, force_collective: bool = False):
        self.ctx, self.index, self.dist, self.world, self.group = ctx, index, dist, world, group
        # a one-rank group still takes the all-gather + merge path when asked to: that is how the RCCL
        # leg is rehearsed on a one-GPU box (tests/test_sharded_gpu.py, bench.py --force-collective)
        self.collective = dist is not None and (world > 1 or force_collective)
        self.device = device or torch.device("cuda", ctx.device)
        index.set_option("id_base", float(id_base))
        self._bufs = {}
        self.stream = None
        if self.device.type == "cuda":
            # a dedicated torch stream (the default stream's handle is 0 = "no stream" to the ABI)
            self.stream = torch.cuda.Stream(self.device)
            ctx.set_stream(self.stream.cuda_stream)

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _buffers(self, b: int, k: int):
        key = (b, k)
        if key not in self._bufs:
            part = packed_part_bytes(b, k)
            local = torch.empty(part, dtype=torch.uint8, device=self.device)
            gathered = torch.empty(part * self.world, dtype=torch.uint8, device=self.device) \
                if self.collective else local
            cos = torch.empty((b, k), dtype=torch.float32, device=self.device)
            ids = torch.empty((b, k), dtype=torch.int64, device=self.device)
            self._bufs[key] = (part, local, gathered, cos, ids)
        return self._bufs[key]

    def search(self, q: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """q: [B, dim] fp32 on this rank's device (same batch on every rank).
        Returns (cos [B,k] fp32, global ids [B,k] int64) on the device, asynchronously."""
        b = q.shape[0]
        part, local, gathered, cos, ids = self._buffers(b, k)
        id_ptr = local.data_ptr()
        cos_ptr = id_ptr + b * k * 8
        caller = None
        if self.stream is not None:
            caller = torch.cuda.current_stream(self.device)
            self.stream.wait_stream(caller)            # q (and the previous readers of cos / ids) are the caller's work
        with self._on_stream():
            if not self.collective:
                self.index.search_device(q.data_ptr(), b, k, cos.data_ptr(), ids.data_ptr())
            else:
                self.index.search_device(q.data_ptr(), b, k, cos_ptr, id_ptr)
                self.dist.all_gather_into_tensor(gathered, local, group=self.group)
                g = gathered.data_ptr()
                self.ctx.merge_topk_device(g + b * k * 8, g, part, self.world, b, k, cos.data_ptr(), ids.data_ptr())
        if caller is not None:
            caller.wait_stream(self.stream)            # results are ordered before whatever the caller enqueues next
        return cos, ids

    def synchronize(self) -> None:
        if self.stream is not None:
            self.stream.synchronize()

    def close(self) -> None:
        """Give the context its own stream back (the torch stream dies with this object)."""
        if self.stream is not None:
            self.stream.synchronize()
            try:
                self.ctx.set_stream(0)
            except Exception:
                pass
            self.stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
