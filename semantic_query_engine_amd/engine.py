"""Thin object layer over the C ABI (include/sqe.h): Context, VectorIndex, CacheMatrix.

NumPy in / NumPy out for host callers; raw device pointers (``tensor.data_ptr()``) for
callers that already hold their data in HBM.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
import threading
import weakref
from typing import Optional, Tuple

import numpy as np

from . import _native as N

INDEX_FLAT = 0
INDEX_IVF_FLAT = 1
SCAN_BF16_RESCORE = 0
SCAN_INT8_RESCORE = 2
# buffers of the int8 first pass (VectorIndex.i8_read; include/sqe.h: SQE_I8_*)
I8_ROWS, I8_ROW_SCALES, I8_QUERIES, I8_THRESHOLDS, I8_LIST_COUNTS, I8_LISTS, I8_SAMPLE_BEST, I8_POOL_COUNTS, I8_POOLS = range(9)


def _f32(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


EXCHANGE_AUTO, EXCHANGE_RCCL, EXCHANGE_COPY = 0, 1, 2


class Context:
    """One MI355X device (``device``: the local HIP ordinal -- the form used with one process per GPU), or
    ONE host process driving several (``devices=[0, 1, ...]``: flat indexes created on the context are
    sharded row-wise over them and searched with one exchange step, RCCL all-gather or peer copies;
    device pointers handed to the ``*_device`` calls are memory of ``devices[0]``).  Repeating a device id
    makes logical shards on one device (``exchange`` then has to be AUTO or COPY)."""

    def __init__(self, device: int = 0, devices=None, exchange: int = EXCHANGE_AUTO):
        self.lib = N.load()
        h = C.c_void_p()
        if devices is None:
            ids = (C.c_int32 * 1)(device)
            N.check(self.lib.sqe_create(ids, 1, C.byref(h)))
            self.devices = [device]
        else:
            self.devices = [int(d) for d in devices]
            ids = (C.c_int32 * len(self.devices))(*self.devices)
            N.check(self.lib.sqe_create_sharded(ids, len(self.devices), exchange, C.byref(h)))
        self.handle = h
        self.device = self.devices[0]
        self._children = weakref.WeakSet()      # indexes / caches that must die first

    def group_info(self) -> dict:
        """{"shards": P, "exchange": "rccl" | "copy", "devices": [...]} (one shard for a single-device context)."""
        n, ex = C.c_int32(), C.c_int32()
        devs = (C.c_int32 * 64)()
        N.check(self.lib.sqe_group_info(self.handle, C.byref(n), C.byref(ex), devs, 64))
        return {"shards": n.value, "exchange": {EXCHANGE_RCCL: "rccl", EXCHANGE_COPY: "copy"}.get(ex.value, "auto"),
                "devices": list(devs[:n.value])}

    def close(self) -> None:
        if getattr(self, "handle", None):
            for child in list(self._children):
                child.close()
            self.lib.sqe_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self) -> None:
        N.check(self.lib.sqe_synchronize(self.handle))

    @property
    def stream(self) -> int:
        return int(self.lib.sqe_stream(self.handle) or 0)

    def set_stream(self, hip_stream: int) -> None:
        """Enqueue on a caller-owned stream (e.g. ``torch.cuda.current_stream().cuda_stream``);
        0 restores the context's own stream."""
        N.check(self.lib.sqe_set_stream(self.handle, hip_stream or None))

    def device_info(self):
        name = C.create_string_buffer(128)
        cu = C.c_int32()
        mem = C.c_int64()
        N.check(self.lib.sqe_device_info(self.handle, name, 128, C.byref(cu), C.byref(mem)))
        return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}

    def set_profiling(self, on: bool) -> None:
        N.check(self.lib.sqe_set_profiling(self.handle, int(on)))

    def stats(self) -> dict:
        s = N.Stats()
        N.check(self.lib.sqe_stats(self.handle, C.byref(s)))
        return {f: getattr(s, f) for f, _ in N.Stats._fields_}

    def stats_reset(self) -> None:
        N.check(self.lib.sqe_stats_reset(self.handle))

    # -- cache scan, one-shot (main.py:73-87)
    def cosine_best(self, mat: np.ndarray, q: np.ndarray) -> Tuple[float, int]:
        q = _f32(q).reshape(-1)
        mat = _f32(mat).reshape(-1, q.shape[0]) if mat.size else np.zeros((0, q.shape[0]), np.float32)
        sim = C.c_float()
        idx = C.c_int32()
        N.check(self.lib.sqe_cosine_best(self.handle, mat.ctypes.data, mat.shape[0], q.shape[0],
                                         q.ctypes.data, C.byref(sim), C.byref(idx)))
        return float(sim.value), int(idx.value)

    def cosine_all(self, mat: np.ndarray, q: np.ndarray) -> np.ndarray:
        q = _f32(q).reshape(-1)
        mat = _f32(mat).reshape(-1, q.shape[0])
        out = np.empty(mat.shape[0], np.float32)
        N.check(self.lib.sqe_cosine_all(self.handle, mat.ctypes.data, mat.shape[0], q.shape[0],
                                        q.ctypes.data, out.ctypes.data))
        return out

    def merge_topk_device(self, cos_parts_ptr: int, id_parts_ptr: int, part_stride_bytes: int,
                          P: int, B: int, k: int, cos_out_ptr: int, id_out_ptr: int) -> None:
        N.check(self.lib.sqe_merge_topk_device(self.handle, cos_parts_ptr, id_parts_ptr, part_stride_bytes,
                                               P, B, k, cos_out_ptr, id_out_ptr))


class VectorIndex:
    """Cosine index over ``dim``-d vectors held in HBM (fp32 master + bf16 scanned copy)."""

    def __init__(self, ctx: Context, dim: int = 1024, kind: int = INDEX_FLAT, nlist: int = 0):
        self.ctx = ctx
        self.lib = ctx.lib
        self.dim = dim
        h = C.c_void_p()
        N.check(self.lib.sqe_index_create(ctx.handle, dim, kind, nlist, C.byref(h)))
        self.handle = h
        ctx._children.add(self)

    @classmethod
    def load(cls, ctx: Context, path: str) -> "VectorIndex":
        """Read an index written by ``save`` (sqe_index_load): same rows, bit-identical results."""
        import struct
        with open(path, "rb") as f:
            head = f.read(24)
        if len(head) < 24 or head[:8] != b"SQEIDX01":
            raise N.SqeError(-6, f"{path}: not a saved index")
        _version, dim, _kind, _nlist = struct.unpack_from("<IIII", head, 8)
        self = cls.__new__(cls)
        self.ctx, self.lib, self.dim = ctx, ctx.lib, int(dim)
        h = C.c_void_p()
        N.check(self.lib.sqe_index_load(ctx.handle, path.encode(), C.byref(h)))
        self.handle = h
        ctx._children.add(self)
        return self

    def save(self, path: str) -> None:
        """Write the stored (normalised) rows, and the IVF centroids/assignments if trained, to a local file."""
        N.check(self.lib.sqe_index_save(self.handle, path.encode()))

    def close(self) -> None:
        if getattr(self, "handle", None):
            if self.ctx.handle:                 # a destroyed context already released the device
                self.lib.sqe_index_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n = C.c_int64()
        N.check(self.lib.sqe_index_count(self.handle, C.byref(n)))
        return int(n.value)

    def reserve(self, rows: int) -> None:
        N.check(self.lib.sqe_index_reserve(self.handle, rows))

    def set_option(self, key: str, value: float) -> None:
        N.check(self.lib.sqe_index_set_option(self.handle, key.encode(), float(value)))

    def add(self, x: np.ndarray) -> None:
        x = _f32(x)
        if x.size == 0:
            return
        if x.ndim != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected [n, {self.dim}] array, got {x.shape}")
        N.check(self.lib.sqe_index_add(self.handle, x.ctypes.data, x.shape[0]))

    def add_device(self, ptr: int, n: int) -> None:
        N.check(self.lib.sqe_index_add_device(self.handle, ptr, n))

    def update(self, rows: np.ndarray, x: np.ndarray) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        x = _f32(x)
        if x.shape != (rows.shape[0], self.dim):
            raise ValueError(f"expected [{rows.shape[0]}, {self.dim}] array for {rows.shape[0]} rows, got {x.shape}")
        if rows.size:
            N.check(self.lib.sqe_index_update(self.handle, rows.ctypes.data, x.ctypes.data, rows.shape[0]))

    def get_rows(self, rows) -> np.ndarray:
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((rows.shape[0], self.dim), np.float32)
        if rows.size:
            N.check(self.lib.sqe_index_get_rows(self.handle, rows.ctypes.data, rows.shape[0], out.ctypes.data))
        return out

    # -- IVF-flat only
    def train(self, x: np.ndarray, iters: int = 20, seed: int = 0) -> None:
        """Spherical k-means on the sample ``x`` (host array), then (re)assignment of stored rows."""
        x = _f32(x)
        N.check(self.lib.sqe_index_train(self.handle, x.ctypes.data, x.shape[0], iters, seed))

    def train_device(self, ptr: int, n: int, iters: int = 20, seed: int = 0) -> None:
        N.check(self.lib.sqe_index_train_device(self.handle, ptr, n, iters, seed))

    def ivf_export(self, nlist: int) -> Tuple[np.ndarray, np.ndarray]:
        """-> (centroids float32 [nlist, dim], list id of every stored row int32 [count])."""
        cen = np.empty((nlist, self.dim), np.float32)
        asg = np.empty(len(self), np.int32)
        N.check(self.lib.sqe_index_ivf_export(self.handle, cen.ctypes.data, asg.ctypes.data))
        return cen, asg

    def i8_last(self) -> dict:
        """What the last search answered by the int8 first pass launched (sqe_index_i8_last)."""
        L = N.I8Launch()
        N.check(self.lib.sqe_index_i8_last(self.handle, L))
        return {name: getattr(L, name) for name, _ in N.I8Launch._fields_}

    def i8_read(self, what: int, dtype, count: int, offset_bytes: int = 0) -> np.ndarray:
        """`count` elements of `dtype` from one of the int8 pass's device buffers (sqe_index_i8_read; what = I8_*)."""
        out = np.empty(count, dtype)
        N.check(self.lib.sqe_index_i8_read(self.handle, what, offset_bytes, out.ctypes.data, out.nbytes))
        return out

    def search(self, q: np.ndarray, k: int, nprobe: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        """-> (cos [B,k] float32, ids [B,k] int64), best first, ties to the lowest id,
        (-inf, -1) padded."""
        q = _f32(q)
        if q.ndim == 1:
            q = q[None]
        if q.shape[1] != self.dim:
            raise ValueError(f"expected [B, {self.dim}] queries, got {q.shape}")
        b = q.shape[0]
        cos = np.empty((b, k), np.float32)
        ids = np.empty((b, k), np.int64)
        if b:
            N.check(self.lib.sqe_index_search(self.handle, q.ctypes.data, b, k, nprobe,
                                              cos.ctypes.data, ids.ctypes.data))
        return cos, ids

    def search_device(self, q_ptr: int, b: int, k: int, cos_ptr: int, id_ptr: int, nprobe: int = 0) -> None:
        """Asynchronous on the context stream; all pointers are device pointers."""
        N.check(self.lib.sqe_index_search_device(self.handle, q_ptr, b, k, nprobe, cos_ptr, id_ptr))


class CacheMatrix:
    """Resident cache matrix for the lfu_cache_get scan (main.py:73-87): slots hold raw
    embeddings; ``best(order, q)`` returns (sim, list position) of the first strict max."""

    def __init__(self, ctx: Context, capacity: int, dim: int = 1024):
        self.ctx = ctx
        self.lib = ctx.lib
        self.capacity, self.dim = capacity, dim
        h = C.c_void_p()
        N.check(self.lib.sqe_cache_create(ctx.handle, capacity, dim, C.byref(h)))
        self.handle = h
        self._lock = threading.Lock()
        ctx._children.add(self)

    def close(self) -> None:
        if getattr(self, "handle", None):
            if self.ctx.handle:
                self.lib.sqe_cache_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_slot(self, slot: int, vec: np.ndarray) -> None:
        vec = _f32(vec).reshape(-1)
        if vec.shape[0] != self.dim:            # the C side reads dim floats: never hand it a shorter buffer
            raise ValueError(f"cache embedding has {vec.shape[0]} values, the cache matrix holds {self.dim}-d rows")
        N.check(self.lib.sqe_cache_set_slot(self.handle, slot, vec.ctypes.data))

    def best(self, order, q: np.ndarray) -> Tuple[float, int]:
        order = np.ascontiguousarray(order, dtype=np.int32)
        q = _f32(q).reshape(-1)
        if q.shape[0] != self.dim:
            raise ValueError(f"query embedding has {q.shape[0]} values, the cache matrix holds {self.dim}-d rows")
        sim = C.c_float()
        pos = C.c_int32()
        N.check(self.lib.sqe_cache_best(self.handle, order.ctypes.data, order.shape[0], q.ctypes.data,
                                        C.byref(sim), C.byref(pos)))
        return float(sim.value), int(pos.value)
