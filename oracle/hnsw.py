"""ctypes front of oracle/hnsw.cpp: the CPU HNSW index the reference's OpenSearch mapping asks for
(app/main.py:272-276: nmslib hnsw, cosinesimil, m = 64, ef_construction = 500).  TEST / BASELINE
INFRASTRUCTURE: used by tests/ and by the cpu_baseline leg of bench.py only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libhnsw.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            subprocess.run(["make", "-C", _HERE, "_build/libhnsw.so"], check=True, capture_output=True)
        lib = C.CDLL(_LIB_PATH)
        lib.hnsw_build.restype = C.c_void_p
        lib.hnsw_build.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int]
        lib.hnsw_search.restype = None
        lib.hnsw_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.hnsw_max_level.restype = C.c_int
        lib.hnsw_max_level.argtypes = [C.c_void_p]
        lib.hnsw_free.restype = None
        lib.hnsw_free.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


class HnswIndex:
    """Cosine HNSW over rows normalised as the reference does before indexing (x / (||x|| + 1e-9), main.py:315-316)."""

    def __init__(self, x: np.ndarray, m: int = 64, ef_construction: int = 500, seed: int = 0, threads: int = 0):
        x = np.asarray(x, dtype=np.float32)
        self.xn = np.ascontiguousarray(x / (np.linalg.norm(x, axis=1, keepdims=True) + 1e-9), dtype=np.float32)
        self.lib = _load()
        self.handle = self.lib.hnsw_build(self.xn.ctypes.data, self.xn.shape[0], self.xn.shape[1], m, ef_construction, seed, threads)

    def search(self, q: np.ndarray, k: int, ef_search: int = 100, threads: int = 0):
        """-> (cos float32 [nq, k], ids int64 [nq, k]); ef_search = 100 is the k-NN plugin's default."""
        q = np.asarray(q, dtype=np.float32)
        qn = np.ascontiguousarray(q / (np.linalg.norm(q, axis=1, keepdims=True) + 1e-9), dtype=np.float32)
        ids = np.empty((qn.shape[0], k), np.int64)
        cos = np.empty((qn.shape[0], k), np.float32)
        self.lib.hnsw_search(self.handle, qn.ctypes.data, qn.shape[0], k, ef_search, threads, ids.ctypes.data, cos.ctypes.data)
        return cos, ids

    @property
    def max_level(self) -> int:
        return int(self.lib.hnsw_max_level(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.hnsw_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
