"""BERT WordPiece tokenizer (host-only C++ in libsqe): ``WordPieceTokenizer(vocab_path)``."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _native as N


class WordPieceTokenizer:
    def __init__(self, vocab_path: str = None, vocab_text: str = None):
        self.lib = N.load()
        if vocab_text is None:
            with open(vocab_path, "rb") as f:
                data = f.read()
        else:
            data = vocab_text.encode("utf-8")
        h = C.c_void_p()
        N.check(self.lib.sqe_tokenizer_create(data, len(data), C.byref(h)))
        self.handle = h

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.sqe_tokenizer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, text: str, max_len: int = 512) -> List[int]:
        raw = text.encode("utf-8")
        ids = (C.c_int32 * max_len)()
        n = C.c_int32()
        N.check(self.lib.sqe_tokenize(self.handle, raw, len(raw), max_len, ids, C.byref(n)))
        return list(ids[: n.value])

    def encode_batch(self, texts: Sequence[str], max_len: int = 512) -> Tuple[np.ndarray, np.ndarray]:
        """-> (ids int32 [n, max_len] zero padded, lens int32 [n])."""
        n = len(texts)
        raws = [t.encode("utf-8") for t in texts]
        arr = (C.c_char_p * n)(*raws)
        sizes = np.array([len(r) for r in raws], dtype=np.int64)
        ids = np.zeros((n, max_len), dtype=np.int32)
        lens = np.zeros(n, dtype=np.int32)
        if n:
            N.check(self.lib.sqe_tokenize_batch(self.handle, arr, sizes.ctypes.data_as(N.c_i64_p), n, max_len,
                                                ids.ctypes.data, lens.ctypes.data))
        return ids, lens
