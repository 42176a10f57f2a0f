"""BERT encoder object over the C ABI (sqe_encoder_*): the model that ran inside Ollama for
``ollama_embed_text`` (main.py:134-145).  Weights are named as in the HF ``BertModel`` state dict."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping

import numpy as np

from . import _native as N
from .engine import Context

BERT_LARGE = dict(vocab_size=30522, hidden=1024, layers=24, heads=16, inter=4096, max_pos=512, type_vocab=2,
                  ln_eps=1e-12)


class BertEncoder:
    def __init__(self, ctx: Context, **cfg):
        self.ctx, self.lib = ctx, ctx.lib
        full = dict(BERT_LARGE)
        full.update(cfg)
        self.cfg = full
        c = N.BertCfg(full["vocab_size"], full["hidden"], full["layers"], full["heads"], full["inter"],
                      full["max_pos"], full["type_vocab"], full["ln_eps"])
        h = C.c_void_p()
        N.check(self.lib.sqe_encoder_create(ctx.handle, C.byref(c), C.byref(h)))
        self.handle = h
        ctx._children.add(self)

    def close(self) -> None:
        if getattr(self, "handle", None):
            if self.ctx.handle:
                self.lib.sqe_encoder_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, weights: Mapping[str, "np.ndarray"]) -> None:
        """weights: name -> fp32 array (NumPy or anything with ``.numpy()``); pooler entries are ignored."""
        for name, w in weights.items():
            if name.startswith("pooler.") or name.endswith("position_ids"):
                continue
            a = np.ascontiguousarray(w.numpy() if hasattr(w, "numpy") else w, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            N.check(self.lib.sqe_encoder_load_tensor(self.handle, name.encode(), a.ctypes.data, shape, a.ndim))
        N.check(self.lib.sqe_encoder_finalize(self.handle))

    def encode_ids(self, ids: np.ndarray, lens: np.ndarray) -> np.ndarray:
        """ids int32 [B,S] (anything past lens[b] is ignored), lens [B] -> CLS embeddings fp32 [B, hidden]."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        b, s = ids.shape
        out = np.empty((b, self.cfg["hidden"]), np.float32)
        if b:
            N.check(self.lib.sqe_encode(self.handle, ids.ctypes.data, lens.ctypes.data, b, s, out.ctypes.data))
        return out

    def encode_ids_device(self, ids_ptr: int, lens_ptr: int, b: int, s: int, out_ptr: int) -> None:
        N.check(self.lib.sqe_encode_device(self.handle, ids_ptr, lens_ptr, b, s, out_ptr))
