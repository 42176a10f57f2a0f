"""Local model files -> encoder weights + WordPiece vocabulary.

The reference names a model, ``mxbai-embed-large:latest`` (main.py:29), and lets Ollama fetch and run it.
Here a model is always a LOCAL PATH, never a name, and nothing is downloaded:

* a Hugging Face directory: ``config.json`` + ``model.safetensors`` + ``vocab.txt`` (or ``tokenizer.json``);
* a GGUF file as Ollama stores it under ``~/.ollama/models/blobs/`` (llama.cpp BERT architecture,
  F32 / F16 / BF16 tensors, vocabulary in ``tokenizer.ggml.tokens``).

Both readers are small parsers of the published container formats (no torch, no network): safetensors =
u64 header length + JSON header + raw little-endian data; GGUF v2/v3 = typed key/value metadata + tensor
table + aligned data.  Tensor names are mapped onto the ``BertModel`` state-dict names the encoder loads
(encoder.py).  No real checkpoint exists in this environment: the readers are tested on files written by
the tests themselves (tests/test_weights.py).
"""
from __future__ import annotations

import json
import mmap
import os
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

# ------------------------------------------------------------------ dtype helpers


def _bf16_to_f32(raw: np.ndarray) -> np.ndarray:
    return (raw.astype(np.uint32) << 16).view(np.float32)


def _as_f32(buf, dtype: str, count: int, offset: int) -> np.ndarray:
    if dtype in ("F32", "f32"):
        return np.frombuffer(buf, dtype="<f4", count=count, offset=offset).astype(np.float32)
    if dtype in ("F16", "f16"):
        return np.frombuffer(buf, dtype="<f2", count=count, offset=offset).astype(np.float32)
    if dtype in ("BF16", "bf16"):
        return _bf16_to_f32(np.frombuffer(buf, dtype="<u2", count=count, offset=offset))
    if dtype in ("F64", "f64"):
        return np.frombuffer(buf, dtype="<f8", count=count, offset=offset).astype(np.float32)
    raise ValueError(f"unsupported tensor dtype {dtype}")


# ------------------------------------------------------------------ safetensors


def read_safetensors(path: str) -> Dict[str, np.ndarray]:
    """name -> float32 array (F32 / F16 / BF16 / F64 tensors; integer tensors such as position_ids are skipped)."""
    out: Dict[str, np.ndarray] = {}
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            (hlen,) = struct.unpack_from("<Q", mm, 0)
            if hlen > len(mm) - 8:
                raise ValueError("safetensors: header length exceeds file size")
            header = json.loads(bytes(mm[8:8 + hlen]).decode("utf-8"))
            base = 8 + hlen
            for name, info in header.items():
                if name == "__metadata__":
                    continue
                dt = info["dtype"]
                if dt not in ("F32", "F16", "BF16", "F64"):
                    continue
                lo, hi = info["data_offsets"]
                shape = tuple(int(x) for x in info["shape"])
                count = int(np.prod(shape)) if shape else 1
                if base + hi > len(mm):
                    raise ValueError(f"safetensors: tensor {name} runs past the end of the file")
                out[name] = _as_f32(mm, dt, count, base + lo).reshape(shape)
        finally:
            mm.close()
    return out


# ------------------------------------------------------------------ GGUF

_GGUF_SCALARS = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<?", 10: "<Q", 11: "<q", 12: "<d"}
_GGML_TYPES = {0: ("F32", 4), 1: ("F16", 2), 30: ("BF16", 2)}


class _Reader:
    def __init__(self, buf):
        self.buf, self.pos = buf, 0

    def scalar(self, fmt):
        (v,) = struct.unpack_from(fmt, self.buf, self.pos)
        self.pos += struct.calcsize(fmt)
        return v

    def string(self) -> str:
        n = self.scalar("<Q")
        s = bytes(self.buf[self.pos:self.pos + n]).decode("utf-8", errors="replace")
        self.pos += n
        return s

    def value(self, t):
        if t in _GGUF_SCALARS:
            return self.scalar(_GGUF_SCALARS[t])
        if t == 8:
            return self.string()
        if t == 9:
            et = self.scalar("<I")
            n = self.scalar("<Q")
            return [self.value(et) for _ in range(n)]
        raise ValueError(f"gguf: unknown metadata type {t}")


def read_gguf(path: str) -> Tuple[Dict[str, object], Dict[str, np.ndarray]]:
    """-> (metadata dict, tensors name -> float32 array in row-major [outer, ..., inner] order)."""
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            r = _Reader(mm)
            if bytes(mm[0:4]) != b"GGUF":
                raise ValueError("not a GGUF file")
            r.pos = 4
            version = r.scalar("<I")
            if version not in (2, 3):
                raise ValueError(f"gguf: unsupported version {version}")
            n_tensors = r.scalar("<Q")
            n_kv = r.scalar("<Q")
            meta: Dict[str, object] = {}
            for _ in range(n_kv):
                key = r.string()
                t = r.scalar("<I")
                meta[key] = r.value(t)
            infos = []
            for _ in range(n_tensors):
                name = r.string()
                nd = r.scalar("<I")
                dims = [r.scalar("<Q") for _ in range(nd)]
                ttype = r.scalar("<I")
                off = r.scalar("<Q")
                infos.append((name, dims, ttype, off))
            align = int(meta.get("general.alignment", 32))
            data0 = (r.pos + align - 1) // align * align
            tensors: Dict[str, np.ndarray] = {}
            for name, dims, ttype, off in infos:
                if ttype not in _GGML_TYPES:
                    raise ValueError(f"gguf: tensor {name} has quantised/unsupported type {ttype}; "
                                     "only F32/F16/BF16 checkpoints are loaded")
                dt, _ = _GGML_TYPES[ttype]
                count = int(np.prod(dims)) if dims else 1
                shape = tuple(int(d) for d in reversed(dims))        # ggml lists the innermost dimension first
                tensors[name] = _as_f32(mm, dt, count, data0 + off).reshape(shape)
            return meta, tensors
        finally:
            mm.close()


_GGUF_LAYER = {
    "attn_q": "attention.self.query", "attn_k": "attention.self.key", "attn_v": "attention.self.value",
    "attn_output": "attention.output.dense", "attn_output_norm": "attention.output.LayerNorm",
    "ffn_up": "intermediate.dense", "ffn_down": "output.dense", "layer_output_norm": "output.LayerNorm",
}
_GGUF_TOP = {
    "token_embd.weight": "embeddings.word_embeddings.weight",
    "position_embd.weight": "embeddings.position_embeddings.weight",
    "token_types.weight": "embeddings.token_type_embeddings.weight",
    "token_embd_norm.weight": "embeddings.LayerNorm.weight",
    "token_embd_norm.bias": "embeddings.LayerNorm.bias",
}


def gguf_to_bert_names(tensors: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """llama.cpp BERT tensor names -> BertModel state-dict names (a fused attn_qkv is split in three)."""
    out: Dict[str, np.ndarray] = {}
    for name, w in tensors.items():
        if name in _GGUF_TOP:
            out[_GGUF_TOP[name]] = w
            continue
        parts = name.split(".")
        if len(parts) == 4 and parts[0] == "blk":
            layer, kind, leaf = parts[1], parts[2], parts[3]
            prefix = f"encoder.layer.{layer}."
            if kind == "attn_qkv":
                h = w.shape[0] // 3
                for i, nm in enumerate(("query", "key", "value")):
                    out[f"{prefix}attention.self.{nm}.{leaf}"] = np.ascontiguousarray(w[i * h:(i + 1) * h])
            elif kind in _GGUF_LAYER:
                out[f"{prefix}{_GGUF_LAYER[kind]}.{leaf}"] = w
    return out


def gguf_vocab(meta: Dict[str, object]) -> str:
    """``tokenizer.ggml.tokens`` -> vocab.txt text.  llama.cpp stores BERT word pieces in 'phantom space' form
    (continuations without '##', word starts prefixed with U+2581, ``[SPECIAL]`` untouched); undo that."""
    toks = meta.get("tokenizer.ggml.tokens")
    if not isinstance(toks, list):
        raise ValueError("gguf: no tokenizer.ggml.tokens")
    lines = []
    for t in toks:
        if t.startswith("[") and t.endswith("]"):
            lines.append(t)
        elif t.startswith("▁"):
            lines.append(t[1:])
        else:
            lines.append("##" + t)
    return "\n".join(lines) + "\n"


# ------------------------------------------------------------------ Hugging Face directory


def normalise_bert_names(tensors: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Drop the ``bert.`` / ``model.`` prefixes some exports add; pooler and buffers are ignored by the loader."""
    out = {}
    for name, w in tensors.items():
        for p in ("bert.", "model.", "roberta."):
            if name.startswith(p):
                name = name[len(p):]
        out[name] = w
    return out


def vocab_from_tokenizer_json(path: str) -> str:
    with open(path, "r", encoding="utf-8") as f:
        tj = json.load(f)
    vocab = tj["model"]["vocab"]
    inv = sorted(vocab.items(), key=lambda kv: kv[1])
    if [i for _, i in inv] != list(range(len(inv))):
        raise ValueError("tokenizer.json: vocabulary ids are not 0..n-1")
    return "\n".join(t for t, _ in inv) + "\n"


def load_local_model(path: str) -> Tuple[Dict[str, float], Dict[str, np.ndarray], str]:
    """path: HF directory or .gguf file -> (encoder config kwargs, weights by BertModel name, vocab.txt text)."""
    if os.path.isdir(path):
        with open(os.path.join(path, "config.json"), "r", encoding="utf-8") as f:
            c = json.load(f)
        cfg = dict(vocab_size=c["vocab_size"], hidden=c["hidden_size"], layers=c["num_hidden_layers"],
                   heads=c["num_attention_heads"], inter=c["intermediate_size"],
                   max_pos=c["max_position_embeddings"], type_vocab=c.get("type_vocab_size", 2),
                   ln_eps=c.get("layer_norm_eps", 1e-12))
        st = os.path.join(path, "model.safetensors")
        weights = normalise_bert_names(read_safetensors(st))
        vt = os.path.join(path, "vocab.txt")
        if os.path.exists(vt):
            with open(vt, "r", encoding="utf-8") as f:
                vocab = f.read()
        else:
            vocab = vocab_from_tokenizer_json(os.path.join(path, "tokenizer.json"))
        return cfg, weights, vocab
    meta, tensors = read_gguf(path)
    arch = meta.get("general.architecture", "bert")
    weights = gguf_to_bert_names(tensors)
    emb = weights["embeddings.word_embeddings.weight"]
    cfg = dict(vocab_size=emb.shape[0], hidden=int(meta[f"{arch}.embedding_length"]),
               layers=int(meta[f"{arch}.block_count"]), heads=int(meta[f"{arch}.attention.head_count"]),
               inter=int(meta[f"{arch}.feed_forward_length"]),
               max_pos=weights["embeddings.position_embeddings.weight"].shape[0],
               type_vocab=weights["embeddings.token_type_embeddings.weight"].shape[0],
               ln_eps=float(meta.get(f"{arch}.attention.layer_norm_epsilon", 1e-12)))
    return cfg, weights, gguf_vocab(meta)


def embedder_from_local(ctx, path: str, max_len: int = 512):
    """Build the tokenizer + encoder pair (retrieval.Embedder) from local files; the GPU library must be present."""
    from .encoder import BertEncoder
    from .retrieval import Embedder
    from .tokenizer import WordPieceTokenizer
    cfg, weights, vocab = load_local_model(path)
    enc = BertEncoder(ctx, **cfg)
    enc.load_weights(weights)
    return Embedder(enc, WordPieceTokenizer(vocab_text=vocab), max_len=min(max_len, cfg["max_pos"]))
