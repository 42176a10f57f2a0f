"""Local model files -> weights + vocabulary (semantic_query_engine_amd/weights.py).  CPU tests: the two
container readers against files written here (safetensors through the `safetensors` package and
transformers' `save_pretrained`; GGUF through a small writer that follows the published layout), the
llama.cpp name and vocabulary mappings.  A GPU test loads a saved toy model end to end."""
import json
import os
import struct

import numpy as np
import pytest

from semantic_query_engine_amd import weights as W


def _bf16_bytes(a: np.ndarray) -> bytes:
    u = a.astype(np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)       # round to nearest even
    return r.tobytes()


def test_safetensors_f32_f16_bf16(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((5, 7)).astype(np.float32)
    b = rng.standard_normal((3,)).astype(np.float16)
    c = rng.standard_normal((4, 2)).astype(np.float32)
    blobs = [("a", "F32", a.shape, a.tobytes()), ("b", "F16", b.shape, b.tobytes()),
             ("c", "BF16", c.shape, _bf16_bytes(c)), ("ids", "I64", (1, 4), np.arange(4, dtype=np.int64).tobytes())]
    header, off, data = {"__metadata__": {"format": "pt"}}, 0, b""
    for name, dt, shape, raw in blobs:
        header[name] = {"dtype": dt, "shape": list(shape), "data_offsets": [off, off + len(raw)]}
        off += len(raw)
        data += raw
    hj = json.dumps(header).encode()
    p = tmp_path / "m.safetensors"
    p.write_bytes(struct.pack("<Q", len(hj)) + hj + data)
    t = W.read_safetensors(str(p))
    assert set(t) == {"a", "b", "c"}                       # the integer buffer is skipped
    assert np.array_equal(t["a"], a) and t["a"].dtype == np.float32
    assert np.array_equal(t["b"], b.astype(np.float32))
    assert np.allclose(t["c"], c, rtol=2 ** -8) and np.array_equal(t["c"], W._bf16_to_f32(np.frombuffer(_bf16_bytes(c), "<u2")).reshape(4, 2))


def test_safetensors_written_by_the_safetensors_package(tmp_path):
    from safetensors.numpy import save_file
    rng = np.random.default_rng(1)
    d = {"x.weight": rng.standard_normal((8, 3)).astype(np.float32), "x.bias": rng.standard_normal(8).astype(np.float16)}
    save_file(d, str(tmp_path / "f.safetensors"))
    t = W.read_safetensors(str(tmp_path / "f.safetensors"))
    assert np.array_equal(t["x.weight"], d["x.weight"])
    assert np.array_equal(t["x.bias"], d["x.bias"].astype(np.float32))


def _toy_hf_dir(tmp_path, vocab_words):
    import torch
    from transformers import BertConfig, BertModel
    cfg = BertConfig(vocab_size=len(vocab_words), hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=256, max_position_embeddings=64, type_vocab_size=2)
    torch.manual_seed(0)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    d = tmp_path / "toy_model"
    model.save_pretrained(str(d), safe_serialization=True)
    (d / "vocab.txt").write_text("\n".join(vocab_words) + "\n", encoding="utf-8")
    return model, str(d)


VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "the", "cat", "sat", "on", "mat", "##s", "##ting", "a", ".", ","] + \
        [f"w{i}" for i in range(49)]


def test_hf_directory_matches_state_dict(tmp_path):
    model, d = _toy_hf_dir(tmp_path, VOCAB)
    cfg, weights, vocab = W.load_local_model(d)
    assert cfg == dict(vocab_size=len(VOCAB), hidden=128, layers=2, heads=2, inter=256, max_pos=64, type_vocab=2, ln_eps=1e-12)
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    for name, ref in sd.items():
        if ref.dtype != np.float32:
            continue
        assert np.array_equal(weights[name], ref), name
    assert vocab.split("\n")[:5] == VOCAB[:5]


def test_tokenizer_json_vocab(tmp_path):
    p = tmp_path / "tokenizer.json"
    p.write_text(json.dumps({"model": {"type": "WordPiece", "vocab": {t: i for i, t in enumerate(VOCAB)}}}), encoding="utf-8")
    assert W.vocab_from_tokenizer_json(str(p)) == "\n".join(VOCAB) + "\n"


# ---- a GGUF writer for the test (layout as published: header, typed KV pairs, tensor table, aligned data)
def _gs(s: str) -> bytes:
    b = s.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


def _write_gguf(path, meta, tensors, align=32):
    out = b"GGUF" + struct.pack("<I", 3) + struct.pack("<Q", len(tensors)) + struct.pack("<Q", len(meta))
    for k, v in meta.items():
        out += _gs(k)
        if isinstance(v, str):
            out += struct.pack("<I", 8) + _gs(v)
        elif isinstance(v, float):
            out += struct.pack("<I", 6) + struct.pack("<f", v)
        elif isinstance(v, int):
            out += struct.pack("<I", 4) + struct.pack("<I", v)
        elif isinstance(v, list):
            out += struct.pack("<I", 9) + struct.pack("<I", 8) + struct.pack("<Q", len(v)) + b"".join(_gs(s) for s in v)
    blobs, off = [], 0
    for name, (arr, ttype) in tensors.items():
        raw = arr.astype(np.float32).tobytes() if ttype == 0 else arr.astype(np.float16).tobytes() if ttype == 1 else _bf16_bytes(arr)
        dims = list(reversed(arr.shape))
        out += _gs(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims)
        out += struct.pack("<I", ttype) + struct.pack("<Q", off)
        pad = (-len(raw)) % align
        blobs.append(raw + b"\0" * pad)
        off += len(raw) + pad
    out += b"\0" * ((-len(out)) % align)
    with open(path, "wb") as f:
        f.write(out + b"".join(blobs))


def test_gguf_roundtrip_names_and_vocab(tmp_path):
    rng = np.random.default_rng(2)
    H, I, V = 8, 16, 12
    t = {
        "token_embd.weight": (rng.standard_normal((V, H)), 1),
        "position_embd.weight": (rng.standard_normal((6, H)), 0),
        "token_types.weight": (rng.standard_normal((2, H)), 0),
        "token_embd_norm.weight": (rng.standard_normal(H), 0),
        "token_embd_norm.bias": (rng.standard_normal(H), 0),
        "blk.0.attn_q.weight": (rng.standard_normal((H, H)), 1), "blk.0.attn_q.bias": (rng.standard_normal(H), 0),
        "blk.0.attn_k.weight": (rng.standard_normal((H, H)), 30), "blk.0.attn_k.bias": (rng.standard_normal(H), 0),
        "blk.0.attn_v.weight": (rng.standard_normal((H, H)), 0), "blk.0.attn_v.bias": (rng.standard_normal(H), 0),
        "blk.0.attn_output.weight": (rng.standard_normal((H, H)), 0), "blk.0.attn_output.bias": (rng.standard_normal(H), 0),
        "blk.0.attn_output_norm.weight": (rng.standard_normal(H), 0), "blk.0.attn_output_norm.bias": (rng.standard_normal(H), 0),
        "blk.0.ffn_up.weight": (rng.standard_normal((I, H)), 1), "blk.0.ffn_up.bias": (rng.standard_normal(I), 0),
        "blk.0.ffn_down.weight": (rng.standard_normal((H, I)), 1), "blk.0.ffn_down.bias": (rng.standard_normal(H), 0),
        "blk.0.layer_output_norm.weight": (rng.standard_normal(H), 0), "blk.0.layer_output_norm.bias": (rng.standard_normal(H), 0),
    }
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "▁the", "▁cat", "s", "ting", "▁.", "▁,", "▁a", "▁mat"]
    meta = {"general.architecture": "bert", "bert.embedding_length": H, "bert.block_count": 1,
            "bert.attention.head_count": 2, "bert.feed_forward_length": I,
            "bert.attention.layer_norm_epsilon": 1e-12, "tokenizer.ggml.model": "bert", "tokenizer.ggml.tokens": toks}
    p = str(tmp_path / "m.gguf")
    _write_gguf(p, meta, t)
    m2, t2 = W.read_gguf(p)
    assert m2["bert.block_count"] == 1 and m2["tokenizer.ggml.tokens"] == toks
    for name, (arr, ttype) in t.items():
        tol = 0 if ttype == 0 else 2 ** -10 if ttype == 1 else 2 ** -8
        assert t2[name].shape == arr.shape
        assert np.allclose(t2[name], arr.astype(np.float32), rtol=tol, atol=1e-6 if ttype else 0), name
    cfg, weights, vocab = W.load_local_model(p)
    assert cfg["hidden"] == H and cfg["layers"] == 1 and cfg["heads"] == 2 and cfg["inter"] == I
    assert cfg["vocab_size"] == V and cfg["max_pos"] == 6 and cfg["type_vocab"] == 2
    assert weights["encoder.layer.0.attention.self.query.weight"].shape == (H, H)
    assert np.array_equal(weights["encoder.layer.0.intermediate.dense.bias"], t["blk.0.ffn_up.bias"][0].astype(np.float32))
    assert np.array_equal(weights["encoder.layer.0.output.LayerNorm.weight"], t["blk.0.layer_output_norm.weight"][0].astype(np.float32))
    assert vocab.split("\n")[:8] == ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "the", "cat", "##s", "##ting"]


def test_gguf_fused_qkv_is_split(tmp_path):
    rng = np.random.default_rng(3)
    w = rng.standard_normal((12, 4)).astype(np.float32)
    out = W.gguf_to_bert_names({"blk.3.attn_qkv.weight": w})
    assert np.array_equal(out["encoder.layer.3.attention.self.query.weight"], w[0:4])
    assert np.array_equal(out["encoder.layer.3.attention.self.key.weight"], w[4:8])
    assert np.array_equal(out["encoder.layer.3.attention.self.value.weight"], w[8:12])


def test_quantised_gguf_is_refused(tmp_path):
    p = str(tmp_path / "q.gguf")
    out = b"GGUF" + struct.pack("<I", 3) + struct.pack("<Q", 1) + struct.pack("<Q", 0)
    out += _gs("x") + struct.pack("<I", 1) + struct.pack("<Q", 32) + struct.pack("<I", 2) + struct.pack("<Q", 0)   # Q4_0
    out += b"\0" * 64
    with open(p, "wb") as f:
        f.write(out)
    with pytest.raises(ValueError, match="quantised"):
        W.read_gguf(p)


@pytest.mark.gpu
def test_embedder_from_local_directory(tmp_path):
    """Saved toy BertModel + vocab.txt -> Embedder; embeddings match the torch model run on the same token ids."""
    import torch
    from semantic_query_engine_amd import Context
    model, d = _toy_hf_dir(tmp_path, VOCAB)
    ctx = Context(0)
    emb = W.embedder_from_local(ctx, d, max_len=64)
    texts = ["The cat sat on a mat.", "cats sitting , w3 w4 w5", ""]
    got = emb.embed(texts)
    ids, lens = emb.tokenizer.encode_batch(texts, 64)
    assert ids[0, 0] == 2 and ids[0, lens[0] - 1] == 3          # [CLS] ... [SEP]
    s = int(lens.max())
    mask = (np.arange(s)[None, :] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        ref = model(input_ids=torch.from_numpy(ids[:, :s].astype(np.int64)), attention_mask=torch.from_numpy(mask)).last_hidden_state[:, 0].numpy()
    for i in range(len(texts)):
        c = float(np.dot(got[i], ref[i]) / (np.linalg.norm(got[i]) * np.linalg.norm(ref[i])))
        assert c >= 0.999, (i, c)
