"""fp32 CPU restatement of the embedding model behind ``ollama_embed_text``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference posts each text to Ollama's ``/api/embeddings`` with model
``mxbai-embed-large:latest`` (main.py:29, 134-145) and gets 1024 floats back; the
arithmetic lives in Ollama/llama.cpp, which is absent from /root/reference and
unpinned there -> PARITY UNPINNED by the reference.  This file restates
the published architecture (BERT-large post-LN encoder, erf-GELU, CLS pooling,
SURVEY Appendix A) and is cross-checked against ``transformers.BertModel`` in
tests/test_oracle_bert.py.  Weight names follow the HF ``BertModel`` state dict so a
real ``model.safetensors`` can be dropped in.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List

import numpy as np
import torch


@dataclass
class BertCfg:
    vocab_size: int = 30522
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    inter: int = 4096
    max_pos: int = 512
    type_vocab: int = 2
    ln_eps: float = 1e-12

    @staticmethod
    def toy() -> "BertCfg":
        """Small config that exercises the same code path (head_dim stays 64)."""
        return BertCfg(vocab_size=512, hidden=128, layers=2, heads=2, inter=512, max_pos=64)


def weight_names(cfg: BertCfg) -> List[str]:
    names = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
             "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight",
             "embeddings.LayerNorm.bias"]
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."
        names += [p + "attention.self.query.weight", p + "attention.self.query.bias",
                  p + "attention.self.key.weight", p + "attention.self.key.bias",
                  p + "attention.self.value.weight", p + "attention.self.value.bias",
                  p + "attention.output.dense.weight", p + "attention.output.dense.bias",
                  p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias",
                  p + "intermediate.dense.weight", p + "intermediate.dense.bias",
                  p + "output.dense.weight", p + "output.dense.bias",
                  p + "output.LayerNorm.weight", p + "output.LayerNorm.bias"]
    return names


def weight_shape(cfg: BertCfg, name: str):
    h, i = cfg.hidden, cfg.inter
    if name.endswith("word_embeddings.weight"):
        return (cfg.vocab_size, h)
    if name.endswith("position_embeddings.weight"):
        return (cfg.max_pos, h)
    if name.endswith("token_type_embeddings.weight"):
        return (cfg.type_vocab, h)
    if "intermediate.dense" in name:
        return (i, h) if name.endswith("weight") else (i,)
    if ".output.dense" in name and "attention" not in name:
        return (h, i) if name.endswith("weight") else (h,)
    if name.endswith("weight") and "LayerNorm" not in name:
        return (h, h)
    return (h,)


def random_weights(cfg: BertCfg, seed: int = 0, bf16_round: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded N(0, 0.02) matrices / embeddings, N(0, 0.02) biases, LayerNorm weight
    1 + N(0, 0.02) (so the affine part is exercised).  With ``bf16_round`` every
    tensor is rounded to bf16 and widened back, so the fp32 oracle and the bf16 GPU
    encoder hold bit-identical parameters.  Generated per tensor from
    ``seed * 100003 + index`` so the GPU box regenerates them without shipping 668 MB."""
    out: Dict[str, torch.Tensor] = {}
    for idx, name in enumerate(weight_names(cfg)):
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        t = torch.randn(weight_shape(cfg, name), generator=g, dtype=torch.float32) * 0.02
        if "LayerNorm.weight" in name:
            t = t + 1.0
        if bf16_round:
            t = t.to(torch.bfloat16).to(torch.float32)
        out[name] = t
    return out


def _ln(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


@torch.no_grad()
def bert_encode(w: Dict[str, torch.Tensor], cfg: BertCfg, ids: np.ndarray, lens: np.ndarray,
                return_hidden: bool = False):
    """ids int [B,S] (padded with anything past lens), lens int [B] -> CLS row fp32 [B,H].
    Keys at positions >= lens[b] are masked out of the softmax; padded query rows are
    computed but never read."""
    ids_t = torch.as_tensor(np.asarray(ids), dtype=torch.long)
    b, s = ids_t.shape
    lens_t = torch.as_tensor(np.asarray(lens), dtype=torch.long)
    h, nh = cfg.hidden, cfg.heads
    dh = h // nh
    x = (w["embeddings.word_embeddings.weight"][ids_t]
         + w["embeddings.position_embeddings.weight"][:s][None]
         + w["embeddings.token_type_embeddings.weight"][0][None, None])
    x = _ln(x, w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], cfg.ln_eps)
    key_ok = torch.arange(s)[None, :] < lens_t[:, None]                     # [B,S]
    bias = torch.zeros(b, 1, 1, s)
    bias.masked_fill_(~key_ok[:, None, None, :], float("-inf"))
    hidden = [x]
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."
        def lin(t, n):
            return t @ w[p + n + ".weight"].T + w[p + n + ".bias"]
        q = lin(x, "attention.self.query").view(b, s, nh, dh).transpose(1, 2)
        k = lin(x, "attention.self.key").view(b, s, nh, dh).transpose(1, 2)
        v = lin(x, "attention.self.value").view(b, s, nh, dh).transpose(1, 2)
        sc = q @ k.transpose(-1, -2) / math.sqrt(dh) + bias
        ctx = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(b, s, h)
        x = _ln(x + lin(ctx, "attention.output.dense"),
                w[p + "attention.output.LayerNorm.weight"], w[p + "attention.output.LayerNorm.bias"],
                cfg.ln_eps)
        inter = lin(x, "intermediate.dense")
        inter = 0.5 * inter * (1.0 + torch.erf(inter / math.sqrt(2.0)))
        x = _ln(x + lin(inter, "output.dense"),
                w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], cfg.ln_eps)
        hidden.append(x)
    cls = x[:, 0].contiguous().numpy()
    if return_hidden:
        return cls, [t.numpy() for t in hidden]
    return cls
