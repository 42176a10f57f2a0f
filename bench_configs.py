#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs that are not the headline line of bench.py:

    --mode e2e     config 3: on-GPU BERT-large (random bf16 weights) encoding a 64-query batch -> k-NN over
                   N rows; reports the encode / search latency split
    --mode ivf     config 5: IVF-flat nlist=4096 nprobe=32 on clustered data vs the flat scan at the same N
    --mode encode  encoder throughput on full-length chunks (B x 512 tokens), tokens/s and MFMA TFLOP/s
    --mode ingest  config 1 plumbing: text chunks -> tokens -> embeddings -> index, chunks/s
    --mode hard    flat scan when 0 / 8 / 200 / all queries fail the exactness certificate
    --mode cache   cache scan latency (1000 x 1024) on the GPU vs the reference's Python loop (main.py:73-87)

Each mode prints one JSON line.  Single GPU.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

D = 1024
BLOCK = 1 << 20


BERT_LARGE = dict(vocab_size=30522, hidden=1024, layers=24, heads=16, inter=4096, max_pos=512, type_vocab=2)


def random_bert_weights(cfg=BERT_LARGE, seed: int = 0):
    """Seeded N(0, 0.02) BERT parameters by BertModel state-dict name (LayerNorm weights around 1): the model
    geometry the reference's embedding model has, with random values -- no checkpoint exists offline."""
    h, i = cfg["hidden"], cfg["inter"]
    shapes = {"embeddings.word_embeddings.weight": (cfg["vocab_size"], h), "embeddings.position_embeddings.weight": (cfg["max_pos"], h),
              "embeddings.token_type_embeddings.weight": (cfg["type_vocab"], h), "embeddings.LayerNorm.weight": (h,),
              "embeddings.LayerNorm.bias": (h,)}
    for l in range(cfg["layers"]):
        p = f"encoder.layer.{l}."
        for nm in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            shapes[p + nm + ".weight"], shapes[p + nm + ".bias"] = (h, h), (h,)
        shapes[p + "intermediate.dense.weight"], shapes[p + "intermediate.dense.bias"] = (i, h), (i,)
        shapes[p + "output.dense.weight"], shapes[p + "output.dense.bias"] = (h, i), (h,)
        for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
            shapes[p + nm + ".weight"], shapes[p + nm + ".bias"] = (h,), (h,)
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in shapes.items():
        w = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        out[name] = w + 1.0 if name.endswith("LayerNorm.weight") else w
    return out


def timed(fn, sync, iters, warmup=2):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    sync()
    return (time.perf_counter() - t0) / iters * 1e3


def build_random_index(ctx, rows, dev, kind=0, nlist=0, clustered=None):
    from semantic_query_engine_amd import VectorIndex
    idx = VectorIndex(ctx, D, kind, nlist)
    idx.reserve(rows)
    for b in range((rows + BLOCK - 1) // BLOCK):
        n = min(BLOCK, rows - b * BLOCK)
        g = torch.Generator(device=dev).manual_seed(1000 + b)
        x = torch.randn((n, D), generator=g, device=dev)
        if clustered is not None:
            lab = torch.randint(0, clustered.shape[0], (n,), generator=g, device=dev)
            x = clustered[lab] + 0.3 * x
        torch.cuda.synchronize()
        idx.add_device(x.data_ptr(), n)
        ctx.synchronize()
        del x
    return idx


def mode_e2e(args, ctx, dev):
    from semantic_query_engine_amd.encoder import BertEncoder
    enc = BertEncoder(ctx)
    enc.load_weights(random_bert_weights())
    idx = build_random_index(ctx, args.rows, dev)
    out = {"mode": "e2e", "rows": args.rows, "batch": 64, "k": 10, "cases": []}
    for s in (16, 32, 128):
        g = torch.Generator(device=dev).manual_seed(s)
        ids = torch.randint(1000, BERT_LARGE["vocab_size"], (64, s), generator=g, device=dev, dtype=torch.int32)
        lens = torch.full((64,), s, device=dev, dtype=torch.int32)
        emb = torch.empty((64, D), device=dev)
        cos = torch.empty((64, 10), device=dev)
        idk = torch.empty((64, 10), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        enc_ms = timed(lambda: enc.encode_ids_device(ids.data_ptr(), lens.data_ptr(), 64, s, emb.data_ptr()), ctx.synchronize, 10)
        srch_ms = timed(lambda: idx.search_device(emb.data_ptr(), 64, 10, cos.data_ptr(), idk.data_ptr()), ctx.synchronize, 10)
        flops = 64 * s * (24 * 2 * (4 * 1024 * 1024 + 2 * 1024 * 4096) + 24 * 4 * s * 1024)
        out["cases"].append({"seq_len": s, "encode_ms": round(enc_ms, 3), "search_ms": round(srch_ms, 3),
                             "encode_tflops": round(flops / enc_ms / 1e9, 1)})
    print(json.dumps(out), flush=True)


def mode_encode(args, ctx, dev):
    from semantic_query_engine_amd.encoder import BertEncoder
    enc = BertEncoder(ctx)
    enc.load_weights(random_bert_weights())
    b, s = args.batch, 512
    g = torch.Generator(device=dev).manual_seed(1)
    ids = torch.randint(1000, BERT_LARGE["vocab_size"], (b, s), generator=g, device=dev, dtype=torch.int32)
    lens = torch.full((b,), s, device=dev, dtype=torch.int32)
    emb = torch.empty((b, D), device=dev)
    torch.cuda.synchronize()
    ms = timed(lambda: enc.encode_ids_device(ids.data_ptr(), lens.data_ptr(), b, s, emb.data_ptr()), ctx.synchronize, 3, 1)
    flops = b * s * (24 * 2 * (4 * 1024 * 1024 + 2 * 1024 * 4096) + 24 * 4 * s * 1024)
    print(json.dumps({"mode": "encode", "batch": b, "seq_len": s, "ms": round(ms, 2), "tokens_per_s": round(b * s / ms * 1e3),
                      "chunks_per_s": round(b / ms * 1e3, 1), "mfma_tflops": round(flops / ms / 1e9, 1),
                      "frac_of_bf16_peak": round(flops / ms / 1e9 / 2500.0, 4)}), flush=True)


def mode_ivf(args, ctx, dev):
    from semantic_query_engine_amd import INDEX_IVF_FLAT
    g = torch.Generator(device=dev).manual_seed(99)
    centres = torch.randn((4096, D), generator=g, device=dev)
    nlist, nprobe, b, k = 4096, 32, args.batch, 10
    flat = build_random_index(ctx, args.rows, dev, clustered=centres)
    ivf = build_random_index(ctx, args.rows, dev, INDEX_IVF_FLAT, nlist, clustered=centres)
    # train on a 1M-row sample (the first block of the same recipe), 20 Lloyd iterations, seed 0
    gs = torch.Generator(device=dev).manual_seed(1000)
    n_s = min(BLOCK, args.rows)
    xs = torch.randn((n_s, D), generator=gs, device=dev)
    lab = torch.randint(0, 4096, (n_s,), generator=gs, device=dev)
    xs = centres[lab] + 0.3 * xs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ivf.train_device(xs.data_ptr(), n_s, iters=20, seed=0)
    ctx.synchronize()
    train_s = time.perf_counter() - t0
    del xs
    gq = torch.Generator(device=dev).manual_seed(5)
    q = centres[torch.randint(0, 4096, (b,), generator=gq, device=dev)] + 0.3 * torch.randn((b, D), generator=gq, device=dev)
    cf = torch.empty((b, k), device=dev); jf = torch.empty((b, k), dtype=torch.int64, device=dev)
    ci = torch.empty((b, k), device=dev); ji = torch.empty((b, k), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    flat_ms = timed(lambda: flat.search_device(q.data_ptr(), b, k, cf.data_ptr(), jf.data_ptr()), ctx.synchronize, 5)
    ivf_ms = timed(lambda: ivf.search_device(q.data_ptr(), b, k, ci.data_ptr(), ji.data_ptr(), nprobe=nprobe), ctx.synchronize, 5)
    hits = sum(len(set(a.tolist()) & set(r.tolist())) for a, r in zip(ji.cpu(), jf.cpu()))
    print(json.dumps({"mode": "ivf", "rows": args.rows, "nlist": nlist, "nprobe": nprobe, "batch": b, "train_s": round(train_s, 2),
                      "flat_ms": round(flat_ms, 3), "flat_qps": round(b / flat_ms * 1e3), "ivf_ms": round(ivf_ms, 3),
                      "ivf_qps": round(b / ivf_ms * 1e3), "recall_at_10_vs_flat": round(hits / (b * k), 4)}), flush=True)


def mode_hard(args, ctx, dev):
    """Cost of certificate failures: a tight cluster of 3000 near-identical rows inside a random index, and
    0 / 8 / 200 / all of the 1024 queries aimed at it (those cannot be certified from bf16 scores and take the
    compacted collect pass)."""
    idx = build_random_index(ctx, args.rows, dev)
    g = torch.Generator(device=dev).manual_seed(3)
    centre = torch.randn((1, D), generator=g, device=dev)
    cluster = centre + 3e-3 * torch.randn((3000, D), generator=g, device=dev)
    torch.cuda.synchronize()
    idx.add_device(cluster.data_ptr(), 3000)
    ctx.synchronize()
    b, k = args.batch, 10
    cos = torch.empty((b, k), device=dev); ids = torch.empty((b, k), dtype=torch.int64, device=dev)
    out = {"mode": "hard", "rows": args.rows + 3000, "batch": b, "cases": []}
    for n_hard in (0, 8, 200, b):
        q = torch.randn((b, D), generator=g, device=dev)
        if n_hard:
            q[:n_hard] = centre + 3e-3 * torch.randn((n_hard, D), generator=g, device=dev)
        torch.cuda.synchronize()
        ctx.stats_reset()
        ms = timed(lambda: idx.search_device(q.data_ptr(), b, k, cos.data_ptr(), ids.data_ptr()), ctx.synchronize, 5)
        out["cases"].append({"hard_queries": n_hard, "uncertified": int(ctx.stats()["uncertified"]), "ms": round(ms, 3)})
    print(json.dumps(out), flush=True)


def mode_ingest(args, ctx, dev):
    """Config 1 plumbing: text chunks of 512 words -> C++ WordPiece -> BERT-large (random bf16 weights) ->
    normalise + index, through the reference-named bulk call (`embed_texts_in_batches`, main.py:148-169) and
    `OpenSearchIndexer.add_embeddings` (main.py:309-338).  Synthetic text over a synthetic 8k-word vocabulary."""
    import asyncio
    from semantic_query_engine_amd import retrieval as RT
    from semantic_query_engine_amd.encoder import BertEncoder
    from semantic_query_engine_amd.tokenizer import WordPieceTokenizer
    enc = BertEncoder(ctx)
    enc.load_weights(random_bert_weights())
    rng = np.random.default_rng(0)
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
    words = sorted({"".join(rng.choice(letters, rng.integers(3, 10))) for _ in range(9000)})[:8000]
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words + ["##" + w[:3] for w in words[:2000]]
    vocab += [f"[unused{i}]" for i in range(BERT_LARGE["vocab_size"] - len(vocab))]
    tok = WordPieceTokenizer(vocab_text="\n".join(vocab) + "\n")
    RT.configure_embedder(RT.Embedder(enc, tok))
    n = args.batch if args.batch != 1024 else 2048
    texts = [" ".join(words[j] for j in rng.integers(0, len(words), 512)) for _ in range(n)]
    docs = [{"doc_id": f"PMC{i // 11}.txt", "text": t} for i, t in enumerate(texts)]
    asyncio.run(RT.embed_texts_in_batches(texts[:128]))                      # warm-up
    t0 = time.perf_counter()
    ids, lens = tok.encode_batch(texts, 512)
    tok_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    emb = asyncio.run(RT.embed_texts_in_batches(texts))
    embed_s = time.perf_counter() - t0
    ix = RT.OpenSearchIndexer(RT.GpuSearchClient(ctx, dim=D), "medical-search-index")
    t0 = time.perf_counter()
    ix.add_embeddings(emb, docs)
    ctx.synchronize()
    add_s = time.perf_counter() - t0
    print(json.dumps({"mode": "ingest", "chunks": n, "mean_tokens": round(float(lens.mean()), 1),
                      "tokenize_only_s": round(tok_s, 3), "embed_s": round(embed_s, 3), "index_add_s": round(add_s, 3),
                      "chunks_per_s": round(n / (embed_s + add_s), 1),
                      "note": "embed_s includes tokenisation (overlapped with the GPU on a worker thread)"}), flush=True)


def mode_cache(args, ctx, dev):
    from oracle import retrieval as R      # CPU baseline leg only: the reference's Python loop, restated (main.py:73-87)
    from semantic_query_engine_amd.retrieval import SemanticLfuCache
    rng = np.random.default_rng(0)
    m = rng.standard_normal((1000, D)).astype(np.float32)
    cache = SemanticLfuCache(ctx, max_items=1000)
    ora = R.LfuCacheOracle(max_items=1000)
    for i in range(1000):
        cache.put(m[i:i + 1], f"r{i}")
        ora.put(m[i:i + 1], f"r{i}")
    qv = (m[123] + 0.01 * rng.standard_normal(D).astype(np.float32))[None]
    t0 = time.perf_counter()
    for _ in range(200):
        cache.get(qv)
    gpu_ms = (time.perf_counter() - t0) / 200 * 1e3
    t0 = time.perf_counter()
    for _ in range(3):
        ora.get(qv)
    ref_ms = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"mode": "cache", "entries": 1000, "gpu_get_ms": round(gpu_ms, 4),
                      "reference_python_loop_ms": round(ref_ms, 2), "same_answer": cache.get(qv) == ora.get(qv)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True, choices=["e2e", "ivf", "encode", "cache", "hard", "ingest"])
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=1024)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from semantic_query_engine_amd import Context
    ctx = Context(0)
    {"e2e": mode_e2e, "ivf": mode_ivf, "encode": mode_encode, "cache": mode_cache, "hard": mode_hard, "ingest": mode_ingest}[args.mode](args, ctx, dev)


if __name__ == "__main__":
    main()
