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


def encoder_flops(tokens: int, s: int) -> float:
    """per token 24 * 2 * (4 * 1024^2 + 2 * 1024 * 4096) = 604 MFLOP of GEMMs plus
    24 * 4 * S * 1024 of attention (654 MFLOP per token at S = 512)."""
    return float(tokens) * (24 * 2 * (4 * 1024 * 1024 + 2 * 1024 * 4096) + 24 * 4 * s * 1024)


def mfma_roofline(flops: float, ms: float) -> dict:
    tf = flops / ms / 1e9
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
            "traffic": None, "algorithmic_flops": flops}


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


def cpu_encoder_baseline(cases, max_seconds: float = 40.0) -> dict:
    """CPU baseline of the encoder leg: torch CPU `BertModel` with the
    BERT-large configuration and seeded random weights (fp32, no pooler), all host cores this process may use, for each
    (batch, seq_len) in `cases` -- (1, 512) is what Ollama does per request (main.py:134-145: one text per call).
    A reported baseline, timed AFTER the GPU measurements; bounded: one warm-up + as many passes as fit in
    `max_seconds` per case (at least one)."""
    import transformers
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # a one-GPU job on the GPU box has a CPU share of 16 whatever the affinity mask says (256): r03's first run took
    # 33 s for ONE 32-token text with 256 threads on that share
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    cfg = transformers.BertConfig(vocab_size=BERT_LARGE["vocab_size"], hidden_size=1024, num_hidden_layers=24,
                                  num_attention_heads=16, intermediate_size=4096, max_position_embeddings=512,
                                  type_vocab_size=2, hidden_act="gelu", layer_norm_eps=1e-12)
    torch.manual_seed(0)
    model = transformers.BertModel(cfg, add_pooling_layer=False).eval()
    out = {"kind": "port", "cores": int(cores), "model": "transformers.BertModel, BERT-large config, random fp32 weights, torch CPU",
           "cases": []}
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for b, sl in cases:
            ids = torch.randint(1000, BERT_LARGE["vocab_size"], (b, sl), generator=g)
            t0 = time.perf_counter()
            model(input_ids=ids)                                  # warm-up (allocations, thread pool)
            first = time.perf_counter() - t0
            n, t0 = 0, time.perf_counter()
            while n == 0 or (time.perf_counter() - t0 + first < max_seconds and n < 5):
                model(input_ids=ids)
                n += 1
            sec = (time.perf_counter() - t0) / n
            out["cases"].append({"batch": b, "seq_len": sl, "ms": round(sec * 1e3, 1), "tokens_per_s": round(b * sl / sec),
                                 "texts_per_s": round(b / sec, 2), "tflops": round(encoder_flops(b * sl, sl) / sec / 1e12, 3),
                                 "passes": n})
    return out


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
        from semantic_query_engine_amd import SCAN_BF16_RESCORE, SCAN_INT8_RESCORE
        idx.set_option("scan_mode", SCAN_INT8_RESCORE)
        srch8_ms = timed(lambda: idx.search_device(emb.data_ptr(), 64, 10, cos.data_ptr(), idk.data_ptr()), ctx.synchronize, 10)
        idx.set_option("scan_mode", SCAN_BF16_RESCORE)
        srch_ms = timed(lambda: idx.search_device(emb.data_ptr(), 64, 10, cos.data_ptr(), idk.data_ptr()), ctx.synchronize, 10)
        flops = encoder_flops(64 * s, s)
        # search at B = 64 is HBM-bound: one read of the bf16 scan copy (N * D * 2 + B * D * 4 + B * k * 12)
        sbytes = args.rows * D * 2 + 64 * D * 4 + 64 * 10 * 12
        out["cases"].append({"seq_len": s, "encode_ms": round(enc_ms, 3), "search_ms": round(srch_ms, 3),
                             "search_ms_int8_first_pass": round(srch8_ms, 3),
                             "encode_tflops": round(flops / enc_ms / 1e9, 1),
                             "roofline_encode": mfma_roofline(flops, enc_ms),
                             "roofline_search": {"bound": "hbm", "achieved": round(sbytes / srch_ms / 1e6, 1), "peak": 8000.0,
                                                 "unit": "GB/s", "frac": round(sbytes / srch_ms / 1e6 / 8000.0, 4), "traffic": None,
                                                 "algorithmic_bytes": sbytes}})
    if not args.no_cpu_baseline:
        del idx
        out["cpu_baseline"] = cpu_encoder_baseline([(1, 32), (64, 32), (1, 512)])
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
    flops = encoder_flops(b * s, s)
    res = {"mode": "encode", "batch": b, "seq_len": s, "ms": round(ms, 2), "tokens_per_s": round(b * s / ms * 1e3),
           "chunks_per_s": round(b / ms * 1e3, 1), "mfma_tflops": round(flops / ms / 1e9, 1),
           "frac_of_bf16_peak": round(flops / ms / 1e9 / 2500.0, 4), "roofline": mfma_roofline(flops, ms)}
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_encoder_baseline([(1, 512), (b, 512)])
    print(json.dumps(res), flush=True)


def build_clustered(ctx, rows, dev, centres, kind, nlist, keep_host=None):
    """Clustered index of SURVEY 8(d) (4096 Gaussian centres x sigma 0.3), block by block; `keep_host` (a list)
    receives the raw blocks as NumPy arrays for the CPU oracle."""
    from semantic_query_engine_amd import VectorIndex
    idx = VectorIndex(ctx, D, kind, nlist)
    idx.reserve(rows)
    for b in range((rows + BLOCK - 1) // BLOCK):
        n = min(BLOCK, rows - b * BLOCK)
        g = torch.Generator(device=dev).manual_seed(1000 + b)
        x = torch.randn((n, D), generator=g, device=dev)
        lab = torch.randint(0, centres.shape[0], (n,), generator=g, device=dev)
        x = centres[lab] + 0.3 * x
        torch.cuda.synchronize()
        idx.add_device(x.data_ptr(), n)
        ctx.synchronize()
        if keep_host is not None:
            keep_host.append(x.cpu().numpy())
        del x
    return idx


def mode_ivf(args, ctx, dev):
    """Config 5.  Two checks of the IVF answer on a 64-query probe, neither of them the HIP flat scan:
    `parity_vs_oracle_ivf` = share of returned ids equal to oracle.retrieval.ivf_search run on the centroids and
    list assignment the index exports; `recall_at_10_vs_exact` = recall against an independent exact scan (torch
    fp32 matmul over the same rows).  The flat index is timed beside it on the same data."""
    from oracle import retrieval as R      # checker only: never timed here
    from semantic_query_engine_amd import INDEX_FLAT, INDEX_IVF_FLAT
    g = torch.Generator(device=dev).manual_seed(99)
    centres = torch.randn((4096, D), generator=g, device=dev)
    nlist, nprobe, b, k = 4096, 32, args.batch, 10
    host_blocks = []
    ivf = build_clustered(ctx, args.rows, dev, centres, INDEX_IVF_FLAT, nlist, keep_host=host_blocks)
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
    ci = torch.empty((b, k), device=dev); ji = torch.empty((b, k), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ivf_ms = timed(lambda: ivf.search_device(q.data_ptr(), b, k, ci.data_ptr(), ji.data_ptr(), nprobe=nprobe), ctx.synchronize, 5)
    # batch sweep, timed HERE: before the NumPy / OpenBLAS checks below, whose worker threads keep spinning on the box's 16-CPU
    # share afterwards and delayed the host side of these short searches (r03: batch 1024 measured 7.2 ms after them against 4.4)
    sweep_ms = {}
    for bb in (1, 8, 64, 256, 1024):
        if bb > b:
            break
        co = torch.empty((bb, k), device=dev); jo = torch.empty((bb, k), dtype=torch.int64, device=dev)
        sweep_ms[bb] = timed(lambda: ivf.search_device(q.data_ptr(), bb, k, co.data_ptr(), jo.data_ptr(), nprobe=nprobe), ctx.synchronize, 10)
    # ---- the checks (64-query probe)
    probe = torch.cat([torch.arange(0, 32), torch.arange(b - 32, b)]).to(dev) if b >= 64 else torch.arange(b, device=dev)
    got = ji[probe].cpu().numpy()
    qp = q[probe]
    qn = qp / (qp.norm(dim=1, keepdim=True) + 1e-9)
    best_s = torch.full((probe.numel(), 0), -1e30, device=dev)
    best_i = torch.zeros((probe.numel(), 0), dtype=torch.long, device=dev)
    lo = 0
    for blk in host_blocks:                                # independent exact scan, block by block
        x = torch.from_numpy(blk).to(dev)
        xn = x / (x.norm(dim=1, keepdim=True) + 1e-9)
        sc = torch.cat([best_s, qn @ xn.T], 1)
        ii = torch.cat([best_i, torch.arange(lo, lo + x.shape[0], device=dev).expand(probe.numel(), -1)], 1)
        top = torch.topk(sc, k, dim=1)
        best_s, best_i = top.values, torch.gather(ii, 1, top.indices)
        lo += x.shape[0]
        del x, xn, sc, ii
    exact_ids = best_i.cpu().numpy()
    recall_exact = float(np.mean([len(set(a.tolist()) & set(e.tolist())) / k for a, e in zip(got, exact_ids)]))
    centroids, assign = ivf.ivf_export(nlist)
    xn_host = np.concatenate([R.normalize_rows(blk) for blk in host_blocks])
    del host_blocks
    _, ref_ids = R.ivf_search(xn_host, R.normalize_rows(qp.cpu().numpy()), centroids, assign, k, nprobe)
    parity = float(np.mean(got == ref_ids))
    del xn_host
    # ---- batch sweep (r02 verdict: the reference issues B = 1, main.py:355): IVF at 1 / 8 / 64 / 256 / 1024 of the same
    # queries (timed above).  Algorithmic bytes of a point: the rows of the DISTINCT lists its queries probe (from the exported
    # assignment and the oracle's probe order on the exported centroids) x D x 1 byte (int8 list scan).
    sweep = []
    cen64 = centroids.astype(np.float64)
    list_len = np.bincount(assign, minlength=nlist)
    qn_all = R.normalize_rows(q.cpu().numpy()).astype(np.float64)
    for bb in (1, 8, 64, 256, 1024):
        if bb > b:
            break
        t_ivf = sweep_ms[bb]
        probes = np.argsort(-(qn_all[:bb] @ cen64.T), axis=1, kind="stable")[:, :nprobe]
        rows_touched = int(list_len[np.unique(probes)].sum())
        by = rows_touched * D * 1          # int8 list scan (r03): one byte per element of a probed row
        sweep.append({"batch": bb, "ivf_ms": round(t_ivf, 4), "ivf_qps": round(bb / t_ivf * 1e3), "rows_in_probed_lists": rows_touched,
                      "roofline": {"bound": "hbm", "achieved": round(by / t_ivf / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                                   "frac": round(by / t_ivf / 1e6 / 8000.0, 4), "traffic": None, "algorithmic_bytes": by}})
    # ---- the flat scan on the same data, for the QPS comparison only
    flat = build_clustered(ctx, args.rows, dev, centres, INDEX_FLAT, 0)
    cf = torch.empty((b, k), device=dev); jf = torch.empty((b, k), dtype=torch.int64, device=dev)
    flat_ms = timed(lambda: flat.search_device(q.data_ptr(), b, k, cf.data_ptr(), jf.data_ptr()), ctx.synchronize, 5)
    ivf_bytes = args.rows * D * 1                          # every list is probed at B = 1024: one read of the int8 copy (r03; bf16 scan: x 2)
    for pt in sweep:                                       # the flat index at the same batch sizes (every query takes the collect pass here)
        bb = pt["batch"]
        t_flat = timed(lambda: flat.search_device(q.data_ptr(), bb, k, cf.data_ptr(), jf.data_ptr()), ctx.synchronize, 5)
        pt["flat_ms"] = round(t_flat, 4); pt["flat_qps"] = round(bb / t_flat * 1e3)
    print(json.dumps({"mode": "ivf", "rows": args.rows, "nlist": nlist, "nprobe": nprobe, "batch": b, "train_s": round(train_s, 2),
                      "flat_ms": round(flat_ms, 3), "flat_qps": round(b / flat_ms * 1e3), "ivf_ms": round(ivf_ms, 3),
                      "ivf_qps": round(b / ivf_ms * 1e3), "probe_queries": int(probe.numel()),
                      "parity_vs_oracle_ivf": round(parity, 4), "recall_at_10_vs_exact": round(recall_exact, 4),
                      "batch_sweep": sweep,
                      "roofline": {"bound": "hbm", "achieved": round(ivf_bytes / ivf_ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                                   "frac": round(ivf_bytes / ivf_ms / 1e6 / 8000.0, 4), "traffic": None,
                                   "algorithmic_bytes": ivf_bytes}}), flush=True)


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
    from semantic_query_engine_amd import SCAN_BF16_RESCORE, SCAN_INT8_RESCORE
    out = {"mode": "hard", "rows": args.rows + 3000, "batch": b, "cases": []}
    for n_hard in (0, 8, 200, b):
        q = torch.randn((b, D), generator=g, device=dev)
        if n_hard:
            q[:n_hard] = centre + 3e-3 * torch.randn((n_hard, D), generator=g, device=dev)
        torch.cuda.synchronize()
        case = {"hard_queries": n_hard}
        for name, mode in (("bf16", SCAN_BF16_RESCORE), ("int8", SCAN_INT8_RESCORE)):     # both first passes end in the same bf16 collect pass
            idx.set_option("scan_mode", mode)
            ctx.stats_reset()
            ms = timed(lambda: idx.search_device(q.data_ptr(), b, k, cos.data_ptr(), ids.data_ptr()), ctx.synchronize, 5)
            case[f"uncertified_{name}"] = int(ctx.stats()["uncertified"])
            case[f"ms_{name}"] = round(ms, 3)
        out["cases"].append(case)
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
    ap.add_argument("--no-cpu-baseline", action="store_true", help="encode / e2e: skip the torch CPU BertModel baseline")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from semantic_query_engine_amd import Context
    ctx = Context(0)
    {"e2e": mode_e2e, "ivf": mode_ivf, "encode": mode_encode, "cache": mode_cache, "hard": mode_hard, "ingest": mode_ingest}[args.mode](args, ctx, dev)


if __name__ == "__main__":
    main()
