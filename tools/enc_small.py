#!/usr/bin/env python3
"""Encoder at the shapes the reference calls it with (main.py:172-180: one query; BASELINE config 3: 64 queries of
16-128 tokens): BERT-large, random weights, B x S tokens; prints one JSON line per (B, S).
    python tools/enc_small.py --cases 64x16,64x32,64x128,1x16,1x32 [--iters 30]
Under rocprofv3 --kernel-trace --stats it gives the per-kernel summary of exactly these shapes."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_configs import random_bert_weights, timed, encoder_flops, mfma_roofline
from semantic_query_engine_amd import Context
from semantic_query_engine_amd.encoder import BertEncoder

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="64x16,64x32,64x128,1x16,1x32,1x128")
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
ctx = Context(0); dev = torch.device("cuda", 0)
enc = BertEncoder(ctx); enc.load_weights(random_bert_weights())
for case in args.cases.split(","):
    b, s = (int(v) for v in case.split("x"))
    g = torch.Generator(device=dev).manual_seed(s)
    ids = torch.randint(1000, 30000, (b, s), generator=g, device=dev, dtype=torch.int32)
    lens = torch.full((b,), s, device=dev, dtype=torch.int32)
    emb = torch.empty((b, 1024), device=dev)
    torch.cuda.synchronize()
    ms = timed(lambda: enc.encode_ids_device(ids.data_ptr(), lens.data_ptr(), b, s, emb.data_ptr()), ctx.synchronize, args.iters, 5)
    fl = encoder_flops(b * s, s)
    print(json.dumps({"batch": b, "seq_len": s, "encode_ms": round(ms, 4), "roofline": mfma_roofline(fl, ms),
                      "weight_read_floor_ms": round(668e6 / 6.3e12 * 1e3, 3)}), flush=True)
