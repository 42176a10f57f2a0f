#!/usr/bin/env python3
"""Single-query latency as the reference issues it (one /ask = one embed_query + one cache scan + one k-NN):
BERT-large encode of 1 x S tokens (random weights), cache get over 1000 entries, search of a 32,717-row index."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_configs import random_bert_weights, timed
from semantic_query_engine_amd import Context, VectorIndex
from semantic_query_engine_amd.encoder import BertEncoder
ctx = Context(0); dev = torch.device("cuda", 0)
enc = BertEncoder(ctx); enc.load_weights(random_bert_weights())
idx = VectorIndex(ctx, 1024)
x = torch.randn((32717, 1024), device=dev); torch.cuda.synchronize(); idx.add_device(x.data_ptr(), 32717); ctx.synchronize()
out = {"graph": os.environ.get("SQE_ENC_GRAPH", "1"), "cases": []}
for s in (16, 32, 64):
    ids = torch.randint(1000, 30000, (1, s), device=dev, dtype=torch.int32); lens = torch.full((1,), s, device=dev, dtype=torch.int32)
    emb = torch.empty((1, 1024), device=dev); cos = torch.empty((1, 3), device=dev); nid = torch.empty((1, 3), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    enc_ms = timed(lambda: (enc.encode_ids_device(ids.data_ptr(), lens.data_ptr(), 1, s, emb.data_ptr()), ctx.synchronize()), lambda: None, 30, 5)
    srch_ms = timed(lambda: (idx.search_device(emb.data_ptr(), 1, 3, cos.data_ptr(), nid.data_ptr()), ctx.synchronize()), lambda: None, 30, 5)
    out["cases"].append({"tokens": s, "encode_ms": round(enc_ms, 3), "search_ms": round(srch_ms, 3)})
print(json.dumps(out))
