#!/usr/bin/env python3
"""A/B of an encoder knob at the reference's shapes (knobs build): run it twice with SQE_LIB=.../libsqe_knobs.so, the knob unset and set
(e.g. SQE_ENC_LN_FOLD=0); prints the latency of a few (batch, tokens) shapes and a hash of the embeddings -- the two runs must print the
same hashes when the knob only changes WHERE something is computed."""
import hashlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_configs import random_bert_weights, timed
from semantic_query_engine_amd import Context
from semantic_query_engine_amd.encoder import BertEncoder
ctx = Context(0); dev = torch.device("cuda", 0)
enc = BertEncoder(ctx); enc.load_weights(random_bert_weights())
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("SQE_ENC")}, "cases": []}
g = torch.Generator(device="cpu"); g.manual_seed(11)
for b, s in ((1, 16), (1, 32), (4, 16), (1, 64), (3, 21)):
    ids = torch.randint(1000, 30000, (b, s), generator=g, dtype=torch.int32).to(dev); lens = torch.full((b,), s, device=dev, dtype=torch.int32)
    emb = torch.empty((b, 1024), device=dev)
    torch.cuda.synchronize()
    ms = timed(lambda: (enc.encode_ids_device(ids.data_ptr(), lens.data_ptr(), b, s, emb.data_ptr()), ctx.synchronize()), lambda: None, 50, 5)
    out["cases"].append({"batch": b, "tokens": s, "encode_ms": round(ms, 4), "sha": hashlib.sha1(emb.cpu().numpy().tobytes()).hexdigest()[:12]})
print(json.dumps(out))
