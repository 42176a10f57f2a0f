#!/usr/bin/env python3
"""Tuning run of the ring-GEMM tile menu (encoder.hip: RING_MENU) at one token count: every admissible (shape, split-K) of
one GEMM type at a time, the other types on the chooser's default, whole-encoder time per configuration (knobs build,
hipGraphs off so that every forward re-reads the knob).  Prints one JSON line per configuration.
    SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so SQE_ENC_GRAPH=0 python tools/ring_tune.py --batch 64 --seq 32"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SQE_ENC_GRAPH", "0")
from bench_configs import random_bert_weights, timed
from semantic_query_engine_amd import Context
from semantic_query_engine_amd.encoder import BertEncoder

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--seq", type=int, default=32)
ap.add_argument("--iters", type=int, default=12)
args = ap.parse_args()
ctx = Context(0); dev = torch.device("cuda", 0)
enc = BertEncoder(ctx); enc.load_weights(random_bert_weights())
b, s = args.batch, args.seq
ids = torch.randint(1000, 30000, (b, s), device=dev, dtype=torch.int32)
lens = torch.full((b,), s, device=dev, dtype=torch.int32)
emb = torch.empty((b, 1024), device=dev)
torch.cuda.synchronize()
MENU = ["64x64", "128x64", "192x64", "256x64", "128x128", "192x128", "256x128"]

def run(tag):
    ms = timed(lambda: enc.encode_ids_device(ids.data_ptr(), lens.data_ptr(), b, s, emb.data_ptr()), ctx.synchronize, args.iters, 3)
    print(json.dumps({"tokens": b * s, "config": tag, "encode_ms": round(ms, 4)}), flush=True)
    return ms

base = run("chooser default")
os.environ["SQE_RING_XCD"] = "0"; run("chooser default, linear tile walk"); del os.environ["SQE_RING_XCD"]
for name, key, shapes, splits in (("QKV", "SQE_RING_FORCE_0_K1024", range(7), (1,)),
                                  ("FFN-up", "SQE_RING_FORCE_1_K1024", (0, 1, 3, 4, 6), (1,)),
                                  ("out-proj", "SQE_RING_FORCE_2_K1024", (0, 1, 3, 4, 6), (1, 2, 4)),
                                  ("FFN-down", "SQE_RING_FORCE_2_K4096", (0, 1, 3, 4, 6), (1, 2, 4))):
    for m in shapes:
        for sp in splits:
            os.environ[key] = f"{m}:{sp}"
            run(f"{name} {MENU[m]} split {sp}")
    del os.environ[key]
