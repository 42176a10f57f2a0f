#!/usr/bin/env python3
"""IVF search of BASELINE config 5 (10 M x 1024 clustered rows, nlist 4096, nprobe 32) at ONE batch size, a few steps:
run it under `rocprofv3 --kernel-trace --stats` to see where a batch size spends its time (r03: batch 64 took 9.5 ms
against 4.3 ms at batch 1024).    python tools/ivf_batch_probe.py --batch 64 [--rows 10000000]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_configs import BLOCK, D, build_clustered, timed
from semantic_query_engine_amd import INDEX_IVF_FLAT, Context

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--sequence", default="", help="comma-separated batch sizes timed one after the other on the same index (r03: is a batch-1024 search slower after smaller ones?)")
args = ap.parse_args()
ctx = Context(0); dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(99)
centres = torch.randn((4096, D), generator=g, device=dev)
ivf = build_clustered(ctx, args.rows, dev, centres, INDEX_IVF_FLAT, 4096)
gs = torch.Generator(device=dev).manual_seed(1000)
n_s = min(BLOCK, args.rows)
xs = centres[torch.randint(0, 4096, (n_s,), generator=gs, device=dev)] + 0.3 * torch.randn((n_s, D), generator=gs, device=dev)
torch.cuda.synchronize()
ivf.train_device(xs.data_ptr(), n_s, iters=20, seed=0); ctx.synchronize(); del xs
gq = torch.Generator(device=dev).manual_seed(5)
b = args.batch
q = centres[torch.randint(0, 4096, (1024,), generator=gq, device=dev)] + 0.3 * torch.randn((1024, D), generator=gq, device=dev)
ci = torch.empty((1024, 10), device=dev); ji = torch.empty((1024, 10), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for bb in ([int(v) for v in args.sequence.split(",")] if args.sequence else [b]):
    ms = timed(lambda: ivf.search_device(q.data_ptr(), bb, 10, ci.data_ptr(), ji.data_ptr(), nprobe=32), ctx.synchronize, args.iters)
    print(json.dumps({"batch": bb, "ivf_ms": round(ms, 4)}), flush=True)
