#!/usr/bin/env python3
"""r02 verdict item 8: what the ONE host thread of a device group spends enqueueing a search.  P logical shards of device 0
(the one GPU a test box has), 10 M rows in total, batch 1024: host time of sqe_index_search_device without synchronising
(= enqueue only: query copies, ~10 launches per shard, exchange, merge) against the GPU time of one shard's pipeline
(what a real P-GPU group overlaps across devices).  Prints one JSON line."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_query_engine_amd import EXCHANGE_COPY, Context, VectorIndex

ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--scan-mode", choices=["bf16", "int8"], default="bf16")
args = ap.parse_args()
P, n, b, D = args.shards, args.rows, args.batch, 1024
dev = torch.device("cuda", 0)
ctx = Context(devices=[0] * P, exchange=EXCHANGE_COPY)
idx = VectorIndex(ctx, D)
idx.reserve(n)
for blk in range((n + (1 << 20) - 1) >> 20):
    rows = min(1 << 20, n - (blk << 20))
    g = torch.Generator(device=dev).manual_seed(1000 + blk)
    x = torch.randn((rows, D), generator=g, device=dev)
    torch.cuda.synchronize()
    idx.add_device(x.data_ptr(), rows)
    ctx.synchronize()
    del x
from semantic_query_engine_amd import SCAN_BF16_RESCORE, SCAN_INT8_RESCORE
idx.set_option("scan_mode", SCAN_INT8_RESCORE if args.scan_mode == "int8" else SCAN_BF16_RESCORE)
g = torch.Generator(device=dev).manual_seed(12345)
q = torch.randn((b, D), generator=g, device=dev)
cos = torch.empty((b, 10), device=dev); ids = torch.empty((b, 10), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for _ in range(3):
    idx.search_device(q.data_ptr(), b, 10, cos.data_ptr(), ids.data_ptr())
ctx.synchronize()
enq, tot = [], []
for _ in range(args.iters):
    t0 = time.perf_counter()
    idx.search_device(q.data_ptr(), b, 10, cos.data_ptr(), ids.data_ptr())
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
enq.sort(); tot.sort()
shard_step = tot[len(tot) // 2] / P            # the P shards of this one GPU run one after the other
print(json.dumps({"shards": P, "rows": n, "rows_per_shard": n // P, "batch": b, "scan_mode": args.scan_mode,
                  "host_enqueue_ms_median": round(enq[len(enq) // 2], 4), "host_enqueue_ms_min": round(enq[0], 4),
                  "wall_ms_median_all_shards_on_one_gpu": round(tot[len(tot) // 2], 4),
                  "ms_per_shard_step": round(shard_step, 4),
                  "host_enqueue_over_one_shard_step": round(enq[len(enq) // 2] / shard_step, 4),
                  "enqueue": "serial (r02)" if os.environ.get("SQE_GROUP_SERIAL") == "1" else "one worker thread per member",
                  "note": "host_enqueue = wall time of sqe_index_search_device up to its return (nothing is synchronised); on P real GPUs "
                          "the shards' GPU time overlaps and the host enqueue time is the serial part"}))
