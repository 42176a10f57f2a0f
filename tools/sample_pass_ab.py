"""Threshold pass of the int8 first pass: bf16 scan + fp32 re-score of the sample (i8_sample_int8 = 0) against the int8 sample scan
+ order statistic (1).  usage (GPU box): python tools/sample_pass_ab.py [rows ...]  -> one JSON line per (rows, batch, form)"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_query_engine_amd import Context, VectorIndex

D, K = 1024, 10
dev = torch.device("cuda", 0)
ctx = Context(0)
g = torch.Generator(device=dev).manual_seed(5)
for rows in [int(a) for a in sys.argv[1:]] or [1250000, 10000000]:
    idx = VectorIndex(ctx, D)
    idx.reserve(rows)
    for lo in range(0, rows, 1 << 20):
        n = min(1 << 20, rows - lo)
        x = torch.randn((n, D), generator=g, device=dev)
        torch.cuda.synchronize()
        idx.add_device(x.data_ptr(), n)
        ctx.synchronize()
        del x
    for b in (1024, 256, 64, 1):
        q = torch.randn((b, D), generator=g, device=dev)
        ref = None
        for form in (0, 1, 0, 1):
            idx.set_option("i8_sample_int8", form)
            cos = torch.empty((b, K), device=dev)
            ids = torch.empty((b, K), dtype=torch.int64, device=dev)
            for it in range(3):
                idx.search_device(q.data_ptr(), b, K, cos.data_ptr(), ids.data_ptr())
            ctx.synchronize()
            ctx.stats_reset()
            ctx.set_profiling(True)
            n_it = 20
            for it in range(n_it):
                idx.search_device(q.data_ptr(), b, K, cos.data_ptr(), ids.data_ptr())
            ctx.synchronize()
            st = ctx.stats()
            ctx.set_profiling(False)
            if ref is None:
                ref = ids.clone()
            same = bool(torch.equal(ref, ids))
            print(json.dumps({"rows": rows, "batch": b, "sample_int8": form, "sample_ms": round(st["sample_ms"] / n_it, 4),
                              "scan_ms": round(st["scan_ms"] / n_it, 4), "select_ms": round(st["select_ms"] / n_it, 4),
                              "collected_per_query": round(st["i8_collected"] / b), "uncertified": st["uncertified"], "same_ids": same}), flush=True)
    idx.close()
