"""Fixed cost of one int8 / bf16 scan launch: stage times of a search over indexes of 64 k .. 2.5 M rows (batch 1024 and 256).
usage (GPU box): python tools/scan_fixed_cost.py  -> one JSON line per (rows, batch, mode)"""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_query_engine_amd import SCAN_BF16_RESCORE, SCAN_INT8_RESCORE, Context, VectorIndex

D, K = 1024, 10
dev = torch.device("cuda", 0)
ctx = Context(0)
g = torch.Generator(device=dev).manual_seed(5)
for rows in (65536, 262144, 655360, 1310720, 2621440):
    idx = VectorIndex(ctx, D)
    idx.set_option("i8_min_rows", 0)
    x = torch.randn((rows, D), generator=g, device=dev)
    torch.cuda.synchronize()
    idx.add_device(x.data_ptr(), rows)
    ctx.synchronize()
    del x
    for b in (1024, 256):
        q = torch.randn((b, D), generator=g, device=dev)
        cos = torch.empty((b, K), device=dev)
        ids = torch.empty((b, K), dtype=torch.int64, device=dev)
        for mode, name in ((SCAN_INT8_RESCORE, "int8"), (SCAN_BF16_RESCORE, "bf16")):
            idx.set_option("scan_mode", mode)
            for it in range(3):
                idx.search_device(q.data_ptr(), b, K, cos.data_ptr(), ids.data_ptr())
            ctx.synchronize()
            ctx.stats_reset()
            ctx.set_profiling(True)
            n = 20
            for it in range(n):
                idx.search_device(q.data_ptr(), b, K, cos.data_ptr(), ids.data_ptr())
            ctx.synchronize()
            st = ctx.stats()
            ctx.set_profiling(False)
            print(json.dumps({"rows": rows, "batch": b, "mode": name, "scan_ms": round(st["scan_ms"] / n, 4),
                              "sample_ms": round(st.get("sample_ms", 0.0) / n, 4), "select_ms": round(st["select_ms"] / n, 4),
                              "prep_ms": round(st["prep_ms"] / n, 4), "scan_calls": st["scan_calls"]}), flush=True)
    idx.close()
