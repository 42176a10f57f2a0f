#!/usr/bin/env python3
"""Scan throughput across vector dimensions (batch 1024 and 64): rows chosen so the bf16 copy is ~8 GB."""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from semantic_query_engine_amd import Context, VectorIndex

ctx = Context(0)
dev = torch.device("cuda", 0)
for d in (128, 384, 768, 1024, 2048, 4096):
    n = int(4e9 // d) // 256 * 256
    idx = VectorIndex(ctx, d)
    idx.reserve(n)
    blk = 1 << 19
    for lo in range(0, n, blk):
        m = min(blk, n - lo)
        x = torch.randn((m, d), device=dev)
        torch.cuda.synchronize()
        idx.add_device(x.data_ptr(), m)
        ctx.synchronize()
        del x
    for b in (1024, 64):
        q = torch.randn((b, d), device=dev)
        cos = torch.empty((b, 10), device=dev); ids = torch.empty((b, 10), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):
            idx.search_device(q.data_ptr(), b, 10, cos.data_ptr(), ids.data_ptr())
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            idx.search_device(q.data_ptr(), b, 10, cos.data_ptr(), ids.data_ptr())
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(json.dumps({"dim": d, "rows": n, "batch": b, "ms": round(ms, 3), "tflops": round(2.0 * n * d * b / ms / 1e9, 1),
                          "gbps": round(n * d * 2 / ms / 1e6, 1), "unc": ctx.stats()["uncertified"]}), flush=True)
    idx.close()
    torch.cuda.empty_cache()
