"""Hand-run check (not a test): the same int8 search many times -- the number of collected keys is a function of every estimated
score against a fixed threshold, so it must not change between identical calls.  usage (GPU box): python tools/repeat_scan.py [rows] [repeats]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_query_engine_amd import SCAN_BF16_RESCORE, SCAN_INT8_RESCORE, Context, VectorIndex

D, K, ROWS, B = 1024, 10, int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000, 1024
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
ctx = Context(0)
g = torch.Generator(device=dev).manual_seed(9)
idx = VectorIndex(ctx, D)
idx.reserve(ROWS)
for lo in range(0, ROWS, 1 << 20):
    n = min(1 << 20, ROWS - lo)
    x = torch.randn((n, D), generator=g, device=dev)
    torch.cuda.synchronize()
    idx.add_device(x.data_ptr(), n)
    ctx.synchronize()
    del x
q = torch.randn((B, D), generator=g, device=dev)
cos = torch.empty((B, K), device=dev)
ids = torch.empty((B, K), dtype=torch.int64, device=dev)
for mode, name in ((SCAN_INT8_RESCORE, "int8"), (SCAN_BF16_RESCORE, "bf16")):
    idx.set_option("scan_mode", mode)
    seen, ref = {}, None
    for it in range(REP):
        ctx.stats_reset()
        idx.search_device(q.data_ptr(), B, K, cos.data_ptr(), ids.data_ptr())
        ctx.synchronize()
        st = ctx.stats()
        key = (st.get("i8_collected", 0), st.get("i8_rescored", 0), st["uncertified"])
        seen[key] = seen.get(key, 0) + 1
        if ref is None:
            ref = (cos.clone(), ids.clone())
        elif not (torch.equal(ref[0], cos) and torch.equal(ref[1], ids)):
            print(name, "run", it, "RESULT differs from the first run", flush=True)
    print(name, "(collected, rescored, uncertified) -> runs:", seen, flush=True)
idx.close()
