"""What the appends of the int8 collect scan cost: the same 10 M x 1024 scan, batch 1024, top-1, with the default threshold (~2,400
keys per query) and with the threshold at the sample's best score and no anchor (~100 keys per query: 24 x fewer appended keys).
usage (GPU box): python tools/append_cost.py [rows]   -> one JSON line per setting (scan stage ms, keys collected)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_query_engine_amd import Context, VectorIndex

D, K, B = 1024, 1, 1024
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
ctx = Context(0)
idx = VectorIndex(ctx, D)
idx.reserve(ROWS)
g = torch.Generator(device=dev).manual_seed(3)
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
for name, opts in (("default", {}), ("few_keys", {"i8_sample_m": 1, "i8_key_budget": 1}), ("default_again", {"i8_sample_m": 20, "i8_key_budget": 6144})):
    for k_, v_ in opts.items():
        idx.set_option(k_, v_)
    for _ in range(3):
        idx.search_device(q.data_ptr(), B, K, cos.data_ptr(), ids.data_ptr())
    ctx.synchronize()
    ctx.stats_reset()
    ctx.set_profiling(True)
    for _ in range(10):
        idx.search_device(q.data_ptr(), B, K, cos.data_ptr(), ids.data_ptr())
    ctx.synchronize()
    st = ctx.stats()
    ctx.set_profiling(False)
    print(json.dumps({"setting": name, "scan_ms": round(st["scan_ms"] / max(st["scan_calls"], 1), 4), "scan_calls": st["scan_calls"],
                      "keys_per_query": round(st["i8_collected"] / B, 1), "uncertified": st["uncertified"]}), flush=True)
