#!/usr/bin/env python3
"""Practical MFMA ceiling on this box: time torch.matmul (hipBLASLt) on the scan's GEMM shape,
queries [1024 x 1024] x DB block [1M x 1024]^T in bf16, ten blocks = the 10M-row headline scan
without any top-k work.  Random and all-zero operands (the clock the chip holds depends on the data).
Prints one JSON line per case."""
import json
import sys

import torch

dev = torch.device("cuda", 0)
B, N, D, BLOCKS = 1024, 1 << 20, 1024, 10
out = torch.empty((B, N), dtype=torch.bfloat16, device=dev)
for name in ("random", "zeros"):
    if name == "random":
        q = torch.randn((B, D), device=dev).bfloat16()
        db = torch.randn((N, D), device=dev).bfloat16()
    else:
        q = torch.zeros((B, D), device=dev, dtype=torch.bfloat16)
        db = torch.zeros((N, D), device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(q, db.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps * BLOCKS):
        torch.matmul(q, db.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"case": name, "ms_per_10M_rows": round(ms, 3),
                      "tflops": round(2.0 * B * N * D * BLOCKS / ms / 1e9, 1)}), flush=True)
