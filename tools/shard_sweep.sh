#!/bin/bash
# usage (GPU box, repo root): tools/shard_sweep.sh > gpurun_out/shard_sweep.json   (the int8 first pass: bench.py's default)
# One MI355X running the per-rank work of an N-way row shard of the 10M x 1024 index at batch 1024 (the all-gather of
# [B,k] and the merge kernel are not included): what strong scaling can be expected where no multi-GPU box is at hand.
python - <<'PY'
import json, subprocess, sys
cases = []
for n in (1, 2, 4, 8):
    rows = 10_000_000 // n
    out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--rows", str(rows), "--batch", "1024",
                          "--no-cpu-baseline", "--no-gemm-ref", "--recall-queries", "16", "--no-second-leg", "--no-clustered-leg"], capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    cases.append({"rows_per_shard": rows, "equivalent_gpus": n, "queries_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                  "stage_ms": d["stage_ms"], "recall_at_10": d["recall_at_10"], "mfma_tflops": d["roofline"]["mfma_tflops"]})
print(json.dumps({"what": "one MI355X running the per-rank work of an N-way row shard of the 10M x 1024 index, batch 1024 "
                          "(python bench.py --no-second-leg --no-clustered-leg --rows R): the all-gather of [B,k] and the merge kernel are not included", "cases": cases}, indent=1))
PY
