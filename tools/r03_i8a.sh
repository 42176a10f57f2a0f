#!/bin/bash
# r03: first run of the int8 first-pass scan: its GPU tests (all but the 10 M one first), then bench int8 vs bf16 at batch 1024 / 256.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_i8a
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_i8_gpu.py -x -q -k "not full_size" > $out/tests.log 2>&1; rc=$?
tail -25 $out/tests.log
[ $rc -ne 0 ] && { echo "I8 TESTS FAILED rc=$rc"; exit 1; }
for b in 1024 256; do
  for mode in bf16 int8; do
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --batch $b --scan-mode $mode --no-cpu-baseline --no-gemm-ref --recall-queries 32 > $out/bench_${mode}_$b.json 2> $out/bench_${mode}_$b.err || { tail -5 $out/bench_${mode}_$b.err; echo "bench $mode $b failed"; exit 1; }
    python - <<PY
import json
d=json.load(open("$out/bench_${mode}_$b.json"))
print("$mode", $b, "qps", d["value"], "ms/step", d["ms_per_step"], d["stage_ms"], "recall", d["recall_at_10"], "unc", d["uncertified_queries_last_step"], d.get("int8_last_step"), "frac", d["roofline"]["frac"])
PY
  done
done
