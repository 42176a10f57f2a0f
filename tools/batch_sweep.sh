#!/bin/bash
# usage: tools/batch_sweep.sh  -> one line per batch size at 10M x 1024 (queries/s, scan ms, roofline of the binding side)
for b in 1 8 64 128 256 512 1024 2048; do
  python bench.py --scan-mode ${SCAN_MODE:-bf16} --no-second-leg --steps 20 --warmup 3 --batch $b --no-cpu-baseline --no-gemm-ref --recall-queries 16 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(json.dumps({'batch': d['config']['batch'], 'queries_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'scan_ms': d['stage_ms']['scan'], 'select_ms': d['stage_ms']['select_rescore'], 'bound': r['bound'], 'frac': r['frac'], 'hbm_gbps': r['hbm_gbps'], 'mfma_tflops': r['mfma_tflops'], 'kernel': r['kernel'], 'recall_at_10': d['recall_at_10']}))"
done
