#!/bin/bash
# stage times of one shard of a strong-scaling run: 10 M rows over 1 / 2 / 4 / 8 GPUs = 10 / 5 / 2.5 / 1.25 M rows per shard, batch 1024
for rows in 10000000 5000000 2500000 1250000; do for mode in int8 bf16; do
  python bench.py --steps 20 --warmup 3 --rows $rows --batch 1024 --scan-mode $mode --no-second-leg --no-cpu-baseline --no-gemm-ref --recall-queries 16 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print(json.dumps({'rows':$rows,'mode':'$mode','ms_per_step':d['ms_per_step'],'qps':d['value'],'stage_ms':d['stage_ms'],'unc':d['uncertified_queries_last_step']}))"
done; done
