#!/bin/bash
# usage: tools/ab_lib.sh <library .so> "<env assignments>" rows batch  -> one line (A/B of two builds inside one gpurun call)
env SQE_LIB=$1 $2 python bench.py --steps 8 --warmup 3 --rows $3 --batch $4 --no-cpu-baseline --no-gemm-ref --recall-queries 32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1].split('/')[-1], sys.argv[2], sys.argv[3], sys.argv[4], d['stage_ms'], d['roofline']['mfma_tflops'], 'recall', d['recall_at_10'], 'qps', d['value'], 'unc', d.get('uncertified_queries_last_step'))" "$1" "$2" $3 $4
