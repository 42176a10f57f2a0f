#!/bin/bash
# (uses the knobs build: make -C semantic_query_engine_amd/csrc KNOBS=1)
# usage: tools/ab.sh "<env assignments>" rows batch   -> one line: scan ms, TFLOP/s, recall
env SQE_LIB=$(dirname $0)/../semantic_query_engine_amd/libsqe_knobs.so $1 python bench.py --scan-mode ${SCAN_MODE:-bf16} --no-second-leg --steps 5 --warmup 2 --rows $2 --batch $3 --no-cpu-baseline --no-gemm-ref --recall-queries 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], sys.argv[3], d['stage_ms'], d['roofline']['mfma_tflops'], 'recall', d['recall_at_10'], d['value'], 'unc', d.get('uncertified_queries_last_step'))" "$1" $2 $3
