#!/bin/bash
# usage (GPU box, repo root): tools/ab_gemm.sh "<env assignments>" ... -> encoder, 64 x 512 tokens, one line per setting (knobs build)
for rep in 1 2; do
for v in "$@"; do
  echo -n "$v  "
  env SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so $v python bench_configs.py --mode encode --batch 64 2>/dev/null | tail -1 | cut -c1-110
done
done
