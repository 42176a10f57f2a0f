#!/bin/bash
# usage (GPU box, repo root): tools/r03_exp17.sh tags... -> phase stamps (cycles) beside wall time for stamped ablation builds
for t in "$@"; do
  echo "== $t"
  env SQE_LIB=semantic_query_engine_amd/libsqe_gpp$t.so SQE_GEMM_DBG=4 python bench_configs.py --mode encode --batch 64 --no-cpu-baseline 2>&1 | grep "sqe dbg\|\"ms\"" | cut -c1-330
done
