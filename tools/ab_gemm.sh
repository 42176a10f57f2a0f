#!/bin/bash
# usage (GPU box, repo root): tools/ab_gemm.sh -> encoder, 64 x 512 tokens: ping-pong GEMM with every wave issuing DMA first (SQE_GEMM_STAGGER=0) vs staggered (1)
for rep in 1 2 3; do
for v in 0 1; do
  echo -n "SQE_GEMM_STAGGER=$v "
  env SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so SQE_GEMM_STAGGER=$v python bench_configs.py --mode encode --batch 64 2>/dev/null | tail -1 | cut -c1-120
done
done
