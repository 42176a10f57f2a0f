#!/bin/bash
# usage (GPU box, repo root): tools/ab_enc.sh -> encoder 64 x 512 tokens, libsqe_prev.so (an earlier commit, built by hand) vs libsqe.so, interleaved
for rep in 1 2 3; do
for l in libsqe_prev.so libsqe.so; do
  echo -n "$l  "
  env SQE_LIB=semantic_query_engine_amd/$l python bench_configs.py --mode encode --batch 64 2>/dev/null | tail -1 | cut -c1-110
done
done
