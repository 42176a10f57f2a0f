#!/bin/bash
# usage (GPU box, repo root): tools/r03_exp14.sh [tags...] -> the ping-pong GEMM with 0-3 of a wave's four DMA pieces issued from
# its compute phase (tools/build_gpp_ablate.sh d0 d1 d2 d3): the large-batch encoder test on each library, then encode of 64 x 512
# tokens, three rounds
out=gpurun_out/exp14; mkdir -p $out
tags=${@:-d0 d1 d2 d3}
for t in $tags; do
  echo -n "$t test: "
  env SQE_LIB=semantic_query_engine_amd/libsqe_gpp$t.so timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py -q -m gpu -k "large_batch or full_size" 2>&1 | tail -1
done
for rep in 1 2 3; do
for t in $tags; do
  echo -n "$t  "
  env SQE_LIB=semantic_query_engine_amd/libsqe_gpp$t.so python bench_configs.py --mode encode --batch 64 --no-cpu-baseline 2>$out/err_$t.txt | tail -1 | cut -c1-100
done
done
