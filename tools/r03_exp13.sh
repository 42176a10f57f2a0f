#!/bin/bash
# usage (GPU box, repo root): tools/r03_exp13.sh -> where the encoder's ping-pong GEMM spends its phases: ablation builds
# (tools/build_gpp_ablate.sh 8 16 24 32 56 64 0s), encode of 64 x 512 tokens, two rounds
out=gpurun_out/exp13; mkdir -p $out
for rep in 1 2; do
for l in knobs gpp8 gpp16 gpp24 gpp32 gpp56 gpp64; do
  echo -n "$l  "
  env SQE_LIB=semantic_query_engine_amd/libsqe_$l.so python bench_configs.py --mode encode --batch 64 --no-cpu-baseline 2>$out/err_$l.txt | tail -1 | cut -c1-100
done
done
echo "== stamps"
env SQE_LIB=semantic_query_engine_amd/libsqe_gpp0s.so SQE_GEMM_DBG=4 python bench_configs.py --mode encode --batch 64 --no-cpu-baseline 2>&1 | grep "sqe dbg\|\"ms\"" | cut -c1-330
