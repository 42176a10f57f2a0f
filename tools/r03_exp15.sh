#!/bin/bash
# usage (GPU box, repo root): tools/r03_exp15.sh tags... -> encode of 64 x 512 tokens on libsqe_gpp<tag>.so, three rounds
out=gpurun_out/exp15; mkdir -p $out
for rep in 1 2 3; do
for t in "$@"; do
  echo -n "$t  "
  env SQE_LIB=semantic_query_engine_amd/libsqe_gpp$t.so python bench_configs.py --mode encode --batch 64 --no-cpu-baseline 2>$out/err_$t.txt | tail -1 | cut -c1-100
done
done
