#!/bin/bash
# usage (GPU box, repo root): tools/ab_prev.sh [batches] -> libsqe_prev.so (a build of an earlier commit, made by hand) vs libsqe.so, interleaved
L=semantic_query_engine_amd
for rep in 1 2; do
for b in ${1:-1024 512 256}; do
  bash tools/ab_lib.sh $L/libsqe_prev.so "SQE_X=0" 10000000 $b
  bash tools/ab_lib.sh $L/libsqe.so "SQE_X=0" 10000000 $b
done
done
