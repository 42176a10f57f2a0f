#!/bin/bash
# usage (GPU box, repo root): tools/ab_order.sh [batches] -> order of DMA pieces and operand reads in the memory phase of the
# ping-pong scan: SQE_DBG=128 every wave pieces first (r01), 256 every wave reads first, 2048 every wave interleaved,
# 4096 pairs: pieces first / interleaved, 0 pairs: pieces first / reads first (default)
for rep in 1 2; do
for b in ${1:-1024 256}; do
  for d in 128 256 2048 4096 0; do bash tools/ab.sh "SQE_DBG=$d" 10000000 $b; done
done
done
