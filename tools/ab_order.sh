#!/bin/bash
# usage (GPU box, repo root): tools/ab_order.sh -> order of DMA pieces and operand reads in the memory phase of the ping-pong scan:
# SQE_DBG=128 every wave pieces first (r01), 256 every wave reads first, 512 staggered by wave pairs, 0 staggered by parity (default)
for rep in 1 2; do
for b in ${1:-1024 512 256}; do
  for d in 128 256 512 0; do bash tools/ab.sh "SQE_DBG=$d" 10000000 $b; done
done
done
