#!/bin/bash
# usage (GPU box, repo root): tools/ab_defer.sh -> SQE_DBG=1024 (all four DMA pieces from the memory phase) vs 0 (one from the compute phase)
for rep in 1 2; do
for b in ${1:-1024 512 256}; do
  for d in 1024 0; do bash tools/ab.sh "SQE_DBG=$d" 10000000 $b; done
done
done
