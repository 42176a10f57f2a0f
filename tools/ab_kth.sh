#!/bin/bash
# usage (GPU box, repo root): tools/ab_kth.sh [batches] -> k-row bound from the minimum of the 16 group maxima (SQE_DBG=2048) vs their k-th largest (0)
for rep in 1 2; do
for b in ${1:-1024 512 256}; do
  for d in 2048 0; do bash tools/ab.sh "SQE_DBG=$d" 10000000 $b; done
done
done
