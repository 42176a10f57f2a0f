#!/bin/bash
# usage (GPU box, repo root): tools/ab_bound.sh  -> k-row bound on (default) vs off (SQE_DBG=64), interleaved, knobs build
for rep in 1 2; do
for b in 1024 512 256; do
  bash tools/ab.sh "SQE_DBG=64" 10000000 $b
  bash tools/ab.sh "SQE_DBG=0" 10000000 $b
done
done
bash tools/ab.sh "SQE_DBG=16" 10000000 1024
bash tools/ab.sh "SQE_DBG=4" 10000000 1024
