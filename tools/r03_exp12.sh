#!/bin/bash
# int8 scan: append barrier on / off at 2, 3 and 4 query blocks (knobs build), A/B in one call
export TMPDIR=/tmp
out=gpurun_out/r03_exp12
mkdir -p $out
for r in 1 2; do for b in 512 768 1024; do for sync in 0 1; do
  SCAN_MODE=int8 tools/ab.sh "SQE_I8_SYNC=$sync" 10000000 $b | tee -a $out/ab.log
done; done; done
