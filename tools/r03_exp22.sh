#!/bin/bash
# A/B inside one call (knobs build): variants of the int8 scan's one-barrier schedule, SQE_I8_DBG = 0 / 1 / 2 / 4 / 8 / 16 (scan_i8.hip)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp22
mkdir -p $out
for r in 1 2; do for v in ${@:-0 1 2 4 8 16}; do for b in 1024 512; do
  SCAN_MODE=int8 tools/ab_lib.sh semantic_query_engine_amd/libsqe_knobs.so "SQE_I8_DBG=$v" 10000000 $b | tee -a $out/ab.log
done; done; done
