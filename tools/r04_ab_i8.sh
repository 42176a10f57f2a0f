#!/bin/bash
# A/B inside one gpurun call: schedule variants of the int8 ping-pong scan (scan_i8.hip: I8V; libraries from tools/build_variant.sh
# scan_i8 i8v<bits> -DSQE_I8_VARIANT=<bits>) against the knobs build, 10 M x 1024 rows.  usage: tools/r04_ab_i8.sh <out dir> <tags...>
set -o pipefail
export TMPDIR=/tmp
out=$1; shift
mkdir -p $out
for r in 1 2; do for t in "$@"; do for b in ${BATCHES:-1024 256}; do
  SCAN_MODE=int8 tools/ab_lib.sh semantic_query_engine_amd/libsqe_$t.so "X=0" 10000000 $b | tee -a $out/ab.log
done; done; done
