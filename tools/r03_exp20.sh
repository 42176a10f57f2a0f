#!/bin/bash
# A/B inside one call: int8 scan, one barrier per half-step (libsqe_knobs.so) vs a barrier after every phase (libsqe_2bar.so:
# tools/build_variant.sh scan_i8 2bar -DSQE_I8_TWO_BARRIERS); the int8 tests run on the new schedule first
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp20
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_i8_gpu.py -x -q > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
for r in 1 2 3; do for lib in libsqe_2bar.so libsqe_knobs.so; do for b in 1024 512 256; do
  SCAN_MODE=int8 tools/ab_lib.sh semantic_query_engine_amd/$lib "SQE_X=0" 10000000 $b | tee -a $out/ab.log
done; done; done
