#!/bin/bash
# A/B inside one call: int8 scan with the append phase behind its own barrier (libsqe_prev.so) vs appends on the way into the next phase
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp11
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_i8_gpu.py -x -q -k "not full_size" > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
for r in 1 2 3; do for lib in libsqe_prev.so libsqe.so; do for b in 1024 256; do
  SCAN_MODE=int8 tools/ab_lib.sh semantic_query_engine_amd/$lib "SQE_X=0" 10000000 $b | tee -a $out/ab.log
done; done; done
