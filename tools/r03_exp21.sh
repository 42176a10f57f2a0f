#!/bin/bash
# A/B inside one call: bf16 scan, one barrier per half-step (libsqe_knobs.so) vs a barrier after every phase (libsqe_2barpp.so:
# tools/build_variant.sh scan_pp 2barpp -DSQE_PP_TWO_BARRIERS); the search tests run on the new schedule first
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp21
mkdir -p $out
timeout -k 10 1000 python -m pytest tests/test_search_gpu.py tests/test_scan_variants_gpu.py tests/test_retrieval_gpu.py tests/test_i8_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
for r in 1 2 3; do for lib in libsqe_2barpp.so libsqe_knobs.so; do for b in 1024 512 256; do
  SCAN_MODE=bf16 tools/ab_lib.sh semantic_query_engine_amd/$lib "SQE_X=0" 10000000 $b | tee -a $out/ab.log
done; done; done
