#!/bin/bash
# select_i8 (bf16 middle stage; r03c: no sorts): int8 + search + group tests, then the stage times at batch 1024 / 512 / 256 (compare with
# select_rescore 0.40 / 0.25 / 0.146 ms), then batch 256 with G0 issuing before it waits (SQE_I8_DBG=8)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp24
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_i8_gpu.py tests/test_search_gpu.py tests/test_group_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
for r in 1 2; do for b in 1024 512 256; do
  SCAN_MODE=int8 tools/ab_lib.sh semantic_query_engine_amd/libsqe_knobs.so "SQE_X=0" 10000000 $b | tee -a $out/ab.log
done; done
for r in 1 2; do for v in 0 8; do
  SCAN_MODE=int8 tools/ab_lib.sh semantic_query_engine_amd/libsqe_knobs.so "SQE_I8_DBG=$v" 10000000 256 | tee -a $out/ab.log
done; done
