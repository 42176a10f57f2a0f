#!/bin/bash
# one gpurun call of r04: the int8 exact + i8 + encoder tests on the new append path / one-barrier GEMM, then A/B of the append path
out=gpurun_out/r04j; mkdir -p $out
{ timeout -k 10 800 python -m pytest tests/test_i8_exact_gpu.py tests/test_i8_gpu.py tests/test_encoder_gpu.py tests/test_config3_gpu.py -m gpu -q -x > $out/tests.log 2>&1 || [ $? -eq 1 ]; } || exit 1
tail -3 $out/tests.log
for r in 1 2; do for rows in 10000000 1250000; do for t in prevappend knobs; do for b in 1024 256; do
  SCAN_MODE=int8 tools/ab_lib.sh semantic_query_engine_amd/libsqe_$t.so "X=0" $rows $b | tee -a $out/ab.log
done; done; done; done
