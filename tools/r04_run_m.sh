#!/bin/bash
out=gpurun_out/r04m; mkdir -p $out
{ timeout -k 10 500 python -m pytest tests/test_ivf_gpu.py tests/test_config5_gpu.py -m gpu -q -x > $out/tests.log 2>&1 || [ $? -eq 1 ]; } || exit 1
tail -3 $out/tests.log
grep -q passed $out/tests.log && ! grep -q failed $out/tests.log || { tail -40 $out/tests.log; exit 1; }
timeout -k 10 400 tools/r04_ivf_ab.sh $out ""
for b in 512; do for dm in 1 2 1 2; do
  SCAN_MODE=int8 tools/ab_lib.sh semantic_query_engine_amd/libsqe_knobs.so "SQE_I8_DEEP_MAX=$dm" 10000000 $b | tee -a $out/ab_deep_512.log
done; done
