#!/bin/bash
out=gpurun_out/r04n; mkdir -p $out
{ timeout -k 10 900 python -m pytest tests/test_i8_exact_gpu.py tests/test_i8_gpu.py tests/test_search_gpu.py tests/test_ivf_gpu.py tests/test_config5_gpu.py -m gpu -q -x > $out/tests.log 2>&1 || [ $? -eq 1 ]; } || exit 1
tail -3 $out/tests.log
grep -q passed $out/tests.log && ! grep -q failed $out/tests.log || { tail -60 $out/tests.log; exit 1; }
for r in 1 2; do for b in 1 64 128; do for st in 1 0; do
  SCAN_MODE=int8 tools/ab_lib.sh semantic_query_engine_amd/libsqe_knobs.so "SQE_I8_STAGED=$st" 10000000 $b | tee -a $out/ab_stream_small.log
done; done; done
timeout -k 10 400 tools/r04_ivf_ab.sh $out ""
