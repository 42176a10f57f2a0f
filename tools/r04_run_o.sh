#!/bin/bash
out=gpurun_out/r04o; mkdir -p $out
{ timeout -k 10 500 python -m pytest tests/test_ivf_gpu.py tests/test_config5_gpu.py tests/test_group_gpu.py -m gpu -q -x > $out/tests.log 2>&1 || [ $? -eq 1 ]; } || exit 1
tail -3 $out/tests.log
grep -q passed $out/tests.log && ! grep -q failed $out/tests.log || { tail -60 $out/tests.log; exit 1; }
timeout -k 10 400 tools/r04_ivf_ab.sh $out ""
cp semantic_query_engine_amd/libsqe_knobs.so semantic_query_engine_amd/libsqe_noqueue.so
SQE_IVF_QUEUE=0 timeout -k 10 400 tools/r04_ivf_ab.sh $out noqueue
tools/r04_final.sh 2
