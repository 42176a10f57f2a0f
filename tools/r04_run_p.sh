#!/bin/bash
out=gpurun_out/r04p; mkdir -p $out
{ timeout -k 10 500 python -m pytest tests/test_ivf_gpu.py tests/test_config5_gpu.py -m gpu -q -x > $out/tests.log 2>&1 || [ $? -eq 1 ]; } || exit 1
tail -3 $out/tests.log
grep -q passed $out/tests.log && ! grep -q failed $out/tests.log || { tail -60 $out/tests.log; exit 1; }
timeout -k 10 600 tools/r04_ivf_ab.sh $out "" sel1 sel2 sel4 sel7
