#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r04s; mkdir -p $out
{ timeout -k 10 500 python -m pytest tests/test_ivf_gpu.py tests/test_config5_gpu.py -m gpu -q -x > $out/tests.log 2>&1 || [ $? -eq 1 ]; } || exit 1
tail -2 $out/tests.log
grep -q passed $out/tests.log && ! grep -q failed $out/tests.log || { tail -60 $out/tests.log; exit 1; }
rocprofv3 --kernel-trace --output-format csv -d $out/b1_prof -- python3 tools/latency_b1.py > $out/latency_under_prof.json 2> $out/b1_prof.err
python3 tools/trace_tail.py $out/b1_prof 24 | tee $out/search_32k_trace.txt; rm -rf $out/b1_prof
