#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp9
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_group_gpu.py tests/test_threads_gpu.py tests/test_sharded_gpu.py tests/test_i8_gpu.py -x -q -k "not full_size" > $out/tests.log 2>&1; rc=$?
tail -6 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
for mode in int8 bf16; do
  SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so SQE_GROUP_SERIAL=1 python tools/group_host_cost.py --scan-mode $mode >> $out/group_host_cost.jsonl 2>> $out/group_host_cost.err
  SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so python tools/group_host_cost.py --scan-mode $mode >> $out/group_host_cost.jsonl 2>> $out/group_host_cost.err
done
cat $out/group_host_cost.jsonl
