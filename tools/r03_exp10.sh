#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp10
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_ivf_gpu.py tests/test_config5_gpu.py tests/test_group_gpu.py tests/test_persistence_gpu.py tests/test_shim_gpu.py -x -q > $out/tests.log 2>&1; rc=$?
tail -12 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so python tools/ivf_batch_probe.py --sequence 1024,1,8,64,256,1024 > $out/ivf_int8.jsonl 2> $out/ivf_int8.err; cat $out/ivf_int8.jsonl
SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so SQE_IVF_I8=0 python tools/ivf_batch_probe.py --sequence 1024,1,8,64,256,1024 > $out/ivf_bf16.jsonl 2> $out/ivf_bf16.err; cat $out/ivf_bf16.jsonl
python bench_configs.py --mode ivf 2> $out/ivf.err | tail -1 > $out/cfg_ivf.json; cut -c1-420 $out/cfg_ivf.json
