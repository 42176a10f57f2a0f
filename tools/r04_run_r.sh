#!/bin/bash
out=gpurun_out/r04r; mkdir -p $out
{ timeout -k 10 700 python -m pytest tests/test_encoder_gpu.py tests/test_config3_gpu.py tests/test_config1_gpu.py -m gpu -q -x > $out/tests.log 2>&1 || [ $? -eq 1 ]; } || exit 1
tail -3 $out/tests.log
grep -q passed $out/tests.log && ! grep -q failed $out/tests.log || { tail -60 $out/tests.log; exit 1; }
for i in 1 2; do
  SQE_LIB=$(pwd)/semantic_query_engine_amd/libsqe_knobs.so python tools/enc_knob_ab.py 2>/dev/null | tee -a $out/enc_ln_fold_ab.jsonl
  SQE_ENC_LN_FOLD=0 SQE_LIB=$(pwd)/semantic_query_engine_amd/libsqe_knobs.so python tools/enc_knob_ab.py 2>/dev/null | tee -a $out/enc_ln_fold_ab.jsonl
done
python tools/latency_b1.py 2>/dev/null | tee $out/latency_b1.json
