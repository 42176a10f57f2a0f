#!/bin/bash
# one gpurun call of r04: IVF tests + stream-kernel A/B, int8 scan stamps, append cost, encoder one-barrier A/B + repeat check
out=gpurun_out/r04i; mkdir -p $out
{ timeout -k 10 400 python -m pytest tests/test_ivf_gpu.py tests/test_config5_gpu.py tests/test_group_gpu.py -m gpu -q -x > $out/tests.log 2>&1 || [ $? -eq 1 ]; } || exit 1
tail -3 $out/tests.log
timeout -k 10 400 tools/r04_ivf_ab.sh $out "" st_NO_STORE
for rows in 10000000 1250000; do
  SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so SQE_I8_STAMPS=1 timeout -k 10 200 python bench.py --scan-mode int8 --no-second-leg --no-clustered-leg --steps 2 --warmup 1 --rows $rows --no-cpu-baseline --no-gemm-ref --recall-queries 16 > $out/stamps_$rows.json 2> $out/stamps_$rows.txt
  grep "sqe i8 stamps" $out/stamps_$rows.txt | tail -24
done
timeout -k 10 200 python tools/append_cost.py 2>&1 | tee $out/append_cost.jsonl
echo ENC
for l in libsqe.so libsqe_gpp512.so libsqe.so libsqe_gpp512.so; do
  echo -n "$l "; SQE_LIB=semantic_query_engine_amd/$l timeout -k 10 120 python bench_configs.py --mode encode --batch 64 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120
done | tee $out/enc_ab.log
SQE_LIB=semantic_query_engine_amd/libsqe_gpp512.so timeout -k 10 200 python tools/repeat_enc.py 512 12 2>&1 | tail -3 | tee $out/repeat_enc_512.log
SQE_LIB=semantic_query_engine_amd/libsqe_gpp512.so timeout -k 10 200 python tools/repeat_enc.py 128 40 2>&1 | tail -3 | tee -a $out/repeat_enc_512.log
