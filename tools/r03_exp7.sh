#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp7
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_encoder_gpu.py tests/test_config3_gpu.py tests/test_config1_gpu.py tests/test_retrieval_gpu.py -x -q > $out/tests.log 2>&1; rc=$?
tail -5 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
python tools/enc_small.py > $out/enc_small.jsonl 2> $out/enc_small.err; cut -c1-110 $out/enc_small.jsonl
for sm in 50,32 100,16 100,32 50,16 25,64 200,16; do
  for r in 1 2; do
  python bench.py --steps 10 --warmup 3 --batch 1024 --scan-mode int8 --no-second-leg --i8-sample $sm --no-cpu-baseline --no-gemm-ref --recall-queries 16 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$sm', d['ms_per_step'], d['value'], d['stage_ms'], 'recall', d['recall_at_10'], 'unc', d['uncertified_queries_last_step'], d.get('int8_last_step'))" | tee -a $out/i8_sample_sweep.log
  done
done
