#!/bin/bash
# r03: tile-uniform int8 scales (scale on the thresholds, not on the accumulators): int8 + search tests, bench, PMC; IVF batch order probe
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp8
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_i8_gpu.py tests/test_search_gpu.py tests/test_group_gpu.py tests/test_persistence_gpu.py -x -q > $out/tests.log 2>&1; rc=$?
tail -6 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
for r in 1 2; do for b in 1024 256 64; do
  python bench.py --steps 10 --warmup 3 --batch $b --scan-mode int8 --no-second-leg --no-cpu-baseline --no-gemm-ref --recall-queries 16 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print($b, d['ms_per_step'], d['value'], d['stage_ms'], 'recall', d['recall_at_10'], 'unc', d['uncertified_queries_last_step'], d.get('int8_last_step'), d['roofline']['frac'])" | tee -a $out/bench.log
done; done
python tools/ivf_batch_probe.py --sequence 1024,1,1024,64,1024,256,1024 > $out/ivf_sequence.jsonl 2> $out/ivf_sequence.err; cat $out/ivf_sequence.jsonl
