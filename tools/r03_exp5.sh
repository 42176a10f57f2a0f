#!/bin/bash
# r03: int8 small-batch kernels + ring-GEMM tile menu: their GPU tests, encoder timings at the reference's shapes, batch sweep in both scan modes
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp5
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_i8_gpu.py tests/test_encoder_gpu.py tests/test_config3_gpu.py tests/test_config1_gpu.py tests/test_ivf_gpu.py tests/test_config5_gpu.py -x -q -k "not full_size" > $out/tests.log 2>&1; rc=$?
tail -15 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
python tools/enc_small.py > $out/enc_small.jsonl 2> $out/enc_small.err; cat $out/enc_small.jsonl
SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so SQE_ENC_RING=0 python tools/enc_small.py --cases 64x16,64x32,64x128 > $out/enc_small_r02tile.jsonl 2> $out/enc_small_r02tile.err; cat $out/enc_small_r02tile.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $out/enc_prof -- python3 tools/enc_small.py --cases 64x32 --iters 20 > /dev/null 2> $out/enc_prof.err
find $out/enc_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/enc_64x32_kernel_stats.csv
rm -rf $out/enc_prof
head -8 $out/enc_64x32_kernel_stats.csv | cut -c1-200
for mode in int8 bf16; do
  for b in 1 8 64 128 256 512 1024 2048; do
    python bench.py --steps 10 --warmup 3 --batch $b --scan-mode $mode --no-second-leg --no-cpu-baseline --no-gemm-ref --recall-queries 16 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'mode':'$mode','batch':$b,'ms_per_step':d['ms_per_step'],'qps':d['value'],'stage_ms':d['stage_ms'],'recall':d['recall_at_10'],'unc':d['uncertified_queries_last_step'],'bound':r['bound'],'frac':r['frac'],'hbm_gbps':r['hbm_gbps'],'mfma':r['mfma_tflops'],'int8':d.get('int8_last_step')}))" | tee -a $out/batch_sweep.jsonl
  done
done
