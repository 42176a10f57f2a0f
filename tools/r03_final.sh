#!/bin/bash
# usage (GPU box, repo root): tools/r03_final.sh [part]  -> gpurun_out/r03f/*: part 1 = GPU suite + default bench line + the same
# bench under rocprofv3 + PMC passes of both first passes; part 2 = the other configs (encode / e2e with CPU baselines, IVF sweep,
# cache, ingest, hard data), group host cost, 200 k-row HNSW baseline; part 3 = batch sweep of the flat search in both scan modes
export TMPDIR=/tmp
out=gpurun_out/r03f
mkdir -p $out
part=${1:-1}
if [ "$part" = "1" ]; then
  python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
  tail -1 $out/tests.log
  python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
  cut -c1-600 $out/bench.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --no-gemm-ref --steps 10 > $out/bench_under_rocprof.json 2> $out/prof.err || exit 1
  find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv; rm -rf $out/prof
  head -12 $out/kernel_stats.csv | cut -c1-220
  PMC_MODE=int8 bash tools/pmc_scan.sh r03f_i8 > $out/pmc_i8.txt 2>&1; echo "== pmc int8 done"; tail -25 $out/pmc_i8.txt | cut -c1-200
  PMC_MODE=bf16 bash tools/pmc_scan.sh r03f_bf16 > $out/pmc_bf16.txt 2>&1; echo "== pmc bf16 done"; tail -25 $out/pmc_bf16.txt | cut -c1-200
  cp gpurun_out/pmc_r03f_i8_traffic.json gpurun_out/pmc_r03f_bf16_traffic.json $out/ 2>/dev/null
  rm -rf gpurun_out/pmc_r03f_i8 gpurun_out/pmc_r03f_bf16
elif [ "$part" = "3" ]; then
  for mode in int8 bf16; do
    for b in 1 8 64 128 256 512 1024 2048; do
      python bench.py --steps 10 --warmup 3 --batch $b --scan-mode $mode --no-second-leg --no-cpu-baseline --no-gemm-ref --recall-queries 16 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'mode':'$mode','batch':$b,'ms_per_step':d['ms_per_step'],'qps':d['value'],'stage_ms':d['stage_ms'],'recall':d['recall_at_10'],'unc':d['uncertified_queries_last_step'],'bound':r['bound'],'frac':r['frac'],'hbm_gbps':r['hbm_gbps'],'mfma':r['mfma_tflops'],'int8':d.get('int8_last_step')}))" | tee -a $out/batch_sweep.jsonl | cut -c1-230
    done
  done
else
  python bench_configs.py --mode e2e 2> $out/e2e.err | tail -1 > $out/cfg_e2e.json; echo "== e2e done"; cut -c1-400 $out/cfg_e2e.json
  python bench_configs.py --mode encode --batch 64 2> $out/encode.err | tail -1 > $out/cfg_encode.json; echo "== encode done"; cut -c1-400 $out/cfg_encode.json
  python tools/enc_small.py > $out/enc_small.jsonl 2> $out/enc_small.err; cat $out/enc_small.jsonl | cut -c1-160
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/enc_prof -- python3 tools/enc_small.py --cases 64x32 --iters 20 > /dev/null 2> $out/enc_prof.err
  find $out/enc_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/enc_64x32_kernel_stats.csv; rm -rf $out/enc_prof
  python bench_configs.py --mode ivf 2> $out/ivf.err | tail -1 > $out/cfg_ivf.json; echo "== ivf done"; cut -c1-300 $out/cfg_ivf.json
  python bench_configs.py --mode cache 2> $out/cache.err | tail -1 > $out/cfg_cache.json; echo "== cache done"
  python bench_configs.py --mode ingest 2> $out/ingest.err | tail -1 > $out/cfg_ingest.json; echo "== ingest done"
  python bench_configs.py --mode hard 2> $out/hard.err | tail -1 > $out/cfg_hard.json; echo "== hard done"
  python tools/group_host_cost.py --scan-mode bf16 > $out/group_host_cost.jsonl 2> $out/group_host_cost.err
  python tools/group_host_cost.py --scan-mode int8 >> $out/group_host_cost.jsonl 2>> $out/group_host_cost.err; cat $out/group_host_cost.jsonl
  python tools/latency_b1.py > $out/latency_b1.json 2> $out/latency_b1.err; cat $out/latency_b1.json
  python - > $out/hnsw_200k.json 2> $out/hnsw_200k.err <<PY
import json, bench
print(json.dumps(bench.cpu_baseline_hnsw(200000, 1024, 10)))
PY
  cat $out/hnsw_200k.json
fi
