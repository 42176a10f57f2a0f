#!/bin/bash
# usage (GPU box, repo root): tools/r04_final.sh <part>  -> gpurun_out/r04f/*
#   1  the GPU suite + the default bench line + the same bench under rocprofv3 (kernel stats)
#   2  PMC passes (HBM traffic, SQ counters) of both first passes at batch 1024 -> pmc_traffic_{int8,bf16}.json (scan sources hashed)
#   3  the other configs: encode / e2e, encoder at the reference's shapes (+ rocprofv3 summary of 64 x 512, kernel-by-kernel trace of one
#      1 x 16-token forward), hard data, cache, ingest, single-query latency, group host cost
#   5  IVF: its tests, the batch sweep under rocprofv3 (kernel trace), FETCH / WRITE passes
#   4  batch sweep of the flat search in both scan modes, shard sizes, launch fixed cost
export TMPDIR=/tmp
out=gpurun_out/r04f
mkdir -p $out
part=${1:-1}
if [ "$part" = "1" ]; then
  python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
  tail -1 $out/tests.log
  python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
  cut -c1-900 $out/bench.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --no-gemm-ref --steps 10 > $out/bench_under_rocprof.json 2> $out/prof.err || exit 1
  find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv; rm -rf $out/prof
  head -14 $out/kernel_stats.csv | cut -c1-220
elif [ "$part" = "2" ]; then
  PMC_MODE=int8 bash tools/pmc_scan.sh r04f_i8 > $out/pmc_counters_int8.txt 2>&1; echo "== pmc int8 done"; tail -25 $out/pmc_counters_int8.txt | cut -c1-200
  PMC_MODE=bf16 bash tools/pmc_scan.sh r04f_bf16 > $out/pmc_counters_bf16.txt 2>&1; echo "== pmc bf16 done"; tail -25 $out/pmc_counters_bf16.txt | cut -c1-200
  cp gpurun_out/pmc_r04f_i8_traffic.json $out/pmc_traffic_int8.json; cp gpurun_out/pmc_r04f_bf16_traffic.json $out/pmc_traffic_bf16.json
  rm -rf gpurun_out/pmc_r04f_i8 gpurun_out/pmc_r04f_bf16
elif [ "$part" = "3" ]; then
  python bench_configs.py --mode e2e 2> $out/e2e.err | tail -1 > $out/cfg_e2e.json; echo "== e2e done"; cut -c1-400 $out/cfg_e2e.json
  python bench_configs.py --mode encode --batch 64 2> $out/encode.err | tail -1 > $out/cfg_encode.json; echo "== encode done"; cut -c1-400 $out/cfg_encode.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/enc_prof -- python3 bench_configs.py --mode encode --batch 64 --no-cpu-baseline > /dev/null 2> $out/enc_prof.err
  find $out/enc_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/enc_64x512_kernel_stats.csv; rm -rf $out/enc_prof
  python tools/enc_small.py > $out/enc_small.jsonl 2> $out/enc_small.err; cat $out/enc_small.jsonl | cut -c1-160
  python tools/latency_b1.py > $out/latency_b1.json 2> $out/latency_b1.err; cat $out/latency_b1.json
  # one 1 x 16-token forward, kernel by kernel (execution time against launch gaps)
  rocprofv3 --kernel-trace --output-format csv -d $out/b1_prof -- python3 tools/latency_b1.py > /dev/null 2> $out/b1_prof.err
  python3 tools/trace_forward.py $out/b1_prof 10 | tee $out/enc_1x16_forward_trace.txt; rm -rf $out/b1_prof
  python bench_configs.py --mode hard 2> $out/hard.err | tail -1 > $out/cfg_hard.json; echo "== hard done"; cat $out/cfg_hard.json
  python bench_configs.py --mode cache 2> $out/cache.err | tail -1 > $out/cfg_cache.json; echo "== cache done"
  python bench_configs.py --mode ingest 2> $out/ingest.err | tail -1 > $out/cfg_ingest.json; echo "== ingest done"
  python tools/group_host_cost.py --scan-mode int8 > $out/group_host_cost.jsonl 2> $out/group_host_cost.err; cat $out/group_host_cost.jsonl
elif [ "$part" = "5" ]; then
  { timeout -k 10 500 python -m pytest tests/test_ivf_gpu.py tests/test_config5_gpu.py -m gpu -q -x > $out/tests_ivf.log 2>&1 || [ $? -eq 1 ]; } || exit 1
  tail -3 $out/tests_ivf.log
  grep -q passed $out/tests_ivf.log && ! grep -q failed $out/tests_ivf.log || { tail -60 $out/tests_ivf.log; exit 1; }
  # IVF: the sweep, then its kernel trace (longest dispatch of a kernel = the batch-1024 search, shortest list scan = batch 1) and HBM traffic
  tools/r04_ivf_ab.sh $out ""
  for pass in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/ivf_pmc_$pass -- python3 bench_configs.py --mode ivf > /dev/null 2> $out/ivf_pmc_$pass.err
  done
  python3 - $out <<'PY' | tee $out/ivf_pmc_traffic.txt
import csv, glob, sys, collections
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    best = collections.defaultdict(float)
    for f in glob.glob(f"{out}/ivf_pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "ivf_list_" in n or "ivf_select" in n:
                k = n[n.index("ivf_"):][:26]
                best[k] = max(best[k], float(r["Counter_Value"]))      # the batch-1024 dispatch is the largest
    for k, v in best.items():
        # rocprofv3 reports KiB; gfx950 tallies wide streaming reads at half their bytes (MI355X_MICROARCH.md): FETCH x 2
        print(f"{c} {k}: largest dispatch {v * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 1e9:.3f} GB")
PY
  rm -rf $out/ivf_pmc_FETCH_SIZE $out/ivf_pmc_WRITE_SIZE
else
  for mode in int8 bf16; do
    for b in 1 8 64 128 256 512 1024 2048; do
      python bench.py --steps 10 --warmup 3 --batch $b --scan-mode $mode --no-second-leg --no-clustered-leg --no-cpu-baseline --no-gemm-ref --recall-queries 16 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'mode':'$mode','batch':$b,'ms_per_step':d['ms_per_step'],'qps':d['value'],'stage_ms':d['stage_ms'],'recall':d['recall_at_10'],'unc':d['uncertified_queries_last_step'],'bound':r['bound'],'frac':r['frac'],'hbm_gbps':r['hbm_gbps'],'mfma':r['mfma_tflops'],'int8':d.get('int8_last_step')}))" | tee -a $out/batch_sweep.jsonl | cut -c1-230
    done
  done
  python tools/scan_fixed_cost.py > $out/scan_fixed_cost.jsonl 2> $out/scan_fixed_cost.err; cat $out/scan_fixed_cost.jsonl | cut -c1-200
  bash tools/shard_sweep.sh > $out/shard_sizes.json 2> $out/shard_sizes.err; tail -40 $out/shard_sizes.json
fi
