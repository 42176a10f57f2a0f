#!/bin/bash
# r03: tiled DB layout emulation (SQE_DBG=8192: K-slice-major tiles, skewed starts; WRONG scores, so no certificate /
# collect pass: SQE_NO_COLLECT=1 in both arms) against the shipped layout: scan time and L2 fills at batch 1024 / 512 / 256.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp2
mkdir -p $out
for round in 1 2; do
  for b in 1024 512 256; do
    for dbg in 0 8192; do
      tools/ab.sh "SQE_NO_COLLECT=1 SQE_DBG=$dbg" 10000000 $b >> $out/ab.log 2>&1
    done
  done
done
cat $out/ab.log
for b in 1024 256; do
for dbg in 0 8192; do
  SQE_NO_COLLECT=1 SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so SQE_DBG=$dbg rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_$dbg -- python3 bench.py --steps 3 --warmup 1 --rows 10000000 --no-cpu-baseline --no-gemm-ref --recall-queries 8 --batch $b > /dev/null 2> $out/pmc_$dbg.err
  python3 - <<PY | tee -a $out/fetch.log
import csv, glob
tot = n = 0
for f in glob.glob("$out/pmc_$dbg/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_bf16_p" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print("SQE_DBG=$dbg batch $b: %.2f GB per scan launch (%d launches; algorithmic 20.48 GB)" % (tot / max(n, 1) * 2048 / 1e9, n))
PY
  rm -rf $out/pmc_$dbg
done
done
