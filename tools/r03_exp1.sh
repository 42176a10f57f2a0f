#!/bin/bash
# r03, first GPU call: GPU tests on the hazard-audited stores and the K-scaled eps; int8 vs bf16 MFMA under power
# management; timing experiments in the ping-pong scan (knobs build): tiled DB layout emulation (SQE_DBG=8192),
# static wave priority (16384: waves 4-7, 32768: waves 0-3); L2 fills of the tiled emulation.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp1
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; echo TESTS FAILED; exit 1; }
tail -3 $out/tests.log
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power 20000 6 > $out/mfma_power.jsonl && cat $out/mfma_power.jsonl
for round in 1 2; do
  for b in 1024 512 256; do
    for dbg in 0 8192 16384 32768; do
      tools/ab.sh "SQE_DBG=$dbg" 10000000 $b >> $out/ab.log 2>&1
    done
  done
done
cat $out/ab.log
for dbg in 0 8192; do
  SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so SQE_DBG=$dbg rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_$dbg -- python3 bench.py --steps 3 --warmup 1 --rows 10000000 --no-cpu-baseline --no-gemm-ref --recall-queries 8 --batch 1024 > /dev/null 2> $out/pmc_$dbg.err
  python3 - <<PY | tee -a $out/fetch.log
import csv, glob
tot = n = 0
for f in glob.glob("$out/pmc_$dbg/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_bf16_p" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print("SQE_DBG=$dbg batch 1024: %.2f GB per scan launch (%d launches; algorithmic 20.48 GB)" % (tot / max(n, 1) * 2048 / 1e9, n))
PY
done
rm -rf $out/pmc_0 $out/pmc_8192
