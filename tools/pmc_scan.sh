#!/bin/bash
# usage: tools/pmc_scan.sh <tag> <extra env assignments...>   (run on the GPU box from repo root)
# Collects per-kernel PMC counters for the scan kernel in separate passes (gfx950: FETCH_SIZE
# needs its own pass; SQ counters 8 per pass).
tag=$1; shift
export TMPDIR=/tmp
args="--steps 3 --warmup 1 --rows ${PMC_ROWS:-10000000} --no-cpu-baseline --recall-queries 8"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  env "$@" rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}/${name} -- python3 bench.py $args > gpurun_out/pmc_${tag}_${name}.json 2> gpurun_out/pmc_${tag}_${name}.err || echo "pass $name failed"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_${tag}/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "scan_bf16" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            print("${tag}", k, "per_launch=%.6g" % (v / max(n, 1)), "launches=%d" % n)
PY
