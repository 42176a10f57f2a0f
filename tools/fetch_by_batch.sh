#!/bin/bash
# usage (GPU box, repo root): tools/fetch_by_batch.sh -> L2 fills (FETCH_SIZE x 2 KiB) per scan launch at batch 256 / 512 / 1024: one, two, four workgroups per DB chunk
export TMPDIR=/tmp
for b in 256 512 1024; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fb_$b -- python3 bench.py --scan-mode ${SCAN_MODE:-bf16} --no-second-leg --steps 3 --warmup 1 --rows 10000000 --no-cpu-baseline --no-gemm-ref --recall-queries 8 --batch $b > /dev/null 2> gpurun_out/pmc_fb_$b.err
  python3 - <<PY
import csv, glob
tot = n = 0
for f in glob.glob("gpurun_out/pmc_fb_$b/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_bf16_p" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print("batch $b: %.2f GB per scan launch (%d launches; algorithmic 20.48 GB)" % (tot / max(n, 1) * 2048 / 1e9, n))
PY
done
