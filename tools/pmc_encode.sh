#!/bin/bash
# usage: tools/pmc_encode.sh <tag>   (GPU box, repo root): per-kernel SQ counters of the encoder's kernels, two --pmc passes
tag=$1
export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}/p$i -- python3 bench_configs.py --mode encode --batch 64 > /dev/null 2> gpurun_out/pmc_${tag}_p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmc_${tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "sqe::" not in name: continue
        m = re.search(r"::(\w+(?:<[^>]*>)?)\(", name)
        key = m.group(1) if m else name[:60]
        a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print("   %-34s per_launch=%.6g  launches=%d" % (c, v / max(n, 1), n))
PY
