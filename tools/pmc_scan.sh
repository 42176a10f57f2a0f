#!/bin/bash
# usage: tools/pmc_scan.sh <tag> <extra env assignments...>   (run on the GPU box from repo root)
# Collects per-kernel PMC counters for the scan kernel in separate passes (gfx950: FETCH_SIZE
# needs its own pass; SQ counters 8 per pass).
tag=$1; shift
export TMPDIR=/tmp
mode=${PMC_MODE:-bf16}     # bf16 | int8: which first pass is profiled (the kernel filter below follows)
args="--steps 3 --warmup 1 --rows ${PMC_ROWS:-10000000} --no-cpu-baseline --no-gemm-ref --recall-queries 8 --batch ${PMC_BATCH:-1024} --scan-mode $mode --no-second-leg"
mem_passes=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
[ "${PMC_SQ_ONLY:-0}" = "1" ] && mem_passes=()
for pass in "${mem_passes[@]}" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_I8" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  env "$@" rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}/${name} -- python3 bench.py $args > gpurun_out/pmc_${tag}_${name}.json 2> gpurun_out/pmc_${tag}_${name}.err || echo "pass $name failed"
done
sha=$(python3 -c "import bench; print(bench.scan_source_hash('$mode'))")
python3 - <<PY
import csv, glob, collections, json
tot = {}
for d in sorted(glob.glob("gpurun_out/pmc_${tag}/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            # the main scan launches only (the collect-mode instantiation <..., true> exits at once
            # when every query is certified)
            if ("$mode" == "int8" and "scan_i8_" in name) or ("$mode" != "int8" and ("scan_bf16_p" in name or ("scan_bf16_kernel" in name and "false" in name))):
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            print("${tag}", k, "per_launch=%.6g" % (v / max(n, 1)), "launches=%d" % n)
            tot[k] = v / max(n, 1)
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    # rocprofv3 reports KiB; gfx950 tallies the 128-B requests of wide streaming reads at 64 B
    # (MI355X_MICROARCH.md, HBM section): double FETCH_SIZE, take WRITE_SIZE as is
    fetch = tot["FETCH_SIZE"] * 1024 * 2
    write = tot["WRITE_SIZE"] * 1024
    json.dump({"rows": int("${PMC_ROWS:-10000000}"), "batch": int("${PMC_BATCH:-1024}"), "tag": "${tag}", "scan_mode": "$mode", "scan_src_sha": "${sha}",
               "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
               "hbm_bytes_per_launch": fetch + write,
               "note": "FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024, separate --pmc passes, "
                       "average over the scan kernel's launches"},
              open("gpurun_out/pmc_${tag}_traffic.json", "w"), indent=1)
PY
