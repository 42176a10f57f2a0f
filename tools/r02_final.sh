#!/bin/bash
# usage (GPU box, repo root): tools/r02_final.sh -> gpurun_out/r02f_*: the whole GPU suite, the default bench line, the same
# bench under rocprofv3, PMC passes of the scan, phase stamps, batch sweep, hard-data config
export TMPDIR=/tmp
bash tools/final_profile.sh r02f || exit 1
echo "== final_profile done"
bash tools/pmc_scan.sh r02f > gpurun_out/r02f_pmc.txt 2>&1; echo "== pmc done"
bash tools/counters.sh 1024 > gpurun_out/r02f_counters.log 2>&1; echo "== stamps done"
bash tools/batch_sweep.sh > gpurun_out/r02f_batch_sweep.jsonl 2> gpurun_out/r02f_batch_sweep.err; echo "== sweep done"
