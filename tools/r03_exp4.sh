#!/bin/bash
# r03: full GPU suite (group IVF, int8 incl. the 10 M-row case), then the IVF batch-64 anomaly under rocprofv3
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp4
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -25 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
for b in 64 1024; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/ivf_prof_$b -- python3 tools/ivf_batch_probe.py --batch $b > $out/ivf_probe_$b.json 2> $out/ivf_probe_$b.err
  find $out/ivf_prof_$b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/ivf_b${b}_kernel_stats.csv
  rm -rf $out/ivf_prof_$b
  cat $out/ivf_probe_$b.json; grep -E "ivf_|scores_gemm|gemm_ring|normalize" $out/ivf_b${b}_kernel_stats.csv | cut -c1-200 | head -12
done
