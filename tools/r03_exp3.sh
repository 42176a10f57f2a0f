#!/bin/bash
# r03: full GPU test suite on the r03 fixes + new tests; encoder at config-3 / reference shapes (timings + rocprof kernel summary
# of 64 x 32 tokens); IVF batch sweep at 10 M rows.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp3
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; echo TESTS FAILED; exit 1; }
tail -3 $out/tests.log
python tools/enc_small.py > $out/enc_small.jsonl 2> $out/enc_small.err; cat $out/enc_small.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $out/enc_prof -- python3 tools/enc_small.py --cases 64x32 --iters 20 > /dev/null 2> $out/enc_prof.err
find $out/enc_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/enc_64x32_kernel_stats.csv
rm -rf $out/enc_prof
head -30 $out/enc_64x32_kernel_stats.csv
python bench_configs.py --mode ivf 2> $out/ivf.err | tail -1 > $out/cfg_ivf.json; cat $out/cfg_ivf.json
