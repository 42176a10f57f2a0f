#!/bin/bash
# usage (GPU box, repo root): tools/final_profile.sh <tag>  -> gpurun_out/<tag>_*: gpu tests, default bench line,
# the same bench under rocprofv3 --kernel-trace --stats, the hard-data config (collect pass)
tag=$1
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1 || { tail -30 gpurun_out/${tag}_tests.log; exit 1; }
tail -1 gpurun_out/${tag}_tests.log
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --no-cpu-baseline --no-gemm-ref --steps 10 > gpurun_out/${tag}_bench_prof.json 2> gpurun_out/${tag}_prof.err || exit 1
python bench_configs.py --mode hard > gpurun_out/${tag}_cfg_hard.json 2> gpurun_out/${tag}_cfg_hard.err || exit 1
find gpurun_out/${tag}_prof -name "*kernel_stats.csv"
