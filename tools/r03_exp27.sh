#!/bin/bash
# one query of 16 tokens through the encoder: rocprofv3 kernel summary (sum of kernel time against the 0.9 ms wall time)
export TMPDIR=/tmp
out=gpurun_out/r03_exp27; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/enc_small.py --cases 1x16 --iters 50 > $out/enc.json 2> $out/prof.err
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/enc_1x16_kernel_stats.csv; rm -rf $out/prof
cat $out/enc.json | cut -c1-200
cut -d, -f1-4 $out/enc_1x16_kernel_stats.csv | cut -c1-200 | head -16
