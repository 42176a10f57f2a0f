#!/bin/bash
# one gpurun call of r04: what the key stores of the int8 collect scan cost -- scan_i8_pp_kernel's average under rocprofv3 with and
# without them (knobs build, SQE_I8_DBG=32: survivors found, nothing appended -- every query takes the bf16 pass), 10 M and 1.25 M rows
export TMPDIR=/tmp
out=$(realpath gpurun_out/r04k); mkdir -p $out
root=$(pwd)
for rows in 10000000 1250000; do for dbg in 0 32 0 32; do
  cd /tmp
  SQE_LIB=$root/semantic_query_engine_amd/libsqe_knobs.so SQE_I8_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o s -- python3 $root/bench.py --scan-mode int8 --no-second-leg --no-clustered-leg --steps 4 --warmup 1 --rows $rows --no-cpu-baseline --no-gemm-ref --recall-queries 16 > /dev/null 2> $out/err.txt
  cd $root
  echo "rows $rows SQE_I8_DBG=$dbg: $(grep scan_i8_pp_kernel $out/p/s_kernel_stats.csv | cut -d, -f2-4 | tr -d '\"')  (calls, total ns, average ns)" | tee -a $out/no_store_ab.log
  rm -rf $out/p
done; done
for r in 1 2; do for rows in 10000000 1250000; do for t in knobs i8v129; do for b in 1024 256; do
  SCAN_MODE=int8 tools/ab_lib.sh semantic_query_engine_amd/libsqe_$t.so "X=0" $rows $b | tee -a $out/ab_young.log
done; done; done; done
timeout -k 10 300 env SQE_LIB=semantic_query_engine_amd/libsqe_i8v129.so python -m pytest tests/test_i8_exact_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee $out/tests_i8v129.log
