#!/bin/bash
# usage (GPU box, repo root): tools/counters.sh [batches]
#  1. knobs build, SQE_DBG=32: core clock of the scan as shipped (two clock reads per workgroup, nothing else)
#  2. knobs + STAMPS=1 / STAMPS=2: per-phase cycle sums of the ping-pong scan (compute side / memory side too)
#  3. counters build (COUNTERS=1): filter event counters, k-row bound on (SQE_DBG=32) and off (96)
C=semantic_query_engine_amd/csrc
make -C $C KNOBS=1 -j16 > /dev/null || exit 1
make -C $C KNOBS=1 STAMPS=1 OBJDIR=/tmp/st1 OUT=/tmp/libsqe_st1.so -j16 > /dev/null || exit 1
make -C $C KNOBS=1 STAMPS=2 OBJDIR=/tmp/st2 OUT=/tmp/libsqe_st2.so -j16 > /dev/null || exit 1
make -C $C KNOBS=1 COUNTERS=1 OBJDIR=/tmp/cnt OUT=/tmp/libsqe_cnt.so -j16 > /dev/null || exit 1
run() { env SQE_LIB=$1 SQE_DBG=$2 python bench.py --scan-mode ${SCAN_MODE:-bf16} --no-second-leg --steps 3 --warmup 2 --rows 10000000 --batch $3 --no-cpu-baseline --no-gemm-ref --recall-queries 16 2>&1 >/dev/null | grep "sqe dbg" | tail -${4:-40}; }
for b in ${1:-1024 256}; do
  echo "== knobs build, SQE_DBG=32, batch $b"; run $C/../libsqe_knobs.so 32 $b 1
  echo "== STAMPS=1, batch $b"; run /tmp/libsqe_st1.so 32 $b
  echo "== STAMPS=2, batch $b"; run /tmp/libsqe_st2.so 32 $b
  echo "== counters build, SQE_DBG=32 batch $b"; run /tmp/libsqe_cnt.so 32 $b 1
  echo "== counters build, SQE_DBG=96 batch $b"; run /tmp/libsqe_cnt.so 96 $b 1
done
