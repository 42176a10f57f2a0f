#!/bin/bash
# usage (GPU box, repo root): tools/counters.sh  -> filter event counters + core clock of the scan (counters build), per batch
make -C semantic_query_engine_amd/csrc KNOBS=1 COUNTERS=1 OBJDIR=/tmp/cnt OUT=/tmp/libsqe_cnt.so -j16 > /dev/null || exit 1
for dbg in 32 96; do
for b in 1024 256; do
  echo "== SQE_DBG=$dbg batch $b"
  env SQE_LIB=/tmp/libsqe_cnt.so SQE_DBG=$dbg python bench.py --steps 2 --warmup 1 --rows 10000000 --batch $b --no-cpu-baseline --no-gemm-ref --recall-queries 16 2>&1 >/dev/null | grep "sqe dbg" | tail -1
done
done
