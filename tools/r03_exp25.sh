#!/bin/bash
# ring GEMM with the next K step's fragments prefetched into registers: encoder tests, then tools/enc_small.py on libsqe_nopf.so
# (tools/build_variant.sh encoder nopf -DSQE_RING_PREFETCH=0) and on libsqe_knobs.so, twice
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp25
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_encoder_gpu.py tests/test_config3_gpu.py tests/test_config1_gpu.py tests/test_config5_gpu.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
for r in 1 2; do for lib in nopf knobs; do
  echo "== $lib"
  SQE_LIB=semantic_query_engine_amd/libsqe_$lib.so python tools/enc_small.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['batch'], d['seq_len'], d['encode_ms'])" | tee -a $out/enc_$lib.log
done; done
