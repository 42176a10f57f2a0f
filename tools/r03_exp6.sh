#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp6
mkdir -p $out
SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so SQE_ENC_GRAPH=0 python tools/ring_tune.py --batch 64 --seq 32 > $out/ring_tune_2048.jsonl 2> $out/ring_tune_2048.err; cat $out/ring_tune_2048.jsonl
SQE_LIB=semantic_query_engine_amd/libsqe_knobs.so SQE_ENC_GRAPH=0 python tools/ring_tune.py --batch 64 --seq 16 > $out/ring_tune_1024.jsonl 2> $out/ring_tune_1024.err; cat $out/ring_tune_1024.jsonl
