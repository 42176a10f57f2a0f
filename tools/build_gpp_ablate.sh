#!/bin/bash
# usage (repo root, after `make -C semantic_query_engine_amd/csrc KNOBS=1`): tools/build_gpp_ablate.sh 8 16 24 32 64 0s ...   (a trailing s: phase stamps too, SQE_GEMM_DBG=4)
# -> semantic_query_engine_amd/libsqe_gpp<bits>.so: the knobs library with the encoder's ping-pong GEMM built with
# -DSQE_GPP_ABLATE=<bits> (encoder.hip, namespace gpp); timing only, results are wrong.
set -e
cd "$(dirname "$0")/../semantic_query_engine_amd/csrc"
for bits in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function \
      -DSQE_DEBUG_KNOBS -DSQE_GPP_ABLATE=${bits%s} $( [[ $bits == *s ]] && echo -DSQE_PHASE_STAMPS=1 ) -c encoder.hip -o build_knobs/encoder_gpp$bits.o &
done
wait
for bits in "$@"; do
  objs=$(ls build_knobs/*.o | grep -v "encoder")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsqe_gpp$bits.so $objs build_knobs/encoder_gpp$bits.o -ldl -lpthread
done
