#!/bin/bash
# usage (repo root, after `make -C semantic_query_engine_amd/csrc KNOBS=1`): tools/build_gpp_ablate.sh 8 16 24 32 64 0s ...
# -> semantic_query_engine_amd/libsqe_gpp<tag>.so: the knobs library with the encoder's ping-pong GEMM (encoder.hip, namespace gpp)
# built with -DSQE_GPP_ABLATE=<bits> (tag = the bits; timing only, results are wrong);
# a trailing s adds the phase stamps (SQE_GEMM_DBG=4).
set -e
cd "$(dirname "$0")/../semantic_query_engine_amd/csrc"; mkdir -p build_variants
for tag in "$@"; do
  t=${tag%s}
  flags=""
  [[ $tag == *s ]] && flags="$flags -DSQE_PHASE_STAMPS=1"
  flags="$flags -DSQE_GPP_ABLATE=$t"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function \
      -DSQE_DEBUG_KNOBS $flags -c encoder.hip -o build_variants/encoder_gpp$tag.o -Rpass-analysis=kernel-resource-usage 2> build_variants/encoder_gpp$tag.txt &
done
wait
for tag in "$@"; do
  echo "$tag: gemm_pp spills $(grep -A12 'gemm_pp_kernel' build_variants/encoder_gpp$tag.txt | grep 'VGPRs Spill' | sed 's/.*Spill: //;s/ .*//' | tr '\n' ' ')"
  objs=$(ls build_knobs/*.o | grep -v "encoder")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsqe_gpp$tag.so $objs build_variants/encoder_gpp$tag.o -ldl -lpthread
done
