#!/bin/bash
# usage (repo root, after `make -C semantic_query_engine_amd/csrc KNOBS=1`): tools/build_variant.sh <source> <tag> <flags...>
#   e.g. tools/build_variant.sh scan_i8 2bar -DSQE_I8_TWO_BARRIERS
# -> semantic_query_engine_amd/libsqe_<tag>.so: the knobs library with <source>.hip rebuilt with the extra flags (A/B runs)
set -e
src=$1; tag=$2; shift 2
cd "$(dirname "$0")/../semantic_query_engine_amd/csrc"
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -DSQE_DEBUG_KNOBS "$@" \
    -c $src.hip -o build_variants/${src}_$tag.o -Rpass-analysis=kernel-resource-usage 2> build_variants/${src}_$tag.txt
grep -A12 "Function Name" build_variants/${src}_$tag.txt | grep "Function Name\|VGPRs:\|VGPRs Spill\|ScratchSize" | sed 's/.*remark: [^ ]* *//' | paste - - - - | cut -c1-200
objs=$(ls build_knobs/*.o | grep -v "/${src}\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsqe_$tag.so $objs build_variants/${src}_$tag.o -ldl -lpthread
