#!/bin/bash
# usage (GPU box, repo root): tools/ab_refresh.sh -> bound-table fetch schedule: the r02a one (SQE_DBG=4096: 32 / 128 / every 2nd tile) vs the default (4 / 32 / every 4th)
for rep in 1 2; do for b in 1024 512 256; do for d in 4096 0; do bash tools/ab.sh "SQE_DBG=$d" 10000000 $b; done; done; done
