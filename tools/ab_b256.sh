# where the batch-256 scan loses its time (knobs build): no filter / no DMA / no global bound / fast path only / two-stage kernel
for e in "SQE_X=0" "SQE_DBG=4" "SQE_DBG=2" "SQE_DBG=8" "SQE_DBG=16" "SQE_DBG=1" "SQE_SCAN=v0"; do bash tools/ab.sh "$e" 10000000 ${1:-256}; done
