for e in "SQE_X=0" "SQE_DBG=4" "SQE_DBG=8" "SQE_DBG=16" "SQE_SCAN=v0" "SQE_DBG=1"; do bash tools/ab.sh "$e" 1250000 1024; done
for e in "SQE_X=0" "SQE_DBG=4"; do bash tools/ab.sh "$e" 2500000 1024; done
