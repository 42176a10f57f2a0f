for b in 256 512; do for e in "SQE_SCAN=pp" "SQE_SCAN=p8" "SQE_SCAN=v0"; do bash tools/ab.sh "$e" 10000000 $b; done; done
