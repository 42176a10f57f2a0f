for e in "SQE_SCAN128=2" "SQE_SCAN128=3"; do for b in 128 96; do bash tools/ab.sh "$e" 10000000 $b; done; done
bash tools/ab.sh "SQE_X=0" 10000000 64
