#!/bin/bash
# IVF int8 copy in list order: IVF tests, then config 5 (bench_configs --mode ivf); then the int8 scan variants again (exp22: 0 4 20)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_exp23
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_ivf_gpu.py tests/test_config5_gpu.py tests/test_group_gpu.py tests/test_persistence_gpu.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
python bench_configs.py --mode ivf 2> $out/ivf.err | tail -1 > $out/cfg_ivf.json; python - <<PY
import json
d=json.load(open("$out/cfg_ivf.json"))
print("ivf_ms", d["ivf_ms"], "qps", d["ivf_qps"], "recall", d["recall_at_10_vs_exact"], "parity", d["parity_vs_oracle_ivf"])
for pt in d["batch_sweep"]: print(pt["batch"], pt["ivf_ms"], pt["roofline"]["frac"])
PY
bash tools/r03_exp22.sh 0 4 20 > $out/exp22.log 2>&1; grep "^libsqe" $out/exp22.log | awk '{print $2, $4, $8, $NF, $(NF-2)}' | sort
