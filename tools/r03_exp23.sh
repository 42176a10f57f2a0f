#!/bin/bash
# IVF int8 copy in tiled list order: IVF tests, config 5 (bench_configs --mode ivf), the batch sequence probe
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
python tools/ivf_batch_probe.py --sequence 1024,1,8,64,256,1024 > $out/ivf_seq.jsonl 2> $out/ivf_seq.err; cut -c1-200 $out/ivf_seq.jsonl
