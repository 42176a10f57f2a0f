#!/bin/bash
# A/B inside one gpurun call: the IVF list scan under rocprofv3 (kernel trace), one run of bench_configs.py --mode ivf per library;
# prints, per library, the longest dispatch of each ivf kernel (= the batch-1024 search) and the search times the run reports.
# usage: tools/r04_ivf_ab.sh <out dir> <library tags ...>   (tag "" = libsqe.so, otherwise libsqe_<tag>.so)
export TMPDIR=/tmp
out=$(realpath $1); shift
mkdir -p $out
root=$(pwd)
for t in "$@"; do
  lib=$root/semantic_query_engine_amd/libsqe${t:+_$t}.so
  cd /tmp
  SQE_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${t:-shipped} -o ivf -- python3 $root/bench_configs.py --mode ivf > $out/cfg_ivf_${t:-shipped}.json 2> $out/ivf_${t:-shipped}.err
  cd $root
  echo "== ${t:-shipped}: $(python3 -c "import json,sys; d=json.load(open('$out/cfg_ivf_${t:-shipped}.json')); print('ivf_ms', d['ivf_ms'], [ (p['batch'], p['ivf_ms']) for p in d['batch_sweep']])")" | tee -a $out/ab.log
  python3 - $out/prof_${t:-shipped} <<'PY' | tee -a $out/ab.log
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
mx = collections.defaultdict(int)
for row in csv.DictReader(open(f[0])):
    n = row['Kernel_Name']
    if 'ivf_' in n:
        mx[n.split('(')[0][:60]] = max(mx[n.split('(')[0][:60]], int(row['End_Timestamp']) - int(row['Start_Timestamp']))
for n, v in sorted(mx.items(), key=lambda kv: -kv[1]):
    print(f"   longest dispatch {v / 1e3:9.1f} us  {n}")
PY
done
