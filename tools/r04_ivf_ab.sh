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
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n[n.index('sqe::') :] if 'sqe::' in n else n
    return n.replace('(anonymous namespace)::', '').replace('sqe::', '').split('(')[0][:44]
dur = collections.defaultdict(list)
for r in rows:
    if 'sqe::' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']:
        dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in sorted(dur.items(), key=lambda kv: -max(kv[1])):
    if n.startswith('ivf_') or 'gemm_ring' in n or 'normalize' in n or 'quantize_rows' in n:
        print(f"   {n:46s} dispatches {len(v):5d}   longest {max(v):9.1f} us (batch 1024)   shortest {min(v):7.1f} us (batch 1)")
# the kernels of the LAST batch-1 search of the run (the sweep ends with the flat index: take the last ivf_select whose strip pass was short)
# (a gated strip select of the collect mode is a ~4 us no-op: only dispatches that did work count)
sel = [i for i, r in enumerate(rows) if 'ivf_select_kernel' in r['Kernel_Name'] and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 10000]
one = min(sel, key=lambda i: int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp']))
j = one
while j > 0 and 'ivf_select_kernel' not in rows[j - 1]['Kernel_Name'] and one - j < 12: j -= 1
t0 = int(rows[j]['Start_Timestamp'])
print("   one query (batch 1), kernel by kernel:")
for r in rows[j:one + 1]:
    print(f"      +{(int(r['Start_Timestamp']) - t0) / 1e3:7.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  {short(r['Kernel_Name'])}")
print(f"      = {(int(rows[one]['End_Timestamp']) - t0) / 1e3:.1f} us from the first kernel's start to the last one's end")
PY
done
