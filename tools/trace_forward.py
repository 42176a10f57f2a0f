#!/usr/bin/env python3
"""Per-kernel execution time and launch gaps of ONE encoder forward out of a rocprofv3 kernel trace.
usage: trace_forward.py <dir with *kernel_trace.csv> [forward index]   (a forward = embed_ln_kernel ... pool_ln_kernel)"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n[n.index('sqe::'):] if 'sqe::' in n else n
    return n.replace('(anonymous namespace)::', '').replace('sqe::', '').split('(')[0][:40]
starts = [i for i, r in enumerate(rows) if 'embed_ln_kernel' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
i0 = starts[which]
i1 = next(i for i in range(i0, len(rows)) if 'pool_ln_kernel' in rows[i]['Kernel_Name'])
ex = collections.defaultdict(list); gap = collections.defaultdict(list)
for i in range(i0, i1 + 1):
    r = rows[i]; n = short(r['Kernel_Name'])
    ex[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    if i > i0: gap[n].append((int(r['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3)
tot = (int(rows[i1]['End_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3
print(f"forward {which}: {i1 - i0 + 1} kernels, {tot:.1f} us first start -> last end; executing {sum(map(sum, ex.values())):.1f} us, gaps {sum(map(sum, gap.values())):.1f} us")
for n, v in sorted(ex.items(), key=lambda kv: -sum(kv[1])):
    g = gap.get(n, [0.0])
    print(f"   {n:42s} x{len(v):3d}  exec mean {sum(v) / len(v):6.2f} us (min {min(v):5.2f})   gap before: mean {sum(g) / len(g):5.2f} us")
