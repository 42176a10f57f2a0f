#!/usr/bin/env python3
"""The last N kernel dispatches of a rocprofv3 kernel trace, with start offsets and durations (what one small search is made of).
usage: trace_tail.py <dir with *kernel_trace.csv> [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    name = r['Kernel_Name']
    name = name[name.index('sqe::'):] if 'sqe::' in name else name
    name = name.replace('(anonymous namespace)::', '').replace('sqe::', '').split('(')[0][:48]
    print(f"+{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {name}  grid {r.get('Grid_Size_X', '?')}/{r.get('Workgroup_Size_X', '?')}")
