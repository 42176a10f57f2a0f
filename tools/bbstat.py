#!/usr/bin/env python3
"""Per-basic-block instruction census of a hipcc -save-temps .s file: MFMAs, spills, barriers, LDS reads,
LDS-DMA pieces and vmcnt waits per block -- to check that a hot loop holds no scratch traffic or vmcnt(0)."""
import re
import sys

cur = None
stats = {}
order = []
for line in open(sys.argv[1]):
    t = line.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        cur = m.group(1); stats[cur] = {}; order.append(cur); continue
    m = re.match(r'^(_Z\w+):', t)
    if m:
        cur = m.group(1)[:40]; stats[cur] = {}; order.append(cur); continue
    if cur is None or not t or t.startswith(';') or t.startswith('.'):
        continue
    op = t.split()[0]
    key = None
    if op.startswith('v_mfma'): key = 'mfma'
    elif op.startswith('scratch_load'): key = 'sld'
    elif op.startswith('scratch_store'): key = 'sst'
    elif op == 's_barrier': key = 'bar'
    elif op.startswith('ds_read') or op.startswith('ds_load'): key = 'dsr'
    elif op.startswith('global_load_lds'): key = 'dma'
    elif op == 's_waitcnt' and 'vmcnt' in t: key = 'vm' + re.search(r'vmcnt\((\d+)\)', t).group(1)
    elif op.startswith('v_accvgpr'): key = 'accmov'
    stats[cur]['n'] = stats[cur].get('n', 0) + 1
    if key: stats[cur][key] = stats[cur].get(key, 0) + 1
for b in order:
    s = stats[b]
    if any(k in s for k in ('mfma', 'sld', 'sst', 'bar', 'dma')):
        print(b, s)
