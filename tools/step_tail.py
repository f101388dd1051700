#!/usr/bin/env python3
"""The end of a replayed train step kernel by kernel (what runs after backward's last kernel), from the kernel trace that
tools/timeline.sh collects under gpurun_out/timeline (it calls this script before deleting the trace):
   python3 tools/step_tail.py [ms_from_end=2.6] [min_us=15]"""
import csv, glob, re, sys
f = glob.glob('gpurun_out/timeline/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rng_advance' in r['Kernel_Name']]
step = rows[idx[-2]:idx[-1]]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
span = float(sys.argv[1]) if len(sys.argv) > 1 else 2.6
min_ns = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 15000          # only kernels longer than this many us
print('step span %.2f ms; kernels that end in its last %.1f ms:' % ((t1 - t0) / 1e6, span))
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e > t1 - span * 1e6 and (e - s) > min_ns:
        print('  q%-2s %8.3f -> %8.3f ms  (%7.1f us)  blocks=%-6d %s' % (r['Queue_Id'], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3,
              int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), re.sub(r'\(.*', '', r['Kernel_Name'])[:70]))
if len(sys.argv) > 2:
    sys.exit(0)
# memcpy / memset nodes of the captured graph: how many, and how long the queue sat idle in front of each
import collections
last_end = {}
gaps = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in step:
    q = r['Queue_Id']; s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name']
    kind = 'copyBuffer' if 'copyBuffer' in n else ('fillBuffer' if 'fillBuffer' in n else None)
    if kind and q in last_end:
        g = gaps[kind]; g[0] += 1; g[1] += max(0, s - last_end[q]) / 1e3; g[2] += (e - s) / 1e3
    last_end[q] = max(last_end.get(q, 0), e)
for k, (n, gap, dur) in gaps.items():
    print('%s nodes: %d per step, queue idle in front of them %.1f us in total, their own time %.1f us' % (k, n, gap, dur))
prev = {}
for r in step:
    q = r['Queue_Id']; s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'copyBuffer' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name']:
        p = prev.get(q)
        print('  %s q%s at %.3f ms (%.1f us) after %s (ended %.3f ms)' % (r['Kernel_Name'][:24], q, (s - t0) / 1e6, (e - s) / 1e3,
              re.sub(r'\(.*', '', p['Kernel_Name'])[:50] if p else '-', (int(p['End_Timestamp']) - t0) / 1e6 if p else 0))
    prev[q] = r
