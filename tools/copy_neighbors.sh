#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/prof_cp; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 1 --warmup 1 --graph off --no-cpu-baseline --no-breakdown > $out/bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_cp/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
c = collections.Counter()
for i, n in enumerate(names):
    if 'copyBuffer' in n:
        c[(names[i-1][:60] if i else '', names[i+1][:60] if i+1 < len(names) else '')] += 1
for k, v in c.most_common(12): print(v, k)
print('total copies', sum(c.values()), 'of', len(names))
PY
