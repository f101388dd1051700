#!/bin/bash
# Isolated per-kernel durations: every kernel of the step serialized on one stream (no overlap), rocprofv3 kernel trace,
# aggregated by (kernel, grid size).
export TMPDIR=/tmp; out=gpurun_out/prof_serial; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 3 --warmup 1 --graph off --no-streams --no-pipeline --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d > $out/bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re, json
f = glob.glob('gpurun_out/prof_serial/*/*kernel_trace.csv')[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'copyBuffer' in n or 'at::native' in n: continue
    n = re.sub(r'\(.*', '', n); n = re.sub(r'^void ', '', n)
    gx = int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) * max(int(r.get('Grid_Size_Y', 1)), 1) * max(int(r.get('Grid_Size_Z', 1)), 1); wx = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1))); k = (n[:70], gx // max(wx, 1))
    a = agg[k]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
steps = 5.0   # warm-up 1 + timed 3 + host-issue probe 1
rows = sorted(((v[1] / steps, v[0] / steps, v[1] / v[0], k) for k, v in agg.items()), reverse=True)
tot = sum(r[0] for r in rows)
print('total kernel us/step %.0f' % tot)
for us_step, n_step, us_avg, k in rows[:45]:
    print('%8.1f us/step  %6.1f x %7.1f us  blocks=%-6d %s' % (us_step, n_step, us_avg, k[1], k[0]))
json.dump([dict(kernel=k[0], blocks=k[1], per_step=n, avg_us=a, us_per_step=u) for u, n, a, k in rows], open('gpurun_out/prof_serial/by_kernel_grid.json', 'w'), indent=0)
PY
head -1 $(ls $out/*/*kernel_trace.csv | head -1) | cut -c1-400; rm -f $out/*/*kernel_trace.csv
