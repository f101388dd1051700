#!/bin/bash
# host time of hipGraphLaunch vs GPU span of a step, at a shrunken workload (same node count)
export TMPDIR=/tmp; out=gpurun_out/launchp; rm -rf $out; mkdir -p $out
export GSTVD_ROW_SPLIT=0
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $out -- python3 bench.py --rows-per-gpu ${1:-2} --seq-len ${2:-32} --steps 12 --warmup 2 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d > $out/bench.log 2>&1
python3 - <<'PY'
import csv, glob
kt = glob.glob('gpurun_out/launchp/*/*kernel_trace.csv')[0]
at = glob.glob('gpurun_out/launchp/*/*hip_api_trace.csv')[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rng_advance' in r['Kernel_Name']]
api = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(at)) if 'GraphLaunch' in r['Function']]
d = sorted(e - s for s, e in api)
print('hipGraphLaunch calls', len(api), 'host duration median %.2f ms  min %.2f  max %.2f' % (d[len(d)//2]/1e6, d[0]/1e6, d[-1]/1e6))
for k in range(4, min(len(idx) - 1, 8)):
    step = rows[idx[k]:idx[k + 1]]
    t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
    # union of busy intervals
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
    cur_s, cur_e = ev[0]; uni = 0
    for s, e in ev[1:]:
        if s > cur_e: uni += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    uni += cur_e - cur_s
    inl = [(s, e) for s, e in api if s <= t1 and e >= t0]
    print('step %d: %d kernels, span %.2f ms, some-kernel-running %.2f ms, sum of durations %.2f ms; graph launches overlapping: %s'
          % (k, len(step), (t1 - t0) / 1e6, uni / 1e6, busy / 1e6, ['%.2f..%.2f' % ((s - t0) / 1e6, (e - t0) / 1e6) for s, e in inl]))
PY
rm -rf $out/*/
