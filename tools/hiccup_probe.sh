#!/bin/bash
# Do the sporadic 100-400 us holes inside replayed steps (tools/trace_report.py lists them per step) coincide with the host's
# hipGraphLaunch of the NEXT replay?  Kernel trace + HIP runtime API trace of the same run, correlated by timestamp.
export TMPDIR=/tmp; out=gpurun_out/hiccup; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $out -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d > $out/bench.log 2>&1
ls $out/*/ | head
python3 - <<'PY'
import csv, glob, re
kt = glob.glob('gpurun_out/hiccup/*/*kernel_trace.csv')[0]
at = glob.glob('gpurun_out/hiccup/*/*hip_api_trace.csv')
print('api trace files', at)
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rng_advance' in r['Kernel_Name']]
api = []
if at:
    for r in csv.DictReader(open(at[0])):
        if 'GraphLaunch' in r['Function']:
            api.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
print('hipGraphLaunch calls', len(api), 'median duration %.1f us' % (sorted(e - s for s, e in api)[len(api) // 2] / 1e3 if api else 0))
for k in range(3, len(idx) - 1):
    step = rows[idx[k]:idx[k + 1]]
    t0 = int(step[0]['Start_Timestamp'])
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
    cur = ev[0][1]; gaps = []
    for s, e in ev[1:]:
        if s > cur: gaps.append((cur, s))
        cur = max(cur, e)
    big = [(a, b) for a, b in gaps if b - a > 40000]
    span = (max(e for _, e in ev) - t0) / 1e6
    notes = []
    for a, b in big:
        hit = [(s, e) for s, e in api if s < b and e > a]
        notes.append('%d us at %.2f ms%s' % ((b - a) / 1e3, (a - t0) / 1e6, ' [inside a hipGraphLaunch call %.0f us long]' % ((hit[0][1] - hit[0][0]) / 1e3) if hit else ''))
    inl = [(s, e) for s, e in api if s >= t0 and s < int(rows[idx[k + 1]]['Start_Timestamp'])]
    print('step %d: %.2f ms; hipGraphLaunch calls starting inside it: %s; holes > 40 us: %s' % (
        k, span, ['%.2f-%.2f ms' % ((s - t0) / 1e6, (e - t0) / 1e6) for s, e in inl], notes))
PY
rm -rf $out/*/
