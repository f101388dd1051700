#!/bin/bash
# Per-stream timeline of one replayed step: where each HIP stream starts / ends, how busy it is, what runs last.
export TMPDIR=/tmp; out=gpurun_out/timeline; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d > $out/bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob('gpurun_out/timeline/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steps are delimited by rng_advance (first kernel of a train step)
idx = [i for i, r in enumerate(rows) if 'rng_advance' in r['Kernel_Name']]
print('steps found', len(idx))
a, b = idx[-2], idx[-1]            # the last complete replayed step
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
print('step span %.2f ms, %d kernels' % ((t1 - t0) / 1e6, len(step)))
def short(n): return re.sub(r'\(.*', '', n)[:48]
by = collections.defaultdict(list)
for r in step: by[(r['Queue_Id'], r.get('Stream_Id', '?'))].append(r)
for k, v in sorted(by.items(), key=lambda kv: int(kv[1][0]['Start_Timestamp'])):
    s = int(v[0]['Start_Timestamp']); e = max(int(r['End_Timestamp']) for r in v)
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in v)
    print('queue/stream %s: %4d kernels, first %.2f ms, last end %.2f ms, busy %.2f ms | last: %s' % (k, len(v), (s - t0) / 1e6, (e - t0) / 1e6, busy / 1e6, short(v[-1]['Kernel_Name'])))
# coarse phases on the whole step: when do ce_fwd / ce_bwd run
for r in step:
    if 'ce_fwd' in r['Kernel_Name'] or 'ce_bwd' in r['Kernel_Name']:
        print('%s at %.2f ms' % (short(r['Kernel_Name']), (int(r['Start_Timestamp']) - t0) / 1e6))
# decoder phase markers: ckv projection (1368 blocks of the 256-tile kernel) .. LM head (240 blocks) .. ce .. ckv dgrad (222 blocks)
for r in step:
    gx = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    n = r['Kernel_Name']
    if ('gemm_dma256_kernel' in n and gx in (1368, 240)) or ('gemm_dma_kernel' in n and gx == 222) or 'vl_split' in n:
        print('%-50s blocks=%-5d start %.2f ms  end %.2f ms' % (short(n), gx, (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6))
# idle gaps: time with no kernel running
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
cur_e, idle = ev[0][1], 0
for s, e in ev[1:]:
    if s > cur_e: idle += s - cur_e
    cur_e = max(cur_e, e)
print('no-kernel-running time inside the step: %.2f ms' % (idle / 1e6))
# concurrency histogram
pts = sorted([(s, 1) for s, e in ev] + [(e, -1) for s, e in ev])
lvl, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[lvl] += t - last; last = t; lvl += d
print('time by number of kernels in flight (ms):', {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
PY
python3 tools/step_tail.py 2.6        # what runs after backward's last kernel
python3 tools/step_tail.py 0.5 0 | tail -n +2 | sed "s/^/  [all]/"     # ... and everything in the last half millisecond
rm -f $out/*/*kernel_trace.csv
