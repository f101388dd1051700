#!/bin/bash
# Where does the main (text) queue wait?  Gap before every kernel of one replayed step on the busiest queue: histogram, and the
# largest gaps with the kernels around them (what the queue was waiting for: the other stream, or just the dependent-launch gap).
export TMPDIR=/tmp; out=gpurun_out/gaps; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d > $out/bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob('gpurun_out/gaps/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rng_advance' in r['Kernel_Name']]
step = rows[idx[-2]:idx[-1]]
t0 = int(step[0]['Start_Timestamp'])
def short(n): return re.sub(r'\(.*', '', n)[:46]
by = collections.defaultdict(list)
for r in step: by[r['Queue_Id']].append(r)
for q, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    gaps = []
    for a, b in zip(v, v[1:]):
        gaps.append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
    tot = sum(g for g in gaps if g > 0)
    hist = collections.Counter(min(int(g // 2) * 2, 40) for g in gaps)
    print('queue %s: %d kernels, sum of gaps %.2f ms, median gap %.1f us' % (q, len(v), tot / 1e3, sorted(gaps)[len(gaps) // 2]))
    print('   gap histogram (us bucket: count):', ' '.join('%d:%d' % (k, hist[k]) for k in sorted(hist)))
    big = sorted(range(len(gaps)), key=lambda i: -gaps[i])[:12]
    for i in sorted(big):
        print('   %7.1f us gap at %.2f ms  after %-46s before %s' % (gaps[i], (int(v[i]['End_Timestamp']) - t0) / 1e6, short(v[i]['Kernel_Name']), short(v[i + 1]['Kernel_Name'])))
PY
rm -f $out/*/*kernel_trace.csv
