#!/bin/bash
# Round 5, GPU session 9: is the operand re-fetch of the fused launch caused by its AdamW epilogue?  FETCH / WRITE of the PLAIN grouped
# weight-gradient launch (GSTVD_FUSE_UPDATE=0: dW stored, AdamW as a pass of its own) under both tile placements.
export TMPDIR=/tmp; out=gpurun_out/r05_s9; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
for v in 0 1; do for c in FETCH_SIZE WRITE_SIZE; do
  GSTVD_FUSE_UPDATE=0 GSTVD_GROUP_ORDER=$v rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc${v}_$c -- python3 bench.py --steps 2 --warmup 1 $LEAN --graph off > /dev/null 2>&1
done; done
python3 - <<'PY' | tee gpurun_out/r05_s9/plain_wgrad_traffic.txt
import csv, glob, collections
for v in (0, 1):
    agg = collections.defaultdict(lambda: [0, 0.0]); dur = []
    for f in sorted(glob.glob('gpurun_out/r05_s9/pmc%d_*/*/*counter_collection.csv' % v)):
        for r in csv.DictReader(open(f)):
            if 'grouped' in r['Kernel_Name'] and 'adamw' not in r['Kernel_Name']:
                a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
                dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    d = {c: x[1] / max(x[0], 1) for c, x in agg.items()}
    print("GSTVD_FUSE_UPDATE=0 GSTVD_GROUP_ORDER=%d plain grouped wgrad: launches %s mean %.0f us under PMC; FETCH_SIZE %.0f KB WRITE_SIZE %.0f KB -> 2*fetch %.2f GB + write %.2f GB per launch (operands once: 1.7 GB; dW out 1.39 GB)"
          % (v, {c: x[0] for c, x in agg.items()}, sum(dur) / max(len(dur), 1), d.get('FETCH_SIZE', 0), d.get('WRITE_SIZE', 0), 2 * d.get('FETCH_SIZE', 0) * 1024 / 1e9, d.get('WRITE_SIZE', 0) * 1024 / 1e9))
PY
rm -rf $out/pmc*_FETCH_SIZE $out/pmc*_WRITE_SIZE
for rnd in 1 2; do for v in 0 1; do
  r=$(GSTVD_FUSE_UPDATE=0 GSTVD_GROUP_ORDER=$v python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "GSTVD_FUSE_UPDATE=0 GSTVD_GROUP_ORDER=$v round $rnd: ms_per_step value = $r" | tee -a $out/plain_wgrad_traffic.txt
done; done
# stand-alone (tools/fused_update_bench.cpp): 24 uniform long-K problems / + 12 short-K ones, library order vs per-XCD queues
for mix in "24 0" "24 12"; do for ord in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/sa_${ord}_$c -- build/fused_update_bench $mix $ord > $out/sa.log 2>&1
  done
  python3 - "$mix" $ord <<'PY' | tee -a gpurun_out/r05_s9/plain_wgrad_traffic.txt
import csv, glob, collections, sys
mix, ord_ = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('gpurun_out/r05_s9/sa_%s_*/*/*counter_collection.csv' % ord_):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'][:36]
        if 'grouped' in n:
            a = agg[n][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for n, v in agg.items():
    d = {c: x[1] / x[0] for c, x in v.items()}
    print('stand-alone %s order %s  %-38s fetch %.2f GB (x2 corrected)  write %.2f GB' % (mix, ord_, n, 2 * d.get('FETCH_SIZE', 0) * 1024 / 1e9, d.get('WRITE_SIZE', 0) * 1024 / 1e9))
PY
  rm -rf $out/sa_${ord}_FETCH_SIZE $out/sa_${ord}_WRITE_SIZE
  build/fused_update_bench $mix $ord 2>&1 | tail -4 | tee -a $out/plain_wgrad_traffic.txt
done; done
