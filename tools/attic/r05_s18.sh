#!/bin/bash
# Round 5, GPU session 18: tile placement of the grouped weight-gradient + AdamW launch in ROUNDS of an XCD's 32 CUs with short-K
# fillers (GSTVD_GROUP_ORDER=3: units <= 27 tiles; 4: pieces of 30 tiles + remainder) against the per-XCD queues (1): parity tests,
# step A/B, kernel time, FETCH / WRITE.
export TMPDIR=/tmp; out=gpurun_out/r05_s18; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
for v in 3 4; do GSTVD_GROUP_ORDER=$v timeout 900 python -m pytest tests/test_fused_update_gpu.py tests/test_round5_gpu.py -m gpu -x -q -k "fused or block_map or grouped or update" 2>&1 | tail -2 | tee -a $out/tests.log; done
for rnd in 1 2; do for v in 1 3 4; do
  r=$(GSTVD_GROUP_ORDER=$v python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])")
  echo "GSTVD_GROUP_ORDER=$v round $rnd: ms_per_step value kernel_us = $r" | tee -a $out/order_ab.txt
done; done
for v in 1 3 4; do for c in FETCH_SIZE WRITE_SIZE; do
  GSTVD_GROUP_ORDER=$v rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc${v}_$c -- python3 bench.py --steps 2 --warmup 1 $LEAN --graph off > /dev/null 2>&1
done; done
python3 - <<'PY' | tee gpurun_out/r05_s18/pmc_order.txt
import csv, glob, collections
for v in (1, 3, 4):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in sorted(glob.glob('gpurun_out/r05_s18/pmc%d_*/*/*counter_collection.csv' % v)):
        for r in csv.DictReader(open(f)):
            if 'grouped_adamw' in r['Kernel_Name']:
                a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    d = {c: x[1] / max(x[0], 1) for c, x in agg.items()}
    print("GSTVD_GROUP_ORDER=%d grouped_adamw: launches %s FETCH_SIZE %.0f KB WRITE_SIZE %.0f KB -> 2*fetch+write = %.2f GB per launch"
          % (v, {c: x[0] for c, x in agg.items()}, d.get('FETCH_SIZE', 0), d.get('WRITE_SIZE', 0), (2 * d.get('FETCH_SIZE', 0) + d.get('WRITE_SIZE', 0)) * 1024 / 1e9))
PY
rm -rf $out/pmc*_FETCH_SIZE $out/pmc*_WRITE_SIZE
