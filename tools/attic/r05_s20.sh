#!/bin/bash
# Round 5, GPU session 20: start barrier between the workgroups of a long-K unit of the fused weight-gradient launch (GSTVD_GROUP_SYNC):
# (the start-barrier code this session measured was removed again afterwards -- GSTVD_GROUP_SYNC has no effect on the current tree; see profiles/r05_group_sync_ab.txt)
# parity tests, step A/B, kernel time, FETCH / WRITE.
export TMPDIR=/tmp; out=gpurun_out/r05_s20; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
GSTVD_GROUP_SYNC=1 timeout 900 python -m pytest tests/test_fused_update_gpu.py tests/test_round5_gpu.py -m gpu -x -q -k "fused or block_map or grouped or update" 2>&1 | tail -2 | tee -a $out/tests.log
for rnd in 1 2 3; do for v in 0 1; do
  r=$(GSTVD_GROUP_SYNC=$v python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "GSTVD_GROUP_SYNC=$v round $rnd: ms_per_step value = $r" | tee -a $out/sync_ab.txt
done; done
for v in 0 1; do
  GSTVD_GROUP_SYNC=$v rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$v -- python3 bench.py --steps 10 --warmup 3 $LEAN > $out/prof$v.log 2>&1
  f=$(ls $out/stats$v/*/*kernel_stats.csv | head -1); echo "GSTVD_GROUP_SYNC=$v: $(grep grouped_adamw $f | sed 's/.*)",//' | cut -d, -f1-3)" | tee -a $out/sync_ab.txt
  rm -rf $out/stats$v
  for c in FETCH_SIZE WRITE_SIZE; do
    GSTVD_GROUP_SYNC=$v rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc${v}_$c -- python3 bench.py --steps 2 --warmup 1 $LEAN --graph off > /dev/null 2>&1
  done
done
python3 - <<'PY' | tee -a gpurun_out/r05_s20/sync_ab.txt
import csv, glob, collections
for v in (0, 1):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in sorted(glob.glob('gpurun_out/r05_s20/pmc%d_*/*/*counter_collection.csv' % v)):
        for r in csv.DictReader(open(f)):
            if 'grouped_adamw' in r['Kernel_Name']:
                a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    d = {c: x[1] / max(x[0], 1) for c, x in agg.items()}
    print("GSTVD_GROUP_SYNC=%d grouped_adamw: FETCH_SIZE %.0f KB WRITE_SIZE %.0f KB -> 2*fetch+write = %.2f GB per launch" % (v, d.get('FETCH_SIZE', 0), d.get('WRITE_SIZE', 0), (2 * d.get('FETCH_SIZE', 0) + d.get('WRITE_SIZE', 0)) * 1024 / 1e9))
PY
rm -rf $out/pmc*_FETCH_SIZE $out/pmc*_WRITE_SIZE
