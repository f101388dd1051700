#!/bin/bash
# Round 5, GPU session 1: new tests, full suite, whole-step A/B of the grouped launch's tile placement (GSTVD_GROUP_ORDER 0 = the
# library's chunked order of rounds 1-4, 1 = per-XCD queues, 2 = + staggered lead-in), its kernel time and fabric traffic, and the
# sharded update on the 1-rank RCCL path.
export TMPDIR=/tmp; out=gpurun_out/r05_s1; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
python3 -X faulthandler -m pytest tests/test_round5_gpu.py tests/test_dp_gpu.py tests/test_fused_update_gpu.py -x -q -m gpu -p no:cacheprovider > $out/new_tests.log 2>&1; echo "rc=$?" >> $out/new_tests.log
tail -15 $out/new_tests.log
python3 -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=8 > $out/gpu_tests_full.log 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests_full.log
tail -4 $out/gpu_tests_full.log
for rnd in 1 2; do for v in 0 1 2; do
  r=$(GSTVD_GROUP_ORDER=$v python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "GSTVD_GROUP_ORDER=$v round $rnd: ms_per_step value = $r" | tee -a $out/order_ab.txt
done; done
for v in 0 1 2; do
  GSTVD_GROUP_ORDER=$v rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$v -- python3 bench.py --steps 10 --warmup 3 $LEAN > $out/prof$v.log 2>&1
  f=$(ls $out/stats$v/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats_order$v.csv; head -4 $out/kernel_stats_order$v.csv | cut -c1-160
  rm -rf $out/stats$v
done
for v in 0 2; do for c in FETCH_SIZE WRITE_SIZE; do
  GSTVD_GROUP_ORDER=$v rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc${v}_$c -- python3 bench.py --steps 2 --warmup 1 $LEAN --graph off > /dev/null 2>&1
done; done
python3 - <<'PY' | tee gpurun_out/r05_s1/pmc_order.txt
import csv, glob, collections
for v in (0, 2):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in sorted(glob.glob('gpurun_out/r05_s1/pmc%d_*/*/*counter_collection.csv' % v)):
        for r in csv.DictReader(open(f)):
            if 'grouped_adamw' in r['Kernel_Name']:
                a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    d = {c: x[1] / max(x[0], 1) for c, x in agg.items()}
    print("GSTVD_GROUP_ORDER=%d grouped_adamw: launches %s FETCH_SIZE %.0f KB WRITE_SIZE %.0f KB -> 2*fetch+write = %.2f GB per launch"
          % (v, {c: x[0] for c, x in agg.items()}, d.get('FETCH_SIZE', 0), d.get('WRITE_SIZE', 0), (2 * d.get('FETCH_SIZE', 0) + d.get('WRITE_SIZE', 0)) * 1024 / 1e9))
PY
rm -rf $out/pmc*_FETCH_SIZE $out/pmc*_WRITE_SIZE
for sh in off on; do
  GSTVD_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 2 --grad-compress bf16 --shard-update $sh --legs off $LEAN 2>$out/dist_$sh.err | tail -1 > $out/bench_force_dist_shard_$sh.json
  python3 -c "import json; d=json.load(open('$out/bench_force_dist_shard_$sh.json')); print('shard-update $sh:', d['ms_per_step'], d['config'].get('optimizer_update'), d['config'].get('hip_graph'))" | tee -a $out/shard_ab.txt
done
