#!/bin/bash
# Round 5, GPU session 22: the driver's bench command five times in fresh processes on one box (run-to-run and warm-up spread of `value`).
export TMPDIR=/tmp; out=gpurun_out/r05_s22; rm -rf $out; mkdir -p $out
for i in 1 2 3 4 5; do
  python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('run $i: value %.1f rounds/s, %.3f ms/step; dominant kernel %.1f us (roofline.frac %.3f), coattn_frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], r['coattn_frac']))" | tee -a $out/bench_repeat.txt
done
