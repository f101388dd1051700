#!/bin/bash
# Round 6, GPU session 13: the vision stream forks BEFORE the text embedding and stages its own inputs (mask concat, feature cast) -- three small
# dependent launches leave the head of the critical chain.  Tests that touch the engine's input path + step A/B (GSTVD_AB_OLD_HEAD=1: previous order).
export TMPDIR=/tmp; out=gpurun_out/r06_s13; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-rows-sensitivity"
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_full_config_gpu.py tests/test_round6_gpu.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc $?"; tail -3 $out/tests.log
for round in 1 2 3; do
  for old in 1 0; do
    GSTVD_AB_OLD_HEAD=$old timeout 600 python3 bench.py --steps 30 --warmup 5 $LEAN --no-breakdown > $out/bench_old${old}_$round.log 2> $out/bench_old${old}_$round.err
    python3 - $out/bench_old${old}_$round.log $old $round <<'PY' | tee -a $out/step_ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print("old_head=%s round %s: ms_per_step %.3f value %.1f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"]))
except Exception as e:
    print("old_head=%s round %s: FAILED %r" % (sys.argv[2], sys.argv[3], e))
PY
  done
done
GSTVD_FORCE_DIST=1 timeout 600 python3 bench.py --steps 20 --warmup 5 $LEAN --no-breakdown --legs off 2> $out/dist.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('forced-dist new head: ms_per_step %.3f hip_graph %s' % (d['ms_per_step'], d['config'].get('hip_graph')))" | tee -a $out/step_ab.txt
