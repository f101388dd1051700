#!/bin/bash
# Round 6, GPU session 12: the one-pass attention backward with the dQ product of chunk c deferred into chunk c + 1's interval
# (double-buffered dS image, two barriers per chunk instead of three): op tests, stand-alone time, step A/B (GSTVD_ATTN_ONEPASS_PIPE 0 / 1).
export TMPDIR=/tmp; out=gpurun_out/r06_s12; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-rows-sensitivity"
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_round5_gpu.py -x -q -m gpu -k "attn or attention or keep or onepass" > $out/tests.log 2>&1; echo "tests rc $?"; tail -3 $out/tests.log
for pipe in 0 1 0 1; do
  for bits in 1 0; do
    GSTVD_ATTN_ONEPASS_PIPE=$pipe timeout 120 python3 tools/attn_probe.py 16 12 256 256 64 0 0.1 50 $bits 2>/dev/null | sed "s/^/pipe=$pipe /" | tee -a $out/attn_probe.txt
  done
done
for round in 1 2; do
  for pipe in 0 1; do
    GSTVD_ATTN_ONEPASS_PIPE=$pipe timeout 600 python3 bench.py --steps 30 --warmup 5 $LEAN --no-breakdown > $out/bench_pipe${pipe}_$round.log 2> $out/bench_pipe${pipe}_$round.err
    python3 - $out/bench_pipe${pipe}_$round.log $pipe $round <<'PY' | tee -a $out/step_ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print("pipe=%s round %s: ms_per_step %.3f value %.1f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"]))
except Exception as e:
    print("pipe=%s round %s: FAILED %r" % (sys.argv[2], sys.argv[3], e))
PY
  done
done
