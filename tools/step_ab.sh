#!/bin/bash
# A/B of an environment switch on the whole train step (bench.py, lean flags): usage  step_ab.sh VAR v1 v2 ... ; two rounds interleaved
var=$1; shift
for rnd in 1 2; do
  for v in "$@"; do
    r=$(env $var=$v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "$var=$v round $rnd: ms_per_step value = $r"
  done
done
