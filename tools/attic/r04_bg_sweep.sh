#!/bin/bash
# whole-step sweep: background (resident-grid) AdamW for every slice but the last x backward-pipeline slice lists; two interleaved rounds
mkdir -p gpurun_out/r4
for rnd in 1 2; do
  for bg in 0 2; do
    for cl in "22,27,27,27,27,192" "500" "22,27,27,27,27,64,64,64" "130,64,64,64" "22,27,27,27,27,48,48,48,48" "130,96,64,32"; do
      r=$(GSTVD_ADAMW_BG=$bg python3 bench.py --steps 30 --warmup 5 --chunk-list $cl --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
      echo "round $rnd BG=$bg chunk-list $cl: $r"
    done
  done
done
