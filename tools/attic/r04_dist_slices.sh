#!/bin/bash
# N>1 code path on one GPU (GSTVD_FORCE_DIST=1: 1-rank RCCL group, bf16 payload, captured), slice lists re-examined after round 4's
# finding that a slice's AdamW stalls the decoder's chain: decoder in five slices (default) / in one / in two; two interleaved rounds
mkdir -p gpurun_out/r4
for rnd in 1 2; do
  for rows in 16 10; do
    for c in "22,27,27,27,27,96,96,32,16" "130,96,96,32,16" "49,81,96,96,32,16" "130,128,64,32,16" "22,108,96,96,32,16"; do
      r=$(GSTVD_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 5 --rows-per-gpu $rows --grad-compress bf16 --chunk-list $c --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'])")
      echo "round $rnd rows $rows chunk-list $c: $r"
    done
  done
done
