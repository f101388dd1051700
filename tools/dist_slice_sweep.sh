#!/bin/bash
# The N>1 code path on one GPU (GSTVD_FORCE_DIST=1: 1-rank RCCL group, bf16 payload, captured) under different slice lists
out=gpurun_out/r2/dist_slice_sweep.txt; mkdir -p gpurun_out/r2; : > $out
for rows in 16 10; do for c in "" 128,48,48,32,32,24 128,128,64,16 192,96,48,16 128,128,96,16 160,128,48,16; do
  echo "== rows=$rows chunks=${c:-default}" >> $out
  GSTVD_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 5 --rows-per-gpu $rows --grad-compress bf16 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d ${c:+--chunk-list $c} 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config'].get('final_loss'))" >> $out
done; done
cat $out
