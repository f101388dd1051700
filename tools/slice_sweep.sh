#!/bin/bash
# N = 1 sweep of the backward pipeline's slice sizes, with and without the separate AdamW stream (run on the GPU box)
out=gpurun_out/r2/slice_sweep.txt; mkdir -p gpurun_out/r2; : > $out
run() { echo "== US=$1 chunks=$2" >> $out; GSTVD_PIPE_UPDATE_STREAM=$1 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d ${2:+--chunk-list $2} 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config'].get('final_loss'))" >> $out; }
for us in 0 1; do for c in "" 128,48,48,32,32,24 160,64,48,32,16 96,96,64,32,16,8 128,128,64,16 64,64,64,64,32,32,16,8; do run $us "$c"; done; done
cat $out
