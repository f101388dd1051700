#!/bin/bash
# N=1 slice-list sweep of the backward pipeline (bench.py --chunk-list, Mi elements): does finishing the decoder's gradients
# (weight-gradient GEMMs + AdamW) in small slices DURING the decoder's latency-bound backward pay, instead of beside the
# encoder's throughput-bound backward?   usage: tools/chunk_sweep.sh "LIST1" "LIST2" ...
mkdir -p gpurun_out
for L in "$@"; do
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-breakdown --chunk-list "$L" 2>/dev/null \
    | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk-list %-28s %.3f ms/step  %.1f rounds/s' % ('$L', d['ms_per_step'], d['value']))"
done | tee -a gpurun_out/chunk_sweep.txt
