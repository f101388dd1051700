#!/bin/bash
# Round 5, GPU session 14: (a) premise of a one-launch backward for the few-query / few-key attention shapes: the two halves of the
# two-part kernel timed alone (GSTVD_ATTN_PROBE_SKIP, a probe switch that existed for this session only); (b) the column reductions
# of the last slice from the vision stream (GSTVD_COLSUM_SIDE); (c) GPU suite.
export TMPDIR=/tmp; out=gpurun_out/r05_s14; rm -rf $out; mkdir -p $out
for shape in "16 12 25 293 64 0 768 18432 0.1" "16 8 37 256 128 0 3072 3072 0.1" "16 8 256 37 128 0 3072 3072 0.1" "16 12 25 25 64 1 2304 2304 0.1" "16 8 37 37 128 0 3072 3072 0.1"; do
  for pr in 0 1 2; do
    echo -n "probe_skip=$pr  " | tee -a $out/attn_halves.txt
    GSTVD_ATTN_PROBE_SKIP=$pr build/attn_bench $shape bwd 2>&1 | tail -1 | tee -a $out/attn_halves.txt
  done
  build/attn_bench $shape 2>&1 | tail -1 | tee -a $out/attn_halves.txt
done
bash tools/r04_step_ab.sh GSTVD_COLSUM_SIDE 0 1 2>&1 | tee $out/colsum_side_ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $out/gpu_tests.log
