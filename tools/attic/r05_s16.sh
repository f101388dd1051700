#!/bin/bash
# Round 5, GPU session 16: block order of the two-part attention backward (all blocks of the longer-running class first, 1-D grid):
# op tests, stand-alone timings of the step's shapes under GSTVD_ATTN_BWD_ORDER = 1 (dK/dV first) / 2 (dQ first) / 0 (rule), step A/B.
export TMPDIR=/tmp; out=gpurun_out/r05_s16; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "attn or attention" 2>&1 | tail -4 | tee $out/attn_tests.log
for shape in "16 12 25 293 64 0 768 18432 0.1" "16 8 37 256 128 0 3072 3072 0.1" "16 8 256 37 128 0 3072 3072 0.1" "16 12 25 25 64 1 2304 2304 0.1" "16 8 37 37 128 0 3072 3072 0.1" "16 12 256 256 64 1 2304 2304 0.1" "10 12 25 293 64 0 768 18432 0.1" "10 8 256 37 128 0 3072 3072 0.1"; do
  for o in 1 2 0; do
    echo -n "order=$o  " | tee -a $out/attn_order.txt
    GSTVD_ATTN_BWD_ORDER=$o build/attn_bench $shape bwd 2>&1 | tail -1 | tee -a $out/attn_order.txt
  done
done
bash tools/r04_step_ab.sh GSTVD_ATTN_BWD_ORDER 1 0 2>&1 | tee $out/step_ab.txt
timeout 600 python -m pytest tests/test_fused_update_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $out/fused_tests.log
