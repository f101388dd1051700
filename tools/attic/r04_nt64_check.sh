#!/bin/bash
# round 4: full-line (64-deep alternating) staging of the row-major NT 256-tile GEMM: correctness (op tests + randomised regression),
# stand-alone timings and K slopes with the switch off / on, whole-step A/B
python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider -k "gemm" 2>&1 | tail -3
python3 tools/gemm_fuzz.py 300 4 2>&1 | tail -4
for v in 0 1; do
  echo "== GSTVD_GEMM_NT64=$v"
  GSTVD_GEMM_NT64=$v python3 tools/gemm_bench.py main 2>/dev/null | grep -E "^nt +(4096x 3072|4096x 2304|4688x18432)"
  GSTVD_GEMM_NT64=$v python3 tools/gemm_bench.py small 2>/dev/null | grep -E "400x30528x"
  GSTVD_GEMM_NT64=$v python3 tools/nt_study.py 2>/dev/null | grep -E "^nt 4096x(3072|2304)" | sed 's/   vendor.*//'
done
python3 tools/nt_study.py 2>/dev/null | grep -E "^nt 4096x(3072|2304)" | sed 's/.*   vendor/vendor/'
bash tools/r04_step_ab.sh GSTVD_GEMM_NT64 0 1
