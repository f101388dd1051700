#!/bin/bash
# L2 prefetch distance sweep of the NT64 tile (fills ahead; 0 = off): correctness spot check + K slopes + whole step
for pf in 0 2 4 6; do
  echo "== GSTVD_NT64_PF=$pf"
  GSTVD_NT64_PF=$pf python3 tools/gemm_bench.py main 2>/dev/null | grep -E "^nt +(4096x 3072|4096x 2304|4688x18432)"
  GSTVD_NT64_PF=$pf python3 tools/nt_study.py 2>/dev/null | grep -E "^nt 4096x(3072|2304)" | sed 's/   vendor.*//'
done
bash tools/r04_step_ab.sh GSTVD_NT64_PF 0 4
