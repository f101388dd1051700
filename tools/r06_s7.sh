#!/bin/bash
# Round 6, GPU session 7: four consumer waves with 128 x 96 wave tiles in the full-line NT 256 x 192 kernel (pc4_tile_nt64, gemm_dma256.hip).
# Two builds of the library (lib/libgstvd_hip.so = GEMM_PC4=1, lib/libgstvd_hip_v0.so = -DGEMM_PC4=0), swapped in place.
export TMPDIR=/tmp; out=gpurun_out/r06_s7; rm -rf $out; mkdir -p $out
L=gst_visdial_amd/lib; cp $L/libgstvd_hip.so $L/v1.keep
use() { cp $L/$1 $L/libgstvd_hip.so; }
LEAN="--no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-rows-sensitivity --no-breakdown"
timeout 900 python3 -X faulthandler -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider -k "gemm" > $out/gemm_tests.log 2>&1; echo "rc=$?" >> $out/gemm_tests.log; tail -3 $out/gemm_tests.log
timeout 600 python3 tools/gemm_fuzz.py > $out/gemm_fuzz.log 2>&1; tail -3 $out/gemm_fuzz.log
for v in v1.keep libgstvd_hip_v0.so v1.keep libgstvd_hip_v0.so; do use $v
  echo "== $v" | tee -a $out/gemm_pc4_ab.txt
  python3 tools/gemm_bench.py main 2>/dev/null | grep -E "nt  4096x 3072|nt  4096x 2304|nn  4096x 3072" | cut -c1-100 | tee -a $out/gemm_pc4_ab.txt
done
for v in v1.keep libgstvd_hip_v0.so; do use $v; echo "== $v" | tee -a $out/gemm_pc4_ab.txt; python3 tools/nt_study.py 2>/dev/null | grep -v amdgpu | cut -c1-260 | tee -a $out/gemm_pc4_ab.txt; done
for rnd in 1 2 3; do for v in libgstvd_hip_v0.so v1.keep; do use $v
  r=$(python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "$v round $rnd: ms_per_step value = $r" | tee -a $out/step_pc4_ab.txt
done; done
use v1.keep; rm -f $L/v1.keep
