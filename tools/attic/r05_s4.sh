#!/bin/bash
# Round 5, GPU session 4: SQ counters of the attention backward (one-pass vs two-part), split-K ring depth A/B
export TMPDIR=/tmp; out=gpurun_out/r05_s4; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
for v in 1 0; do
  GSTVD_ATTN_ONEPASS=$v bash tools/pmc_attn.sh 2>&1 | grep -v amdgpu > $out/pmc_attn_onepass$v.txt; tail -14 $out/pmc_attn_onepass$v.txt
done
for rnd in 1 2; do for v in 8 4; do
  r=$(GSTVD_GEMM64_SK_NS=$v python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "GSTVD_GEMM64_SK_NS=$v round $rnd: ms_per_step value = $r" | tee -a $out/gemm64_sk_ns_ab.txt
done; done
