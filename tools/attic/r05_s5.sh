#!/bin/bash
# Round 5, GPU session 5: which text-chain tile widths leave room for the vision chain (whole-step A/B of existing switches), timeline
export TMPDIR=/tmp; out=gpurun_out/r05_s5; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
ab() {  # name, env assignments...
  name=$1; shift
  r=$(env "$@" python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "$name [$*]: ms_per_step value = $r" | tee -a $out/tile_width_ab.txt
}
for rnd in 1 2; do
  ab base X=0
  ab n96_off GSTVD_GEMM128_N96=0
  ab niu4 GSTVD_GEMM256_NIU=4
  ab both GSTVD_GEMM128_N96=0 GSTVD_GEMM256_NIU=4
  ab ns128_5 GSTVD_GEMM128_NS=5
done
bash tools/trace_step.sh > $out/timeline.txt 2>&1; cp gpurun_out/trace/kernel_trace.csv.gz $out/kernel_trace_steps.csv.gz; head -40 $out/timeline.txt
python3 tools/overlap_stats.py $out/kernel_trace_steps.csv.gz > $out/overlap_stats.txt 2>&1; cat $out/overlap_stats.txt | head -40
