#!/bin/bash
# Round 6: every measurement committed under profiles/r06_* that describes the FINAL build, from ONE script (run on the GPU box; outputs
# under gpurun_out/final, copied to profiles/ by tools/collect_profiles.py r06).   bash tools/final_profiles_r06.sh
# (The records behind the round's decisions were written by tools/r06_s1.sh .. r06_s5.sh while the decisions were made.)
export TMPDIR=/tmp; out=gpurun_out/final; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d --no-rows-sensitivity"
# 1. the full GPU test-suite, whole log kept (the driver's command)
python3 -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=12 > $out/gpu_tests_full.log 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests_full.log
tail -3 $out/gpu_tests_full.log
# 2. the contract line with the driver's flags + per-kernel event breakdown (all side measurements on); wall time of the whole command
t0=$(date +%s.%N)
python3 bench.py --steps 20 --warmup 5 --breakdown-json $out/breakdown_events.json > $out/bench_stdout.log 2> $out/bench_stderr.log
echo "python3 bench.py --steps 20 --warmup 5 (the driver's flags, all side measurements on): $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s wall" > $out/bench_wall.txt
tail -1 $out/bench_stdout.log > $out/bench_n1.json; cat $out/bench_wall.txt; cut -c1-260 $out/bench_n1.json
# 3. rocprofv3 --stats of the same command (kernel averages must agree with roofline.avg_launch_us)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 $LEAN > $out/prof_bench.log 2>&1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats_bench.csv; rm -rf $out/stats
head -8 $out/kernel_stats_bench.csv | cut -c1-150
# 4. HBM traffic per kernel family inside the step (separate --pmc passes) and MFMA pipe utilisation of the step's GEMM shapes
bash tools/pmc_bench.sh > $out/pmc.log 2>&1; cp gpurun_out/pmc3/summary.json $out/pmc_traffic.json; rm -rf gpurun_out/pmc3
bash tools/pmc_mfma.sh > $out/pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma/summary.json $out/pmc_mfma_util.json; rm -rf gpurun_out/pmc_mfma
# 5. timeline of a steady-state replayed step + the compressed kernel trace + alone / overlapped durations per kernel family;
#    and the same for the FORCED-DIST step (1-rank RCCL group, all-reduce form): what the N > 1 step is made of
bash tools/trace_step.sh --no-eval-decode --no-rows-sensitivity > $out/timeline.txt 2>&1; cp gpurun_out/trace/kernel_trace.csv.gz $out/kernel_trace_steps.csv.gz
python3 tools/overlap_stats.py $out/kernel_trace_steps.csv.gz > $out/overlap_stats.txt 2>&1
GSTVD_FORCE_DIST=1 bash tools/trace_step.sh --no-eval-decode --no-rows-sensitivity --grad-compress bf16 --legs off > $out/timeline_dist.txt 2>&1
( echo; echo "== per-kernel totals of the forced-dist step"; python3 tools/by_kernel.py ) >> $out/timeline_dist.txt 2>&1
# 6. the same GEMMs against the vendor BLAS (speed-of-light reference only) + K slopes
( python3 tools/gemm_bench.py all lib; python3 tools/nt_study.py ) 2>/dev/null | grep -v amdgpu > $out/gemm_vs_vendor_blas.txt
# 7. launch floor: the same graph at 2 rows x 32 tokens, kernels per step
( echo "== bench.py at 2 rows x 32 tokens (same number of graph nodes)"; python3 bench.py --rows-per-gpu 2 --seq-len 32 --steps 30 --warmup 3 $LEAN 2>/dev/null | tail -1 | cut -c1-200
  bash tools/trace_step.sh --rows-per-gpu 2 --seq-len 32 --no-eval-decode --no-rows-sensitivity > /dev/null 2>&1; python3 tools/by_kernel.py ) > $out/launch_floor.txt 2>&1
# 8. the bench's launch paths on a 1-GPU box + the N > 1 code path on a 1-rank RCCL group: whole-step graph, segmented, eager; legs in fresh children
( echo "== python3 bench.py --gpus 2 on a 1-GPU box"; python3 bench.py --gpus 2 --steps 3 --warmup 1; echo "exit code $? (must be non-zero)"
  echo "== GSTVD_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 (self-launched; legs as fresh child processes)"
  GSTVD_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 $LEAN 2>/dev/null | tail -1; echo "exit code $?"
  echo "== python3 bench.py --graph off (eager issue, N = 1)"
  python3 bench.py --steps 10 --warmup 3 --graph off $LEAN 2>/dev/null | tail -1 | cut -c1-400 ) > $out/bench_launch_paths.txt 2>&1
for rows in 16 10; do GSTVD_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 2 --rows-per-gpu $rows --grad-compress bf16 --legs off $LEAN 2>/dev/null | tail -1 > $out/bench_force_dist_rows$rows.json; done
GSTVD_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 3 --legs on $LEAN 2>/dev/null | tail -1 > $out/bench_force_dist_legs.json
python3 bench.py --steps 10 --warmup 2 --rows-per-gpu 10 --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-rows-sensitivity 2>/dev/null | tail -1 > $out/bench_n1_rows10.json
tail -4 $out/bench_launch_paths.txt | cut -c1-300; cut -c1-200 $out/bench_force_dist_legs.json
# 9. eval / decode side measurements (decode with and without the questioner's 4-gram ban)
python3 tools/eval_decode_bench.py > $out/eval_decode.json 2> /dev/null; cat $out/eval_decode.json
# 10. 12 clean fresh-process runs of the 1-rank RCCL step: whole-step graph (all-reduce and sharded update) and the segmented form
( for i in 1 2 3 4; do for cfg in "auto off" "auto on" "segmented off"; do set -- $cfg
    GSTVD_FORCE_DIST=1 python3 bench.py --steps 5 --warmup 2 --grad-compress bf16 --graph $1 --shard-update $2 --legs off $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('run $i --graph $1 --shard-update $2: rc ok', d['ms_per_step'], 'ms, hip_graph', d['config']['hip_graph'], d['config']['capture_quiesce'], 'payload by wgrad launch', d['config'].get('payload_written_by_wgrad_launch'))" || echo "run $i --graph $1 --shard-update $2: FAILED"
  done; done ) > $out/rccl_soak.txt 2>&1; tail -4 $out/rccl_soak.txt
# 11. the driver's bench command three more times on this box (repeatability)
( for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-rows-sensitivity 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('run $i:', d['value'], 'rounds/s', d['ms_per_step'], 'ms; roofline frac', r['frac'], 'avg_launch_us', r['avg_launch_us'], 'rocprof', r.get('avg_launch_us_rocprof'))"; done ) > $out/bench_repeat.txt 2>&1; cat $out/bench_repeat.txt
